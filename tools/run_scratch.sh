tests/cpp/host_tests tests/golden/tbl 2>&1 | tail -12

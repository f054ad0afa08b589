cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_aggregate_gpu.py -x -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -k aggregate 2>&1 | tail -3
timeout 300 python tools/agg_debug.py 2>&1 | tail -8

mkdir -p gpurun_out
timeout 300 tests/cpp/multi_gpu_tests 0 0 > gpurun_out/mg2.log 2>&1; echo "rc=$?" >> gpurun_out/mg2.log
timeout 300 tests/cpp/multi_gpu_tests 0 0 0 > gpurun_out/mg3.log 2>&1; echo "rc=$?" >> gpurun_out/mg3.log
tail -20 gpurun_out/mg2.log; tail -8 gpurun_out/mg3.log
timeout 600 python -m pytest tests/test_multi_gpu_cpp_gpu.py tests/test_host_mirror_gpu.py -q -m gpu 2>&1 | tail -5

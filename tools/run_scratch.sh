mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/gputest.log
tail -6 gpurun_out/gputest.log
timeout 300 python tools/agg_debug.py > gpurun_out/agg_debug.log 2>&1; cat gpurun_out/agg_debug.log
bash tools/run_ssb_profile.sh > gpurun_out/ssb_profile.txt 2>&1; tail -40 gpurun_out/ssb_profile.txt

mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/gputest.log
tail -4 gpurun_out/gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
bash tools/collect_profiles.sh "$1"
bash tools/run_ssb_profile.sh > gpurun_out/ssb_profile.txt 2>&1

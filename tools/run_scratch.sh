mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/gputest.log
tail -5 gpurun_out/gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
bash tools/collect_profiles.sh "$1"
timeout 300 tools/hbm_mix_bin 3 > gpurun_out/hbm_mix.txt 2>&1; tail -3 gpurun_out/hbm_mix.txt
timeout 300 tools/hbm_write_bin > gpurun_out/hbm_write.txt 2>&1; tail -3 gpurun_out/hbm_write.txt

cd /root/repo
timeout 900 python -m pytest tests/test_sharded_gpu.py -x -q -s -m gpu -k "one_c_call or rank_simulation or range_flag or ranks_match_untiled" > gpurun_out/r5_run4_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5_run4_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r5_run4_bench.json 2> gpurun_out/r5_run4_bench.err
echo "rc=$?" >> gpurun_out/r5_run4_bench.err

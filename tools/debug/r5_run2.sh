cd /root/repo
bash tools/experiments/winograd_go_nogo.sh r05 > gpurun_out/r5_run2_wino.log 2>&1
timeout 900 python -m pytest tests/test_sharded_gpu.py -x -q -s -m gpu -k "geometry_vs_reference" > gpurun_out/r5_run2_g16.log 2>&1
timeout 600 python bench.py --steps-only --no-cpu-baseline > gpurun_out/r5_run2_bench.json 2> gpurun_out/r5_run2_bench.err

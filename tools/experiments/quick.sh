python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_quick.json | python tools/show_bench.py

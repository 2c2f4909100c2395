for l in 4 8 16; do echo "== LPP $l"; WCT_JACOBI_LPP=$l python tools/experiments/jacobi_bench.py 2>&1 | grep -v amdgpu.ids; done

# per-phase cycle breakdown of the DMA-staged conv kernel: builds a second library with -DWCT_SP_TIMING
set -e
cd collaborative-distillation_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWCT_SP_TIMING -o libwct_hip_timing.so csrc/conv3x3.hip csrc/conv3x3_f16.hip csrc/conv3x3_sp.hip csrc/level1.hip csrc/moments.hip csrc/solve.hip csrc/misc.hip csrc/wct_api.hip
cd ..
echo "built collaborative-distillation_amd/libwct_hip_timing.so; on the GPU box: WCT_LIB_PATH=\$PWD/collaborative-distillation_amd/libwct_hip_timing.so python tools/experiments/sp_timing.py"

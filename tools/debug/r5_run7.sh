cd /root/repo
python tools/experiments/mom_reg_ab.py > gpurun_out/r5_run7_mom.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "moments or full_size_properties or g4 or g3 or levels_vs" > gpurun_out/r5_run7_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5_run7_tests.log
for i in 1 2; do python bench.py --steps 20 --warmup 3 --steps-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ks={k['kernel']:k['ms_per_step'] for k in d['kernels']}
print('ms_per_step', d['ms_per_step'], 'moments', ks.get('moments'), 'l1_moments', ks.get('l1_moments_fused<3-24>'))"; done >> gpurun_out/r5_run7_mom.log 2>&1

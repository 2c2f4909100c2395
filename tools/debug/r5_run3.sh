cd /root/repo
L=$PWD/tools/experiments/libs
echo "== bitwise: current vs r5base" > gpurun_out/r5_run3_ab.log
python tools/experiments/ab_equal.py $L/libwct_r5base.so >> gpurun_out/r5_run3_ab.log 2>&1
echo "== timing A/B (current, r5base, b64 = without the permlane16 store, cs44 = without the l1_moments swizzle)" >> gpurun_out/r5_run3_ab.log
bash tools/experiments/ab_libs.sh "$L/libwct_r5base.so $L/libwct_b64.so $L/libwct_cs44.so" enc_head dec_tail l1_moments l1_decode >> gpurun_out/r5_run3_ab.log 2>&1
SQ=1 bash tools/profile_round.sh r05a cfg2 > gpurun_out/r5_run3_prof.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fused or head or level1 or moments or g4 or g2 or odd or tail or g9 or g7" > gpurun_out/r5_run3_tests.log 2>&1

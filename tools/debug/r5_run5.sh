cd /root/repo
python tools/experiments/mom_reg_ab.py > gpurun_out/r5_run5_mom.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "moments or full_size_properties or g4 or g3 or levels_vs or odd" > gpurun_out/r5_run5_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r5_run5_tests.log
L=$PWD/tools/experiments/libs
bash tools/experiments/ab_libs.sh "$L/libwct_b64.so" moments >> gpurun_out/r5_run5_mom.log 2>&1

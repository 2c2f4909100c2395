cd /root/repo
ls /sys/class/drm/ > gpurun_out/r5_sysfs.txt 2>&1
for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $d"; ls $d; for f in power1_average power1_input freq1_input freq2_input temp1_input temp2_input; do [ -f $d/$f ] && echo "$f: $(cat $d/$f)"; done; done >> gpurun_out/r5_sysfs.txt 2>&1
timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "geometry_vs_reference or capturable or launches_itself or line_contract or device_buffer_collectives or rank_simulation or two_devices or full_size_properties" > gpurun_out/r5_run1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_run1_tests.log
timeout 900 python bench.py > gpurun_out/r5_run1_bench.json 2> gpurun_out/r5_run1_bench.err
echo "bench rc=$?" >> gpurun_out/r5_run1_bench.err

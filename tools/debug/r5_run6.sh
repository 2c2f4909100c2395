cd /root/repo
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/r5_run6_gputests.log 2>&1
echo "rc=$?" >> gpurun_out/r5_run6_gputests.log

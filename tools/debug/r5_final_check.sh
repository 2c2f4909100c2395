cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_final_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r5_final_smoke.log
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_final_gputests.log 2>&1
echo "rc=$?" >> gpurun_out/r5_final_gputests.log

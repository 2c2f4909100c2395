cd /root/repo
timeout 1200 python bench.py > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err
echo "rc=$?" >> gpurun_out/r5_final_bench.err
SQ=1 bash tools/profile_round.sh r05d cfg2 > gpurun_out/r5_final_prof.log 2>&1
bash tools/profile_round.sh r05d_original cfg3 > gpurun_out/r5_final_prof3.log 2>&1

cd /root/repo
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --steps-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,socket
d=json.loads(sys.stdin.read()); t=d['gpu_telemetry'] or {}
print(json.dumps({'ms_per_step':d['ms_per_step'],'latency_ms_median':d['latency_ms_median'],'power_W_avg':t.get('power_W_avg'),'power_W_max':t.get('power_W_max'),'sclk_MHz_avg':t.get('sclk_MHz_avg'),'sclk_MHz_min':t.get('sclk_MHz_min'),'temp_C':t.get('temp_C_avg'),'pci':(t.get('source') or '')[34:46]}))"
done >> gpurun_out/r5_spread.jsonl

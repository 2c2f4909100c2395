for n in 256 512 1024; do echo "== NPC target $n"; WCT_MOM_NPC=$n python tools/experiments/moments_bench.py 2>&1 | grep -v amdgpu; done

"""Aggregate the per-launch phase lines printed by a -DWCT_SP_TIMING build (stdin) by layer shape."""
import collections, re, sys
acc = collections.defaultdict(list)
for l in sys.stdin:
    m = re.match(r"sp cin=(\d+) cout=(\d+) (\d+)x(\d+) pool=\d+: per job-wave cycles: vmwait (\d+) barrier (\d+) issue\+mfma (\d+) epi\+rest (\d+)", l)
    if m:
        acc[tuple(map(int, m.groups()[:4]))].append(tuple(map(int, m.groups()[4:])))
print("cin cout    HxW      n | vmwait barrier   mfma    epi | chunks  MFMA-ideal/job (2 waves/SIMD)  tiles")
for k, v in sorted(acc.items(), key=lambda kv: -kv[0][2] * kv[0][3] * kv[0][0] * kv[0][1]):
    n = len(v)
    a = [sum(x[i] for x in v) / n for i in range(4)]
    cin, cout, H, W = k
    ct = 2 if cout % 64 == 0 else 1
    tiles = ((W + 31) // 32) * ((H + 15) // 16)
    print(f"{cin:4d} {cout:4d} {H:4d}x{W:<4d} n={n:3d} tot={sum(a):6.0f} | {a[0]:6.0f} {a[1]:6.0f} {a[2]:6.0f} {a[3]:6.0f} | {cin // 16:3d} {9 * ct * 2 * 3 * 32 * 2:6d} {tiles:6d} ({tiles / 256:.1f}/CU)")

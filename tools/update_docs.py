#!/usr/bin/env python3
"""Regenerate the measured-figures blocks of README.md and DESIGN.md from ONE bench.py line (VERDICT r3 task 8: the documents used to quote
figures of builds that no longer existed).  The block between `<!-- BENCH:BEGIN -->` and `<!-- BENCH:END -->` in each file is replaced;
everything else is prose and stays hand-written.   usage: python tools/update_docs.py profiles/r04x_bench_line.json"""
import json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
d = json.load(open(src))
p = d.get("passes", {})
par = d.get("parity") or {}
c3 = p.get("cfg3_original", {})
c3p = c3.get("parity", {})
roof = d.get("roofline", {})
cpu = d.get("cpu_baseline") or {}
cli = p.get("cli_pairs_per_s", {})
g = lambda x, f="%.3g": (f % x) if isinstance(x, (int, float)) else "n/a"
lines = ["_All figures in this block come from one `bench.py` line, `%s` (one MI355X; regenerate with `python tools/update_docs.py <line>`)._" % os.path.relpath(src, REPO), ""]
lines.append("* **Headline** (BASELINE configs[1]: `--mode 16x`, 3840×2160 content + 2048×2048 style, style side included): **%s ms per frame = %s MP/s**." % (g(d.get("ms_per_step"), "%.3f"), g(d.get("value"), "%.1f")))
tel = d.get("gpu_telemetry") or {}
lines.append("* Latency (SURVEY 8d: each frame synchronised on its own, median of >= 10): **%s ms** (min %s, max %s); GPU during the timed steps: %s W average (max %s), sclk %s MHz average (min %s)." % (
    g(d.get("latency_ms_median"), "%.3f"), g((d.get("latency_ms_min_max") or [None, None])[0], "%.3f"), g((d.get("latency_ms_min_max") or [None, None])[1], "%.3f"),
    g(tel.get("power_W_avg"), "%.0f"), g(tel.get("power_W_max"), "%.0f"), g(tel.get("sclk_MHz_avg"), "%.0f"), g(tel.get("sclk_MHz_min"), "%.0f")))
f32 = p.get("cfg2_fp32_exact", {})
if f32:
    lines.append("* **The reference's own arithmetic class beside the headline** (`wct_set_conv_mode(0)`: exact-fp32 MFMA products in every convolution, the same frame and call): %s ms per frame = %s MP/s = %s of the 157.3 TF fp32-MFMA roofline (dominant `%s`: %s TF = %s); the f16x3 headline is %s× faster; G13 distance %s (limit 1e-3)." % (
        g(f32.get("ms_per_frame"), "%.2f"), g(f32.get("MPs"), "%.1f"), g(f32.get("algo_frac_fp32_mfma_157TF"), "%.3f"), (f32.get("roofline") or {}).get("kernel"),
        g((f32.get("roofline") or {}).get("achieved"), "%.1f"), g((f32.get("roofline") or {}).get("frac"), "%.3f"), g(f32.get("headline_f16x3_speedup"), "%.2f"),
        g((f32.get("parity") or {}).get("hip_vs_reference"), "%.2e")))
lines.append("* Parity on the timed frame (G13, the reference's own pixels): this library %s (p99.99 %s), oracle %s, limit %s; reference UHD pair (G11) %s." % (
    g(par.get("hip_vs_reference"), "%.2e"), g(par.get("hip_vs_reference_p9999"), "%.2e"), g(par.get("oracle_vs_reference"), "%.2e"), g(par.get("limit"), "%.0e"),
    g((par.get("reference_uhd_pair") or {}).get("hip_vs_reference"), "%.2e")))
lines.append("* Dominant kernel `%s`: %s %s = **%s** of the %s %s roofline (%s of the measured 420 TF random-operand ceiling); %s µs per launch; HBM traffic per launch %s MB against %s MB algorithmic." % (
    roof.get("kernel"), g(roof.get("achieved"), "%.1f"), roof.get("unit"), g(roof.get("frac"), "%.3f"), g(roof.get("peak"), "%.1f"), roof.get("unit"),
    g(roof.get("frac_of_measured_ceiling_420TF"), "%.2f"), g((roof.get("avg_launch_ms") or 0) * 1e3, "%.1f"), g((roof.get("traffic") or 0) / 1e6, "%.1f"), g((roof.get("algo_bytes_per_launch") or 0) / 1e6, "%.1f")))
r4 = p.get("relu4_1_encode", {})
c4 = p.get("cfg4_single_gpu", {})
lines.append("* relu4_1 encode pass (north_star): %s ms = %s of 8 TB/s at 4K; %s ms = %s at 10240×4096." % (g(r4.get("ms"), "%.3f"), g(r4.get("frac_hbm_8TBs"), "%.3f"), g(c4.get("relu4_1_encode_ms"), "%.3f"), g(c4.get("relu4_1_encode_frac_hbm_8TBs"), "%.3f")))
lines.append("* Other passes: cached style statistics %s ms; three frames in flight %s ms per frame (%s MP/s); 10240×4096 untiled on one GPU %s ms (%s MP/s); uint8 in → uint8 out over PCIe %s ms." % (
    g(p.get("style_cached_cascade", {}).get("ms"), "%.3f"), g(p.get("cfg5_per_gpu", {}).get("ms_per_frame"), "%.3f"), g(p.get("cfg5_per_gpu", {}).get("MPs"), "%.1f"),
    g(c4.get("ms_per_frame"), "%.2f"), g(c4.get("MPs"), "%.1f"), g(p.get("u8_end_to_end", {}).get("ms_per_frame"), "%.2f")))
if cli and "8_contents" in cli:
    c8, c32 = cli["8_contents"], cli.get("32_contents", {})
    lines.append("* The reference's timed region as a folder run (N × 4K JPEG contents × 1 style: decode → H2D → cascade → D2H → JPEG save; %s): N = 8: serial loop %s ms per pair, pipelined **%s ms per pair** (%s×; fill + drain of the pipeline included), outputs byte-identical: %s; N = 32: pipelined **%s ms per pair = %s pairs/s**." % (
        cli.get("workload", "").split(";")[-1].strip(), g(c8.get("serial", {}).get("ms_per_pair"), "%.1f"), g(c8.get("pipelined", {}).get("ms_per_pair"), "%.1f"), g(c8.get("speedup"), "%.2f"),
        c8.get("outputs_byte_identical"), g(c32.get("pipelined", {}).get("ms_per_pair"), "%.1f"), g(c32.get("pipelined", {}).get("pairs_per_s"), "%.1f")))
lines.append("* `--mode original`, 1920×1080 (BASELINE configs[2], generated weights; real-weight parity unpinned): %s ms per frame = %s MP/s, %s TF algorithmic; G14 (He-uniform stacks: chaotic under the reference's own arithmetic) this library %s against oracle %s, limit %s; **G15 (well-conditioned generated set, the reference's classes) %s at the literal 1e-3**." % (
    g(c3.get("ms_per_frame"), "%.2f"), g(c3.get("MPs"), "%.1f"), g(c3.get("algo_TFLOPs"), "%.1f"), g(c3p.get("hip_vs_reference"), "%.2e"), g(c3p.get("oracle_vs_reference"), "%.2e"), g(c3p.get("limit"), "%.2e"),
    g((c3p.get("g15_strict") or {}).get("hip_vs_reference"), "%.2e")))
for k, name in (("cfg4_rank_sim", "10240×4096 in 8 strips"), ("cfg2x8_rank_sim", "8 × 3840 columns (weak)")):
    rs = p.get(k, {})
    if rs:
        lines.append("* One rank's share of the 8-GPU job, timed on one GPU (%s): slowest rank of the best arrangement (`%s`) %s ms per frame → %s predicted%s (compute + orchestration only; no link time, no skew)." % (
            name, rs.get("arrangement"), g(rs.get("predicted_ms_per_frame"), "%.2f"), ("%s× one GPU" % g(rs.get("predicted_speedup_vs_1gpu"), "%.2f")) if rs.get("predicted_speedup_vs_1gpu") else ("%s efficiency" % g(rs.get("predicted_efficiency"), "%.3f")), " (round 5's arrangement on the same box: %s efficiency)" % g(rs.get("predicted_efficiency_round5_arrangement"), "%.3f") if rs.get("predicted_efficiency_round5_arrangement") else ""))
        for r, e in sorted((rs.get("ranks") or {}).items()):
            for tag, what in (("torch_distributed_owner", "round 5's arrangement: Python orchestration, style levels dealt out whole"),
                              ("torch_distributed", "Python orchestration over torch.distributed, style in strips"),
                              ("c_cascade_owner", "ONE library call per frame, RCCL inside (`wct_stylize_sharded`), style levels dealt out whole"),
                              ("c_cascade", "ONE library call per frame, RCCL inside (`wct_stylize_sharded`), style in strips")):
                t = e.get(tag)
                if t:
                    lines.append("  * rank %s, %s: %s ms per frame; host %s ms = %s ms pure enqueue (%s of the frame) + %s ms waiting for an old frame's range flag." % (
                        r, what,
                        g(t.get("ms_per_frame"), "%.2f"), g(t.get("host_enqueue_ms"), "%.2f"), g(t.get("pure_enqueue_ms"), "%.2f"), g(t.get("pure_enqueue_share_of_frame"), "%.2f"), g(t.get("range_flag_wait_ms"), "%.2f")))
lines.append("* CPU checker on the same host, the timed frame itself: %s MP/s on %s threads (%s)." % (g(cpu.get("value"), "%.3f"), cpu.get("cores"), cpu.get("kind")))
ks = d.get("kernels") or []
if ks:
    lines += ["", "| kernel family | ms per step | launches | TFLOP/s | algorithmic GB/s |", "|---|---|---|---|---|"]
    for k in ks[:16]:
        lines.append("| `%s` | %.3f | %d | %s | %s |" % (k["kernel"], k["ms_per_step"], k["launches_per_step"], g(k.get("tflops"), "%.1f"), g(k.get("algo_GBs"), "%.0f")))
block = "<!-- BENCH:BEGIN -->\n" + "\n".join(lines) + "\n<!-- BENCH:END -->"
for name in ("README.md", "DESIGN.md"):
    path = os.path.join(REPO, name)
    s = open(path).read()
    if "<!-- BENCH:BEGIN -->" not in s:
        print("%s: no BENCH block, skipped" % name)
        continue
    s = re.sub(r"<!-- BENCH:BEGIN -->.*?<!-- BENCH:END -->", lambda m: block, s, flags=re.S)
    open(path, "w").write(s)
    print("%s: block regenerated from %s" % (name, src))

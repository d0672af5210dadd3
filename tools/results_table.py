"""Markdown results table from a directory of bench lines (tools/collect_evidence.sh output): python tools/results_table.py <dir>"""
import json, os, sys
d = sys.argv[1]
def L(name):
    p = os.path.join(d, name)
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) else None
rows = [("BAIR 64² nf 64, B = 64 (configs[1], the metric's workload)", "bench_bair64.json"),
        ("  B = 16 (share of a 64-sample job on 4 GPUs)", "bench_bair64_b16.json"),
        ("  B = 8 (on 8 GPUs)", "bench_bair64_b8.json"),
        ("  B = 4 (on 16 GPUs; = configs[0]'s batch)", "bench_bair64_b4.json"),
        ("Landscape 128² nf 32, B = 32 (configs[2])", "bench_land128_b32.json"),
        ("DTDB 128², B = 32 (per-GPU share of configs[3])", "bench_dtdb128_b32.json"),
        ("DTDB 128², global 256 on ONE GPU (configs[3], strong, N = 1)", "bench_dtdb128_strong_b256.json"),
        ("128² vid_length 32, B = 16 (per-GPU share of configs[4])", "bench_iper128_t32_b16.json"),
        ("128² vid_length 32, global 128 on ONE GPU (configs[4], strong, N = 1)", "bench_iper128_t32_strong_b128.json")]
print("| workload | frames/s (pipelined stream) | ms/step | one serial call: ms (frames/s) | dominant kernel TFLOP/s alg. (frac of 2.5 P) | all 3×3×3 convs (frac) |")
print("|---|---|---|---|---|---|")
for name, f in rows:
    r = L(f)
    if not r:
        continue
    sc = r.get("single_call")
    print(f"| {name} | {r['value'] / 1e3:.1f} k | {r['ms_per_step']:.2f} | " + (f"{sc['ms']:.2f} ({sc['frames_per_s'] / 1e3:.1f} k)" if sc else "") +
          f" | {r['roofline']['achieved']:.0f} ({r['roofline']['frac']:.3f}) | {r['roofline_all_conv3']['achieved']:.0f} ({r['roofline_all_conv3']['frac']:.3f}) |")
r = L("bench_bair64.json")
if r:
    ro, c, e, s, cb = r["roofline"], r["roofline_cinn"], r["exact_fp32"], r["sustained"], r["cpu_baseline"]
    print()
    print(f"* default line: dominant kernel `{ro['kernel_name']}` {ro['ms_per_step']:.2f} ms of the step, MFMA issue frac {ro['mfma_issue_frac']:.3f}; "
          f"HBM traffic {ro['traffic'] / 1e9:.2f} GB per launch ({ro['traffic_source'][:40]}...)")
    print(f"* sustained: {s['ms_per_step']:.2f} ms per step over {s['seconds']:.1f} s ({s['steps']} steps); chunks {[round(x, 2) for x in s['ms_per_step_by_chunk']]}")
    print(f"* exact fp32 (mma = 0): {e['ms_per_step']:.1f} ms per step = {e['frames_per_s']:.0f} frames/s; its 3×3×3 convs {e['conv3_tflops']:.1f} TFLOP/s = {e['frac']:.3f} of 157.3")
    print(f"* cINN pass (B = 64): inverse {c['inv_latency_us']:.0f} µs / forward {c['fwd_latency_us']:.0f} µs; {c['achieved']:.0f} GB/s = {c['frac']:.4f} of 8 TB/s; "
          f"counted HBM bytes per pass {c['measured_hbm_bytes_per_pass'] / 1e6:.0f} MB ({c['measured_hbm_bytes_source'][:30]})")
    print(f"* CPU oracle, configs[0] (B = 4), {cb['cpu']}, {cb['cores']} threads: {cb['faithful']['frames_per_s']:.1f} frames/s faithful / {cb['folded']['frames_per_s']:.1f} folded")
    sm = r["roofline_all_conv3"].get("sustained_mfma")
    if sm:
        print(f"* MFMA-only loop on live operands on this box: {sm['peak_live_operands']:.0f} TFLOP/s (the power-limited ceiling; data sheet 2 500)")
    print(f"* embedder {r['embedder']['ms_per_batch']:.2f} ms / motion encoder {r['encoder']['ms_per_batch']:.2f} ms per 64 samples (not part of `value`)")

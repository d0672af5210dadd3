#!/usr/bin/env python3
"""HBM bytes per launch of every kernel of one bench step, from rocprofv3 PMC counters.

Run on the GPU box:   python tools/pmc_hbm_traffic.py gpurun_out/pmc_traffic [--config land128]
Two SEPARATE counter passes (FETCH_SIZE, then WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes;
FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read, so
read bytes = 2 * FETCH_SIZE * 1024 (the guide's gfx950 correction); WRITE_SIZE is used as reported.
Writes <outdir>/hbm_traffic.json; copy it to profiles/rNN_<x>_pmc_hbm_traffic.json (bench.py reads the newest such file for the
static `roofline.traffic` / `roofline_cinn.measured_hbm_bytes_per_pass` figures and names it as their source)."""
import collections
import csv
import json
import os
import subprocess
import sys


def one_pass(outdir, counter, extra, parse_only):
    d = os.path.join(outdir, counter.lower())
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"] + extra
    if not parse_only:
        env = dict(os.environ, TMPDIR="/tmp")
        with open(os.path.join(outdir, counter.lower() + ".log"), "w") as log:
            subprocess.run(cmd, check=True, stdout=log, stderr=subprocess.STDOUT, env=env)
    per = collections.defaultdict(lambda: [0, 0.0])
    conv16 = []
    n_dec = 0   # decoder passes in the run (one conv_img launch each): bench.py also decodes a serial reference call after the timed step
    with open(os.path.join(d, "pmc_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if "conv_img" in k:
                n_dec += 1
            if "conv_mfma_f16x3_kernel" in k:
                conv16.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                continue
            if "conv_wino4_f16x3_kernel" in k:
                k = "i2v::conv_wino4_f16x3_kernel [3x3x3 Conv3d, Winograd F(4,3)]"
            elif "conv_wino_f16x3_kernel<3," in k:
                k = "i2v::conv_wino_f16x3_kernel<3, .> [SPADE 3x3 Conv2d, Winograd F(2,3)]"
            elif "conv_wino_f16x3_kernel" in k:
                k = "i2v::conv_wino_f16x3_kernel [3x3x3 Conv3d, Winograd F(2,3)]"
            per[k][0] += 1
            per[k][1] += float(r["Counter_Value"])
    # direct split-fp16 launches of one decoder pass, in dispatch order: per block SPADE conv (3 -> 128) and SPADE
    # gamma/beta (3x3 Conv2d), then conv_0 / conv_1 for the blocks the Winograd tiling does not cover (head_0, g_0, and
    # shapes with odd channel counts).  With 6 blocks: 12 SPADE launches; whatever exceeds 2 per block is 3x3x3.
    conv16.sort()
    # The first two blocks (head_0, g_0: 4x4 and 8x8 maps) run all four of their convs on the direct kernel (SPADE input conv,
    # SPADE gamma|beta conv, conv_0, conv_1); every later block only its SPADE input conv (3 -> 128) and, where the 1x3x3
    # Winograd variant does not apply (maps below 32x32), its gamma|beta conv.  Count from the front: 4 + 4, the rest SPADE.
    per_pass = len(conv16) // max(n_dec, 1) if n_dec and len(conv16) % max(n_dec, 1) == 0 else len(conv16)
    for i0, (_, v) in enumerate(conv16):
        i = i0 % per_pass
        name = "i2v::conv_mfma_f16x3_kernel [3x3x3 Conv3d, direct]" if i < 8 and i % 4 >= 2 else "i2v::conv_mfma_f16x3_kernel [SPADE 3x3 Conv2d]"
        per[name][0] += 1
        per[name][1] += v
    return per, " ".join(cmd)


def calibrate(outdir):
    """FETCH_SIZE against known byte counts (tools/fetch_calib.hip): bytes requested / (FETCH_SIZE * 1024) per access pattern."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "tools", "fetch_calib")
    d = os.path.join(outdir, "fetch_calib")
    out = subprocess.run(["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", exe],
                         check=True, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp")).stdout
    known = {ln.split()[0]: float(ln.split()[1]) for ln in out.splitlines() if ln.startswith("calib_")}
    res = {}
    with open(os.path.join(d, "pmc_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if r["Counter_Name"] == "FETCH_SIZE" and k in known:
                res[k] = {"bytes_requested": known[k], "FETCH_SIZE_KiB": float(r["Counter_Value"]),
                          "bytes_per_counted_byte": known[k] / (float(r["Counter_Value"]) * 1024.0)}
    return res


def main():
    outdir = sys.argv[1]
    if "--calibrate" in sys.argv[2:]:
        os.makedirs(outdir, exist_ok=True)
        res = calibrate(outdir)
        with open(os.path.join(outdir, "fetch_calibration.json"), "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
        return
    extra = [a for a in sys.argv[2:] if a != "--parse-only"]
    parse_only = "--parse-only" in sys.argv[2:]  # re-derive the JSON from the CSVs of an earlier run
    os.makedirs(outdir, exist_ok=True)
    rd, cmd = one_pass(outdir, "FETCH_SIZE", extra, parse_only)
    wr, _ = one_pass(outdir, "WRITE_SIZE", extra, parse_only)
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        n = max(rd[k][0], wr[k][0])
        rb, wb = 2.0 * rd[k][1] * 1024.0, wr[k][1] * 1024.0
        kernels[k] = {"launches": n, "read_bytes": rb, "write_bytes": wb, "hbm_bytes_per_launch": (rb + wb) / max(n, 1)}
    dom = {k: v for k, v in kernels.items() if "[3x3x3 Conv3d" in k}
    n = sum(v["launches"] for v in dom.values())
    tot = sum(v["read_bytes"] + v["write_bytes"] for v in dom.values())
    out = {"command": cmd + "   (second pass: --pmc WRITE_SIZE)",
           "units": "read bytes = 2 * FETCH_SIZE[KiB] * 1024 (gfx950 correction, MI355X_MICROARCH.md HBM section); "
                    "write bytes = WRITE_SIZE[KiB] * 1024",
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))),
           "dominant_kernel": {"name": "the 3x3x3 Conv3d launches: conv_wino4_f16x3_kernel (F(4,3): g_2..g_4), conv_wino_f16x3_kernel (F(2,3): g_1) "
                                       "+ conv_mfma_f16x3_kernel (head_0, g_0)", "launches": n,
                               "hbm_bytes_per_launch": tot / max(n, 1)}}
    with open(os.path.join(outdir, "hbm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["dominant_kernel"]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""HBM bytes per launch of every kernel of one bench step, from rocprofv3 PMC counters.

Run on the GPU box:   python tools/pmc_hbm_traffic.py gpurun_out/pmc_traffic [--config land128]
Two SEPARATE counter passes (FETCH_SIZE, then WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes;
FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read, so
read bytes = 2 * FETCH_SIZE * 1024 (the guide's gfx950 correction); WRITE_SIZE is used as reported.
Writes <outdir>/hbm_traffic.json; copy it to profiles/ (bench.py reads profiles/r01_l_pmc_hbm_traffic.json for `roofline.traffic`)."""
import collections
import csv
import json
import os
import subprocess
import sys


def one_pass(outdir, counter, extra, parse_only):
    d = os.path.join(outdir, counter.lower())
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + extra
    if not parse_only:
        env = dict(os.environ, TMPDIR="/tmp")
        with open(os.path.join(outdir, counter.lower() + ".log"), "w") as log:
            subprocess.run(cmd, check=True, stdout=log, stderr=subprocess.STDOUT, env=env)
    per = collections.defaultdict(lambda: [0, 0.0])
    conv16 = []
    with open(os.path.join(d, "pmc_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "conv_mfma_f16x3_kernel" in k:
                conv16.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                continue
            per[k][0] += 1
            per[k][1] += float(r["Counter_Value"])
    # every generator block launches the split-fp16 kernel three times, in this order: SPADE gamma/beta (3x3 Conv2d),
    # conv_0, conv_1 (3x3x3 Conv3d) -- separate the dominant 3x3x3 launches from the small 2-D ones by dispatch order
    conv16.sort()
    assert len(conv16) % 3 == 0, len(conv16)
    for i, (_, v) in enumerate(conv16):
        k = "i2v::conv_mfma_f16x3_kernel [SPADE gamma/beta 3x3 Conv2d]" if i % 3 == 0 else "i2v::conv_mfma_f16x3_kernel [3x3x3 Conv3d]"
        per[k][0] += 1
        per[k][1] += v
    return per, " ".join(cmd)


def main():
    outdir = sys.argv[1]
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
    dom = {k: v for k, v in kernels.items() if "[3x3x3 Conv3d]" in k}
    n = sum(v["launches"] for v in dom.values())
    tot = sum(v["read_bytes"] + v["write_bytes"] for v in dom.values())
    out = {"command": cmd + "   (second pass: --pmc WRITE_SIZE)",
           "units": "read bytes = 2 * FETCH_SIZE[KiB] * 1024 (gfx950 correction, MI355X_MICROARCH.md HBM section); "
                    "write bytes = WRITE_SIZE[KiB] * 1024",
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))),
           "dominant_kernel": {"name": "conv_mfma_f16x3_kernel, the 3x3x3 Conv3d launches (all tile variants)", "launches": n,
                               "hbm_bytes_per_launch": tot / max(n, 1)}}
    with open(os.path.join(outdir, "hbm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["dominant_kernel"]))


if __name__ == "__main__":
    main()

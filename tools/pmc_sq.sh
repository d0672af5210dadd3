#!/bin/bash
# SQ counter picture of one command (two passes of 8 counters, --kernel-trace only).  usage: pmc_sq.sh <outdir> <cmd...>
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/p1 -o pmc -- "$@" > $out/p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/p2 -o pmc -- "$@" > $out/p2.log 2>&1
# third pass (round 6): instruction counts per kind -- does VALU work of one wave run underneath another wave's MFMAs on the same SIMD?
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/p3 -o pmc -- "$@" > $out/p3.log 2>&1 || \
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/p3 -o pmc -- "$@" > $out/p3.log 2>&1
python3 - $out <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = agg[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"{out}/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k][0] += 1; dur[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, cs in agg.items():
    v = {c: s / n for c, (n, s) in cs.items()}
    print(k)
    if k in dur and dur[k][0] and "GRBM_GUI_ACTIVE" in v:
        ns = dur[k][1] / dur[k][0]
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs
        print(f"   -> {ns / 1e3:.1f} us per launch under the counters, clock {v['GRBM_GUI_ACTIVE'] / 8 / ns:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)")
    for c in sorted(v): print(f"   {c:32s} {v[c]:.4g}")
    if "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print(f"   -> wave time: waitcnt/barrier {v['SQ_WAIT_ANY']/wc:.2f}  issue-stall {v['SQ_WAIT_INST_ANY']/wc:.2f}  issuing {v['SQ_ACTIVE_INST_ANY']/wc:.2f}")
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs of the XCD the counters are read from (32 CUs x 4 SIMDs),
        # GRBM_GUI_ACTIVE is the kernel's active clock count: busy fraction of ONE matrix pipe = sum / (clocks x 32 x 4)
        if "GRBM_GUI_ACTIVE" in v: print(f"   -> MFMA pipe busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] * 32 * 4):.3f} (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 32 CUs x 4 SIMDs))")
    if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v and v["SQ_WAVES"]:
        print(f"   -> per wave: {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f} VALU instructions" + (f", {v['SQ_INSTS_MFMA'] / v['SQ_WAVES']:.0f} MFMA" if 'SQ_INSTS_MFMA' in v else ""))
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # quad-cycles summed over the XCD's SIMDs, like SQ_ACTIVE_INST_ANY: fraction of the kernel's time a SIMD's VALU port is issuing
        print(f"   -> VALU issue busy {4 * v['SQ_ACTIVE_INST_VALU'] / (v['GRBM_GUI_ACTIVE'] * 32 * 4):.3f} (4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE x 32 CUs x 4 SIMDs))")
    if "SQ_LDS_IDX_ACTIVE" in v and "GRBM_GUI_ACTIVE" in v:
        print(f"   -> LDS bank-conflict cycles / LDS active: {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.3f}; LDS latency {v['SQ_INST_LEVEL_LDS']/max(v['SQ_ACTIVE_INST_LDS'],1):.1f}; wait_inst_lds {v['SQ_WAIT_INST_LDS']:.3g}")
PY

#!/bin/bash
# Memory-side counter picture of one command (separate passes, --kernel-trace only).  usage: pmc_mem.sh <outdir> <cmd...>
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/m$i -o pmc -- "$@" > $out/m$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $out/m$i.log)"
done
python3 - $out <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{out}/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rocclr" in k: continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    v = {c: s / n for c, (n, s) in cs.items()}
    print(k)
    for c in sorted(v): print(f"   {c:36s} {v[c]:.4g}")
    if "FETCH_SIZE" in v: print(f"   -> HBM read {2*v['FETCH_SIZE']*1024/1e6:.1f} MB (gfx950 x2 correction), write {v.get('WRITE_SIZE',0)*1024/1e6:.1f} MB per launch")
    if "TCC_HIT_sum" in v: print(f"   -> L2 hit rate {v['TCC_HIT_sum']/(v['TCC_HIT_sum']+v['TCC_MISS_sum']):.3f}")
PY

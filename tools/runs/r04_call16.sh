#!/bin/bash
# round 4, GPU call 16: L2 hit rate of the F(4,3) kernel under brick orders 0 and 2 (TCC_HIT / TCC_MISS, --pmc with --kernel-trace only)
export TMPDIR=/tmp
O=gpurun_out/r04o
mkdir -p $O
for ord in 0 2; do
  I2V_W4_ORDER=$ord timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p$ord -o pmc -- tools/conv16w_check 32 16 64 64 128 128 0 1 > $O/p$ord.log 2>&1
  python3 - $O/p$ord $ord <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "F(4,3)" if "conv_wino4" in r["Kernel_Name"] else "F(2,3)" if "conv_wino_" in r["Kernel_Name"] else "direct" if "conv_mfma" in r["Kernel_Name"] else None
        if k:
            a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    v = {c: s / n for c, (n, s) in cs.items()}
    hit = v.get("TCC_HIT_sum", 0); miss = v.get("TCC_MISS_sum", 0)
    print(f"order {sys.argv[2]} {k}: L2 requests {v.get('TCC_REQ_sum', 0):.3g}, hits {hit:.3g}, misses {miss:.3g} -> hit rate {hit / max(hit + miss, 1):.3f}; fabric read requests {v.get('TCC_EA0_RDREQ_sum', 0):.3g}")
PY
  rm -rf $O/p$ord
done

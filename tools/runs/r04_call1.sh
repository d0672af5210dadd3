#!/bin/bash
# round 4, GPU call 1: full GPU test suite, the new bench line, kernel-level A/B of the brick order / channel-tile width,
# g_1 shapes on F(4,3), tap timing at one row block per weight fragment, cINN launch timeline
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log ) 
( timeout 400 python bench.py --per-layer $O/per_layer_bair64.csv 2>$O/bench_bair64.err | tail -1 > $O/bench_bair64.json )
# brick order A/B on three shapes (B = 8 and B = 32), time only
for ord in 0 1 2; do
  for s in "8 16 64 64 128 128 0 1" "8 16 64 64 256 128 1 0" "8 16 64 64 64 64 0 1" "32 16 64 64 128 128 0 1" "8 8 32 32 256 256 0 1"; do
    echo "== order $ord shape $s" >> $O/order_ab.txt
    I2V_W4_ORDER=$ord timeout 120 tools/conv16w_check $s 2>&1 | grep -E "F\(4,3\) " >> $O/order_ab.txt
  done
done
# HBM reads per launch of the F(4,3) kernel under each order (FETCH_SIZE pass only; B = 32 so that the L2s turn over as in the bench)
for ord in 0 1 2; do
  I2V_W4_ORDER=$ord timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_ord$ord -o pmc -- tools/conv16w_check 32 16 64 64 128 128 0 1 > $O/pmc_ord$ord.log 2>&1
  python3 - $O/pmc_ord$ord $ord >> $O/order_fetch.txt <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "conv_wino4" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print(f"order {sys.argv[2]}: conv_wino4 launches {n}, FETCH_SIZE mean {tot / max(n, 1):.0f} KiB -> read {2 * 1024 * tot / max(n, 1) / 1e9:.3f} GB per launch (x2 correction); V operand of this shape = {32*16*64*64*128*6/1e9:.3f} GB")
PY
  rm -rf $O/pmc_ord$ord
done
# g_1 shapes (16x16 maps) on F(4,3): 64- and 32-channel workgroups, B = 8 and B = 64
for bn in 64 32; do
  for s in "8 4 16 16 512 512 0 1" "8 4 16 16 1024 512 1 0" "64 4 16 16 512 512 0 1" "64 4 16 16 1024 512 1 0"; do
    echo "== BN $bn shape $s" >> $O/g1_f43.txt
    I2V_W4_BN=$bn timeout 120 tools/conv16w_check $s 2>&1 | grep -E "direct |F\(2,3\) |F\(4,3\) " >> $O/g1_f43.txt
  done
done
# cycles per MFMA against row blocks per weight fragment: BN = 64 (pass A 4, pass B 2) and BN = 32 (pass A 2, pass B 1)
for bn in 64 32; do
  echo "== BN $bn" >> $O/taptime_rowblocks.txt
  I2V_W4_BN=$bn timeout 120 tools/conv16w_check_tt 8 16 64 64 128 128 0 1 2>&1 | grep -v "^$" >> $O/taptime_rowblocks.txt
done
timeout 120 tools/conv16w_check_tl 8 16 64 64 128 128 0 1 2>&1 | grep -v "^$" > $O/f43_timeline_base.txt
timeout 120 tools/conv16w_check_tl 8 16 64 64 64 64 0 1 2>&1 | grep -v "^$" >> $O/f43_timeline_base.txt
# cINN launch timeline
timeout 300 python tools/flow_timeline.py > $O/flow_launch_timeline.txt 2>&1
# whole-step effect of the orders and of g_1 on F(4,3)
for ord in 1 2; do
  I2V_W4_ORDER=$ord timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --per-layer $O/per_layer_order$ord.csv 2>/dev/null | tail -1 > $O/bench_order$ord.json
done
I2V_DEC_WINO4=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --per-layer $O/per_layer_wino4_2.csv 2>/dev/null | tail -1 > $O/bench_wino4_2.json
tail -3 $O/pytest.log; cut -c1-400 $O/bench_bair64.json; cat $O/order_fetch.txt

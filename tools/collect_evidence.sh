# Re-creates the per-round evidence files (run on the GPU box: bash tools/collect_evidence.sh <outdir>); copy what is to be
# judged into profiles/ afterwards.
export TMPDIR=/tmp
out=${1:-gpurun_out/evidence}
mkdir -p $out
python bench.py 2>$out/bench_bair64.err | tail -1 > $out/bench_bair64.json
sleep 20
python bench.py --config land128 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_land128_b32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bair -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/prof_bair.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_land -o bench -- python bench.py --config land128 --steps 3 --warmup 1 --no-cpu-baseline > $out/prof_land.log 2>&1
rm -f $out/prof_*/bench_kernel_trace.csv
python bench.py --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_bair64_b8.json
python bench.py --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_bair64_b4.json
python bench.py --config land128 --batch 16 --vid-length 32 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_128_b16_t32.json

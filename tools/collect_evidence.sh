export TMPDIR=/tmp
mkdir -p gpurun_out/r01f
python bench.py 2>gpurun_out/r01f/bench_bair64.err | tail -1 > gpurun_out/r01f/bench_bair64.json
python bench.py --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01f/bench_bair64_b4.json
python bench.py --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01f/bench_bair64_b8.json
python bench.py --config land128 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01f/bench_land128_b32.json
python bench.py --config land128 --batch 16 --vid-length 32 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01f/bench_128_b16_t32.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/prof_bair -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01f/prof_bair.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/prof_land -o bench -- python bench.py --config land128 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01f/prof_land.log 2>&1
rm -f gpurun_out/r01f/prof_*/bench_kernel_trace.csv
ls -R gpurun_out/r01f | head -30

# per-kernel durations of the cINN chain (run on the GPU box): bash tools/flow_prof.sh <tag> [env assignments...]
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/flowprof_$tag
mkdir -p $out
( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o flow -- python $GRAFT_REPO_ROOT/tools/flowtime.py > $out/log.txt 2>&1 )
find $out -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/flowprof_$tag.csv \;
find $out -name "*kernel_trace.csv" -delete
head -8 $GRAFT_REPO_ROOT/gpurun_out/flowprof_$tag.csv | cut -d, -f1-8

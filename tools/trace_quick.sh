# quick kernel trace of the headline iteration on one GPU box:   gpurun -- "bash tools/trace_quick.sh <tag> [bench flags]"   -> gpurun_out/tq/<tag>_stats.txt
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-t}; shift || true
O=$R/gpurun_out/tq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -o b -- python3 $R/bench.py --no-cpu-baseline --no-multi-stream-region --steps 7 --warmup 3 "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd $R
python3 tools/kernel_stats.py $(find $O/trace_$TAG -name "*kernel_trace.csv" | head -n 1) 10 "trace_quick $TAG $*" > $O/${TAG}_stats.txt
python3 tools/gpu_busy.py $(find $O/trace_$TAG -name "*kernel_trace.csv" | head -n 1) 10 3 --gaps > $O/${TAG}_busy.txt 2>&1 || true
rm -rf $O/trace_$TAG
head -n 3 $O/${TAG}_stats.txt

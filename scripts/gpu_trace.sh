#!/bin/bash
# kernel trace + stats of the driver's command (no counters): scripts/gpu_trace.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${TAG}_bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_EXTRA} > gpurun_out/prof/${TAG}_bench_stdout.log 2>&1
tail -1 gpurun_out/prof/${TAG}_bench_stdout.log | cut -c1-200
find gpurun_out/prof -name "${TAG}_bench_kernel_trace.csv" -delete
F=$(find gpurun_out/prof -name "${TAG}_bench_kernel_stats.csv" | head -1)
head -24 $F | cut -c1-150

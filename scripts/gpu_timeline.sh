#!/bin/bash
# one Newton iteration launch by launch + the gap table from a 6-step kernel trace: scripts/gpu_timeline.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-tl}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o ${TAG}_tl -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
F=$(find gpurun_out/prof -name "${TAG}_tl_kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $F -120 full > gpurun_out/prof/${TAG}_iteration_timeline.txt 2>&1
python scripts/trace_gaps.py $F > gpurun_out/prof/${TAG}_gap_table.txt 2>&1
rm -f $F
head -3 gpurun_out/prof/${TAG}_iteration_timeline.txt; head -8 gpurun_out/prof/${TAG}_gap_table.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
S=${S:-2}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o r05c_multi -- python bench.py --multi-only $S --single-value 425000 --steps 3 --warmup 1 ${EXTRA} > gpurun_out/prof/r05c_multi_stdout.log 2>&1
tail -1 gpurun_out/prof/r05c_multi_stdout.log | cut -c1-300
F=$(find gpurun_out/prof -name "r05c_multi_kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $F -60 full > gpurun_out/prof/r05c_multi_timeline_S$S.txt
head -60 gpurun_out/prof/r05c_multi_timeline_S$S.txt
rm -f $F

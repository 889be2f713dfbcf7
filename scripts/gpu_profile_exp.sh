#!/bin/bash
# rocprofv3 kernel summary of scripts/exp_scale.py (args passed through)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o exp -- python scripts/exp_scale.py "$@" > gpurun_out/prof2/stdout.log 2>&1
grep N= gpurun_out/prof2/stdout.log; rm -f gpurun_out/prof2/*kernel_trace.csv

#!/bin/bash
# rocprofv3 kernel-trace summary of the bench workload (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof/bench_stdout.log 2>&1
tail -2 gpurun_out/prof/bench_stdout.log
rm -f gpurun_out/prof/*kernel_trace.csv; ls gpurun_out/prof

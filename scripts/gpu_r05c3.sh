#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
for S in 1 2; do
timeout 900 python bench.py --multi-only $S --single-value 425000 --steps 3 --warmup 1 --param verbose=3 > gpurun_out/r05c/multi_v$S.json 2> gpurun_out/r05c/multi_v$S.err
grep "group step" gpurun_out/r05c/multi_v$S.err | tail -2
done
grep "merged level" gpurun_out/r05c/multi_v2.err | tail -22 | cut -c1-200
cat gpurun_out/r05c/multi_v1.json | cut -c1-300

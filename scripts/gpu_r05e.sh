#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_spd_project.py -m gpu -x -q 2>&1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r05e_bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof/r05e_bench_stdout.log 2>&1
tail -1 gpurun_out/prof/r05e_bench_stdout.log | cut -c1-200
find gpurun_out/prof -name "r05e_bench_kernel_trace.csv" -delete
F=$(find gpurun_out/prof -name "r05e_bench_kernel_stats.csv" | head -1)
grep -E "k_tet_hess|k_contact_assemble_coop|k_cloth_gather|k_mask_matrix" $F | cut -c1-160
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver', d['value'], d['ms_per_step'])"

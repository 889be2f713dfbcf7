#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
for S in 2 4 6; do
  timeout 900 python bench.py --multi-only $S --single-value 425000 --steps 6 --warmup 2 > gpurun_out/r05c/multi_$S.json 2> gpurun_out/r05c/multi_$S.err
  python -c "
import json
d = json.loads(open('gpurun_out/r05c/multi_$S.json').read().strip().splitlines()[-1])
print('S=$S', d.get('ms_per_lock_step'), d.get('speedup_vs_single_scene'), d.get('error'), d.get('solves_unconverged'))
"
done
timeout 600 python -m pytest tests/test_gpu_group.py -m gpu -x -q 2>&1 | tail -3

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 600 python -m pytest tests/test_gpu_group.py -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single 10+3', d['value'], d['ms_per_step'])"
for S in 2 4; do
  timeout 900 python bench.py --multi-only $S --single-value 441000 --steps 10 --warmup 3 > gpurun_out/r05c/multi_${S}.json 2> gpurun_out/r05c/multi_${S}.err
  python -c "
import json
d = json.loads(open('gpurun_out/r05c/multi_${S}.json').read().strip().splitlines()[-1])
print('S=$S', d.get('ms_per_lock_step'), d.get('value'), d.get('speedup_vs_single_scene'), d.get('error'), d.get('solves_unconverged'), d.get('group'))
"
done

#!/bin/bash
# A/B of one tsl_set_param switch on the bench workloads (run on the GPU box through gpurun): scripts/gpu_ab.sh <param> [workload] [reps]
cd $GRAFT_REPO_ROOT
P=$1; WL=${2:-cfg4}; REPS=${3:-2}
for i in $(seq $REPS); do
  for v in 0 1; do
    python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --param $P=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$P=$v', '$WL', round(d['value']), 'el-steps/s', round(d['ms_per_step']), 'ms/step  fwd its', round(c['pcg_iters_per_fwd_solve'],1), 'adj its', c['pcg_iters_per_adjoint_solve'], 'fallbacks', c['solver_fallbacks'], 'K1 us', round(d['roofline']['avg_launch_us'],2))"
  done
done

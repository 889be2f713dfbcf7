#!/bin/bash
# bench lines for a list of "--param" settings (run on the GPU box through gpurun): scripts/gpu_try.sh <workload> "<k=v[,k=v]>" ...
cd $GRAFT_REPO_ROOT
WL=$1; shift
for S in "$@"; do
  ARGS=""
  if [ "$S" != "default" ]; then for kv in ${S//,/ }; do ARGS="$ARGS --param $kv"; done; fi
  python bench.py --workload $WL --steps ${STEPS:-2} --warmup ${WARMUP:-1} --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$S', '$WL', round(d['value']), 'el-steps/s', round(d['ms_per_step']), 'ms/step  newton', c['newton_iters_per_step'], 'fwd its', round(c['pcg_iters_per_fwd_solve'],1), 'adj its', c['pcg_iters_per_adjoint_solve'], 'fallbacks', c['solver_fallbacks'])"
done

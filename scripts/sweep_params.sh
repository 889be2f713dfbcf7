#!/bin/bash
# same-box sweep of the solver's tuning keys on the default bench line (10 + 3 steps): scripts/sweep_params.sh
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step' % d['ms_per_step'], d['config'].get('solves_unconverged'))"; }
echo -n "default: "; run
for kv in direct_leaf=32 direct_leaf=48 direct_leaf=80 direct_leaf=96 direct_leaf=128 direct_small_rounds=1 direct_small_rounds=3 direct_g32_below=0 direct_g32_below=600 direct_g32_below=2000 \
          direct_gemv_wide_below=0 direct_gemv_wide_below=150 direct_gemv_wide_below=600 direct_xcd=16 direct_xcd=256 direct_flow=1 direct_prezero=0 tet_warm=0; do
  echo -n "$kv: "; run --param $kv
done
echo -n "default: "; run

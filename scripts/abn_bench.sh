#!/bin/bash
# same-box comparison of N builds of the library (scripts/abn_bench.sh <reps> <a.so> <b.so> ...): the bench line of each (STEPS + WARMUP steps, default 10 + 3), in turn
R=$1; shift
L=thinshelllab_amd/lib/libtsl_hip.so
cp $L /tmp/keep.so
run() { python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step  %.0f el-steps/s' % (d['ms_per_step'], d['value']))"; }
for r in $(seq 1 $R); do
  for f in "$@"; do cp $f $L; echo -n "$(basename $f): "; run; done
done
cp /tmp/keep.so $L

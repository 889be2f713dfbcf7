#!/bin/bash
# same-box A/B of two builds of the library (scripts/ab_bench.sh <a.so> <b.so> [reps]): the default bench line of each, alternating
A=$1; B=$2; R=${3:-3}
L=thinshelllab_amd/lib/libtsl_hip.so
cp $L /tmp/keep.so
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step  %.0f el-steps/s' % (d['ms_per_step'], d['value']))"; }
for r in $(seq 1 $R); do
  cp $A $L; echo -n "A: "; run
  cp $B $L; echo -n "B: "; run
done
cp /tmp/keep.so $L

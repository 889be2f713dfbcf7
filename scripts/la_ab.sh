cd $GRAFT_REPO_ROOT
# same-box comparison of "direct_lookahead" settings: LA_VALUES="0 103" [WORKLOAD=cfg3] bash scripts/la_ab.sh
run() { python bench.py --workload ${WORKLOAD:-cfg4} --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step  %.0f el-steps/s  lost dataflow launches %s' % (d['ms_per_step'], d['value'], d['config'].get('dataflow_launches', {}).get('lost')))"; }
for r in 1 2; do
for v in $LA_VALUES; do echo -n "lookahead $v: "; run --param direct_lookahead=$v; done
done

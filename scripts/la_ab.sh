cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step  %.0f el-steps/s' % (d['ms_per_step'], d['value']))"; }
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -5
for r in 1 2; do
for v in $LA_VALUES; do echo -n "lookahead $v: "; run --param direct_lookahead=$v; done
done

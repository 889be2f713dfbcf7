#!/bin/bash
# round 5, first measurement: the backward-error stop rule of the first refinement pass, A/B on the driver's command + the parity tests it touches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
STEPS=4 python scripts/probe_berr.py > gpurun_out/r05a/probe_berr2.txt 2>&1; tail -12 gpurun_out/r05a/probe_berr2.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05a/bench_berr_on.json 2> gpurun_out/r05a/bench_berr_on.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --param direct_berr=0 > gpurun_out/r05a/bench_berr_off.json 2> gpurun_out/r05a/bench_berr_off.err
for f in on off; do python - <<PY
import json
d = json.loads(open("gpurun_out/r05a/bench_berr_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d.get("convergence"))
PY
done
python -m pytest tests -m gpu -x -q -k "direct_parity or reference_state or determinism or fullsize" 2>&1 | tail -15

#!/bin/bash
# Probe (not part of the product): the driver's bench command N times with verbose=1; the rollout is chaotic (atomic summation
# order -> different contact sets from step ~5 on), so rare hard systems only show up over many runs.  Collects the lines of
# solves that did not converge and the convergence fields of every bench line.
N=${1:-8}
mkdir -p gpurun_out/flake
for r in $(seq 1 $N); do
  TSL_PARAMS=verbose=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $FLAKE_ARGS > gpurun_out/flake/run_$r.json 2> gpurun_out/flake/run_$r.err
  grep -E "did not converge|perturbed|fallback|unconverged" gpurun_out/flake/run_$r.err | head -20 > gpurun_out/flake/run_$r.msgs
  python - <<PY
import json
d = json.loads(open("gpurun_out/flake/run_$r.json").read().strip().splitlines()[-1]); c = d["config"]
print("run $r: ms/step %.1f unconverged %d attained %s max_rel_residual_fwd %.1e adj %.1e ls %.1f" % (d["ms_per_step"], c["solves_unconverged"], c["solves_accepted_at_attainable_accuracy"], c["max_rel_residual_fwd"], c["max_rel_residual_adjoint"], c["line_search_evals_per_step"]))
PY
  rm -f gpurun_out/flake/run_$r.err
done

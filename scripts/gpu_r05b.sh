#!/bin/bash
# round 5: the super-tile dataflow kernel -- direct-solver tests first (under a timeout: a lost flag costs seconds, not the box), then bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_direct.py -m gpu -x -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -x -q -k "not golden" 2>&1 | tail -15
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r05b/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); r = d["roofline"]; print({k: r[k] for k in r if k != "whole_step"})
PY
tail -5 gpurun_out/r05b/bench.err

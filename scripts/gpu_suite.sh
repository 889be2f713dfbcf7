#!/bin/bash
# the whole GPU suite + smoke; golden bit vectors regenerated first when REGEN=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
if [ "$REGEN" = "1" ]; then python tests/golden/gen_golden_gpu.py > gpurun_out/suite/gen_golden.log 2>&1; tail -3 gpurun_out/suite/gen_golden.log; cp tests/golden/det_*.npz gpurun_out/suite/; fi
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3

#!/bin/bash
# SQ counters (one pass, no tracing) of the factorisation kernels on a short bench run: where the wave cycles of the GEMMs and of
# the block-step kernel go (MFMA busy, waiting, issue stalls, LDS conflicts).  usage (GPU box): bash scripts/gpu_pmc_sq.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
mkdir -p gpurun_out/prof
for KRN in "k_ds_gemm(_x)?<1" "k_ds_gemm(_x)?<0" "k_ds_gj_step" "k_ds_gj_flow" "k_ds_inv_small"; do
  KN=$(echo $KRN | tr -d '<>()?' | sed s/_x//)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-include-regex "$KRN" --output-format csv -d gpurun_out/prof -o ${TAG}_sq_${KN} -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/${TAG}_sq_stdout.log 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/prof/**/${TAG}_sq_${KN}_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    lines = [f"kernel=$KRN dispatches={max(n.values()) if n else 0}"] + [f"{k} sum={v:.6g} mean_per_dispatch={v / n[k]:.6g}" for k, v in sorted(acc.items())]
    open("gpurun_out/prof/${TAG}_sq_${KN}_summary.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    break
PY
  find gpurun_out/prof -name "*counter_collection.csv" -delete
done

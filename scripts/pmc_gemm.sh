#!/bin/bash
# HBM-side counters of the GEMM kernels of the direct solve for two settings of "direct_xcd" (separate --pmc passes, no tracing)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmcx
for X in 0 64; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CNT --kernel-include-regex "k_ds_gemm" --output-format csv -d gpurun_out/pmcx -o x${X}_$CNT -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --param direct_xcd=$X > gpurun_out/pmcx/stdout.log 2>&1
    python - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcx/**/x${X}_${CNT}_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "$CNT": agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(f"direct_xcd=$X $CNT {k}: dispatches {len(v)} mean {sum(v)/len(v):.1f} KB sum {sum(v)/1e6:.2f} GB")
PY
    find gpurun_out/pmcx -name "*counter_collection.csv" -delete
  done
done

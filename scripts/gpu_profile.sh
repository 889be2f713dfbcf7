#!/bin/bash
# rocprofv3 summaries of the bench workload (run on the GPU box through gpurun): kernel trace + stats, then the two
# HBM counters of the PCG SpMV kernel in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01c}
ARGS=${BENCH_ARGS:---steps 2 --warmup 2 --no-cpu-baseline}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${TAG}_bench -- python bench.py $ARGS > gpurun_out/prof/${TAG}_bench_stdout.log 2>&1
tail -1 gpurun_out/prof/${TAG}_bench_stdout.log
rm -f gpurun_out/prof/*kernel_trace.csv
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-include-regex "k_pcg_spmv" --output-format csv -d gpurun_out/prof -o ${TAG}_pmc_$CNT -- python bench.py $ARGS > gpurun_out/prof/${TAG}_pmc_${CNT}_stdout.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof/**/${TAG}_pmc_${CNT}_counter_collection.csv", recursive=True) + glob.glob("gpurun_out/prof/${TAG}_pmc_${CNT}_counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    vals = [float(r["Counter_Value"]) for r in rows if r.get("Counter_Name") == "$CNT"]
    if vals:
        print("$CNT", "dispatches", len(vals), "mean", sum(vals) / len(vals), "min", min(vals), "max", max(vals))
        open("gpurun_out/prof/${TAG}_pmc_${CNT}_summary.txt", "w").write(f"$CNT kernel=k_pcg_spmv dispatches={len(vals)} mean={sum(vals)/len(vals)} min={min(vals)} max={max(vals)}\n")
    break
PY
done
rm -f gpurun_out/prof/*counter_collection.csv
find gpurun_out/prof -name "*counter_collection.csv" -delete
ls gpurun_out/prof

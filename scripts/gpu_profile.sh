#!/bin/bash
# rocprofv3 summaries of the bench workload (run on the GPU box through gpurun):
#   1. kernel trace + stats of the driver's command (python bench.py --steps 20 --warmup 5), 2. the two HBM counters of the kernels
#   of the direct solve in their own passes on a shorter run (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never
#   combined with tracing), 3. the bench lines with cpu_baseline (default command and the driver's command).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
WL=${WORKLOAD:-cfg4}
ARGS=${BENCH_ARGS:---steps 20 --warmup 5 --no-cpu-baseline}
PMC_ARGS=${PMC_ARGS:---steps 3 --warmup 1 --no-cpu-baseline}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${TAG}_bench -- python bench.py --workload $WL $ARGS > gpurun_out/prof/${TAG}_bench_stdout.log 2>&1
tail -1 gpurun_out/prof/${TAG}_bench_stdout.log | cut -c1-300
rm -f gpurun_out/prof/*kernel_trace.csv
for PAIR in "k_ds_gemm1=k_ds_gemm(_x)?<1" "k_ds_extend_panels=k_ds_extend_panels" "k_ds_gemm0=k_ds_gemm(_x)?<0" "k_ds_gj_flow=k_ds_gj_flow" "k_ds_gemv=k_ds_gemv"; do   # name=regex: the Schur / G GEMMs run as k_ds_gemm<mode, 4> and k_ds_gemm_x<mode, 4>
  KN=${PAIR%%=*}; KRN=${PAIR#*=}
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CNT --kernel-include-regex "$KRN" --output-format csv -d gpurun_out/prof -o ${TAG}_pmc_${KN}_$CNT -- python bench.py --workload $WL $PMC_ARGS > gpurun_out/prof/${TAG}_pmc_stdout.log 2>&1
    python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof/**/${TAG}_pmc_${KN}_${CNT}_counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    vals = [float(r["Counter_Value"]) for r in rows if r.get("Counter_Name") == "$CNT"]
    if vals:
        line = f"$CNT kernel=$KRN dispatches={len(vals)} mean={sum(vals)/len(vals)} min={min(vals)} max={max(vals)} sum={sum(vals)}"
        print(line)
        open("gpurun_out/prof/${TAG}_pmc_${KN}_${CNT}_summary.txt", "w").write(line + "\n")
    break
PY
    find gpurun_out/prof -name "*counter_collection.csv" -delete
  done
done
python bench.py --workload $WL > gpurun_out/prof/${TAG}_full_default.json 2> gpurun_out/prof/${TAG}_full_default.err
python bench.py --workload $WL --steps 20 --warmup 5 > gpurun_out/prof/${TAG}_full_driver.json 2> gpurun_out/prof/${TAG}_full_driver.err
# scene groups next to the headline (S scenes of this GPU in lock step, merged factorisations): the multi_scene object of the same line
for S in 2 4; do
  python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --scenes-per-gpu $S > gpurun_out/prof/${TAG}_multi_$S.json 2> gpurun_out/prof/${TAG}_multi_$S.err
done
ls gpurun_out/prof

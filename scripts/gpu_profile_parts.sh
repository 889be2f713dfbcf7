#!/bin/bash
# The profile set of a round in short gpurun calls (each a few minutes): bash scripts/gpu_profile_parts.sh <tag> <part>
#   trace  rocprofv3 --kernel-trace --stats of the driver's command
#   lines  the bench lines (default command, driver's command with cpu_baseline)
#   pmc    FETCH_SIZE / WRITE_SIZE of the kernels of the direct solve, separate passes, no tracing
#   sq     SQ counters of the GEMMs and the inversion kernels (scripts/gpu_pmc_sq.sh)
#   multi  scene groups next to the headline (--scenes-per-gpu 2 / 4) and the device-clock trace of the root's dataflow chain
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}; PART=${2:-trace}
WL=${WORKLOAD:-cfg4}
mkdir -p gpurun_out/prof
case $PART in
trace)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${TAG}_bench -- python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof/${TAG}_bench_stdout.log 2>&1
  tail -1 gpurun_out/prof/${TAG}_bench_stdout.log | cut -c1-300
  F=$(find gpurun_out/prof -name "${TAG}_bench_kernel_trace.csv" | head -1)
  python scripts/trace_timeline.py $F -400 full > gpurun_out/prof/${TAG}_iteration_timeline.txt 2>&1
  python scripts/trace_gaps.py $F > gpurun_out/prof/${TAG}_gap_table.txt 2>&1
  python scripts/trace_counts.py $F gpurun_out/prof/${TAG}_trace_counts.json
  rm -f $F
  head -14 gpurun_out/prof/${TAG}_iteration_timeline.txt ;;
lines)
  python bench.py --workload $WL > gpurun_out/prof/${TAG}_full_default.json 2> gpurun_out/prof/${TAG}_full_default.err
  python bench.py --workload $WL --steps 20 --warmup 5 > gpurun_out/prof/${TAG}_full_driver.json 2> gpurun_out/prof/${TAG}_full_driver.err
  for f in default driver; do tail -1 gpurun_out/prof/${TAG}_full_$f.json | cut -c1-260; done ;;
pmc)
  for PAIR in "k_ds_gemm1=k_ds_gemm(_x)?<1" "k_ds_extend_panels=k_ds_extend_panels" "k_ds_gemm0=k_ds_gemm(_x)?<0" "k_ds_gj_flow=k_ds_gj_flow" "k_ds_gemv=k_ds_gemv"; do
    KN=${PAIR%%=*}; KRN=${PAIR#*=}
    for CNT in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $CNT --kernel-include-regex "$KRN" --output-format csv -d gpurun_out/prof -o ${TAG}_pmc_${KN}_$CNT -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/${TAG}_pmc_stdout.log 2>&1
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
  done ;;
sq)
  bash scripts/gpu_pmc_sq.sh $TAG ;;
multi)
  for S in 2 4; do
    python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --scenes-per-gpu $S > gpurun_out/prof/${TAG}_multi_$S.json 2> gpurun_out/prof/${TAG}_multi_$S.err
    python -c "
import json
d = json.loads(open('gpurun_out/prof/${TAG}_multi_$S.json').read().strip().splitlines()[-1]); m = d.get('multi_scene', {})
print('S=$S single', d['value'], d['ms_per_step'], '| group', m.get('value'), m.get('speedup_vs_single_scene'), m.get('solves_unconverged'), m.get('error'))"
  done
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o ${TAG}_tl -- python bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  F=$(find gpurun_out/prof -name "${TAG}_tl_kernel_trace.csv" | head -1)
  python scripts/trace_timeline.py $F -120 full > gpurun_out/prof/${TAG}_iteration_timeline.txt 2>&1
  python scripts/trace_gaps.py $F > gpurun_out/prof/${TAG}_gap_table.txt 2>&1
  rm -f $F
  head -3 gpurun_out/prof/${TAG}_iteration_timeline.txt
  python scripts/probe_flow_chain.py > gpurun_out/prof/${TAG}_root_chain_trace.txt 2>&1
  tail -3 gpurun_out/prof/${TAG}_root_chain_trace.txt | cut -c1-120 ;;
esac

"""Copies the rocprofv3 summaries that scripts/gpu_profile.sh left under gpurun_out/prof/ into profiles/ (tracked) and prints
the headline numbers.  Run in the repo root after `gpurun -- 'bash scripts/gpu_profile.sh r01b; BENCH_ARGS="--workload cfg4-scaled
--steps 2 --warmup 2 --no-cpu-baseline" bash scripts/gpu_profile.sh r01bs; python bench.py > gpurun_out/prof/full.json; ...'`."""
import csv, json, os, re, shutil, sys

OUT = sys.argv[1] if len(sys.argv) > 1 else "r01c"  # prefix of the files written under profiles/

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof") + os.sep
dst = os.path.join(root, "profiles") + os.sep


def last_json_line(path):
    return [l for l in open(path) if l.startswith('{"metric"')][-1]


for tag, name in ((OUT, "cfg4"), (OUT + "s", "cfg4_scaled")):
    shutil.copy(src + f"{tag}_bench_kernel_stats.csv", dst + f"{OUT}_{name}_kernel_stats.csv")
    line = last_json_line(src + f"{tag}_bench_stdout.log")
    open(dst + f"{OUT}_{name}_bench.json", "w").write(line)
    j = json.loads(line)
    print(name, "profiled run:", round(j["value"]), "element-steps/s,", round(j["ms_per_step"]), "ms/step,", round(j["config"]["pcg_iters_per_fwd_solve"], 1), "its/solve")
    f = open(src + f"{tag}_pmc_FETCH_SIZE_summary.txt").read().strip()
    w = open(src + f"{tag}_pmc_WRITE_SIZE_summary.txt").read().strip()
    open(dst + f"{OUT}_{name}_pmc_k_pcg_spmv.txt", "w").write(f + "\n" + w + "\n")
    fm = float(re.search(r"mean=([0-9.]+)", f).group(1)); wm = float(re.search(r"mean=([0-9.]+)", w).group(1))
    traffic = int(round((2 * fm + wm) * 1024))
    json.dump({"kernel": "k_pcg_spmv", "workload": name.replace("_", "-"), "counters": {"FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm},
               "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; separate --pmc passes",
               "traffic_bytes_per_launch": traffic}, open(dst + f"{OUT}_{name}_pmc_k_pcg_spmv.json", "w"), indent=1)
    print(name, "HBM traffic per k_pcg_spmv launch:", traffic, "B (FETCH_SIZE", fm, "KB, WRITE_SIZE", wm, "KB)")
for s, d in (("full.json", f"{OUT}_cfg4_bench_full.json"), ("full_scaled.json", f"{OUT}_cfg4_scaled_bench_full.json"), ("full_drape.json", f"{OUT}_drape_bench_full.json")):
    if not os.path.exists(src + s):
        continue
    line = last_json_line(src + s)
    open(dst + d, "w").write(line)
    j = json.loads(line); c = j["config"]; r = j["roofline"]
    print(d, round(j["value"]), round(j["ms_per_step"], 1), "K1 us", round(r["avg_launch_us"], 2), round(r.get("avg_launch_us_device_clock", 0), 2), "frac", round(r["frac"], 3),
          "cpu", j.get("cpu_baseline", {}).get("value"), "its", round(c["pcg_iters_per_fwd_solve"], 1), c["newton_iters_per_step"], round(c["pcg_iters_per_adjoint_solve"]))
for name in ("cfg4", "cfg4_scaled"):
    rows = list(csv.DictReader(open(dst + f"{OUT}_{name}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(name, "total GPU ms", round(tot / 1e6))
    for r in rows[:16]:
        print(f"  {r['Name'][:50]:50s} {int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:8.1f} ms avg {float(r['AverageNs'])/1e3:7.2f} us {float(r['Percentage']):5.2f}%")

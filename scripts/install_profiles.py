"""Copies the rocprofv3 summaries that scripts/gpu_profile.sh left under gpurun_out/prof/ into profiles/ (tracked) and prints the
headline numbers.  usage (repo root, after `gpurun -- 'bash scripts/gpu_profile.sh r02a'`):  python scripts/install_profiles.py r02a [workload]"""
import csv, json, os, re, shutil, sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03a"
WL = (sys.argv[2] if len(sys.argv) > 2 else "cfg4").replace("-", "_")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof") + os.sep
dst = os.path.join(root, "profiles") + os.sep


import subprocess
try:
    COMMIT = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    COMMIT = "unknown"


def last_json_line(path):
    return [l for l in open(path) if l.startswith('{"metric"')][-1]


shutil.copy(src + f"{TAG}_bench_kernel_stats.csv", dst + f"{TAG}_{WL}_kernel_stats.csv")
# the in-situ launch durations bench.py quotes next to its own replays (roofline.avg_launch_us_rocprof): name -> calls, total ns, with the set and commit
with open(dst + f"latest_{WL}_kernel_stats.json", "w") as fh:
    rows_ = list(csv.DictReader(open(dst + f"{TAG}_{WL}_kernel_stats.csv")))
    try:
        n_fact = json.load(open(src + f"{TAG}_trace_counts.json"))["factorisations"]
    except (OSError, KeyError, ValueError):
        n_fact = None
    json.dump({"tag": TAG, "commit": COMMIT, "factorisations": n_fact, "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline",
               "kernels": {r["Name"]: {"calls": int(r["Calls"]), "total_ns": float(r["TotalDurationNs"])} for r in rows_}}, fh, indent=0)
open(dst + f"{TAG}_{WL}_bench_profiled.json", "w").write(last_json_line(src + f"{TAG}_bench_stdout.log"))
for s, d in ((f"{TAG}_full_default.json", f"{TAG}_{WL}_bench_default.json"), (f"{TAG}_full_driver.json", f"{TAG}_{WL}_bench_driver_cmd.json")):
    if os.path.exists(src + s):
        line = last_json_line(src + s)
        open(dst + d, "w").write(line)
        j = json.loads(line); c = j["config"]; r = j["roofline"]
        print(d, round(j["value"]), "el-steps/s", round(j["ms_per_step"], 1), "ms/step; unconverged", c["solves_unconverged"], "fallbacks", c["solver_fallbacks"],
              "| roofline", r["kernel"].split(" ")[0], r["bound"], round(r["achieved"], 1), r["unit"], "frac", round(r["frac"], 3), "| cpu", (j.get("cpu_baseline") or {}).get("value"))
for S in (2, 4):
    if os.path.exists(src + f"{TAG}_multi_{S}.json"):
        try:
            line = last_json_line(src + f"{TAG}_multi_{S}.json")
            open(dst + f"{TAG}_{WL}_bench_scenes_per_gpu_{S}.json", "w").write(line)
            m = json.loads(line).get("multi_scene", {})
            print(f"scenes per GPU {S}:", m.get("value"), "el-steps/s,", m.get("speedup_vs_single_scene"), "x one scene, unconverged", m.get("solves_unconverged"), m.get("error"))
        except Exception as e:
            print("multi", S, "failed:", e)
for kn in ("k_ds_gemm1", "k_ds_extend_panels", "k_ds_gemm0", "k_ds_gj_step", "k_ds_gj_flow", "k_ds_gemv"):
    try:
        f = open(src + f"{TAG}_pmc_{kn}_FETCH_SIZE_summary.txt").read().strip()
        w = open(src + f"{TAG}_pmc_{kn}_WRITE_SIZE_summary.txt").read().strip()
    except OSError:
        continue
    open(dst + f"{TAG}_{WL}_pmc_{kn}.txt", "w").write(f + "\n" + w + "\n")
    fm = float(re.search(r"mean=([0-9.]+)", f).group(1)); wm = float(re.search(r"mean=([0-9.]+)", w).group(1))
    traffic = int(round((2 * fm + wm) * 1024))
    json.dump({"kernel": kn, "workload": WL.replace("_", "-"), "tag": TAG, "commit": COMMIT, "counters": {"FETCH_SIZE_KB_mean_per_dispatch": fm, "WRITE_SIZE_KB_mean_per_dispatch": wm},
               "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; separate --pmc passes, no tracing",
               "traffic_bytes_per_launch": traffic}, open(dst + f"{TAG}_{WL}_pmc_{kn}.json", "w"), indent=1)
    # the name bench.py looks for (latest counters; the JSON carries the profile set and the commit)
    shutil.copy(dst + f"{TAG}_{WL}_pmc_{kn}.json", dst + f"latest_{WL}_pmc_{kn}.json")
    print(kn, "HBM traffic per launch:", traffic, "B (FETCH_SIZE", fm, "KB x2, WRITE_SIZE", wm, "KB)")
for f in os.listdir(src):   # SQ counter summaries of scripts/gpu_pmc_sq.sh
    if f.startswith(f"{TAG}_sq_") and f.endswith("_summary.txt"):
        shutil.copy(src + f, dst + f.replace(f"{TAG}_sq_", f"{TAG}_{WL}_sq_").replace("_summary", ""))
rows = list(csv.DictReader(open(dst + f"{TAG}_{WL}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", round(tot / 1e6))
for r in rows[:16]:
    print(f"  {r['Name'][:58]:58s} {int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:8.1f} ms avg {float(r['AverageNs'])/1e3:7.2f} us {float(r['Percentage']):5.2f}%")

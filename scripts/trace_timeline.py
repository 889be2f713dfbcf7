"""Reads a rocprofv3 kernel-trace CSV and prints the kernel sequence of one Newton iteration late in the run with durations and
gaps, plus busy / idle totals per kernel name over the window between the starts of two consecutive factorisations.
usage: trace_timeline.py <kernel_trace.csv> [which]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_ds_assemble_level") and "k_ds_" not in rows[i - 1]["Kernel_Name"]]   # first launch of a factorisation
a, b = marks[which], marks[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
busy = collections.defaultdict(float); cnt = collections.Counter()
prev_end = None; idle = 0.0
lines = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:40]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    if prev_end is not None and gap > 0: idle += gap
    busy[name] += (e - s) / 1e3; cnt[name] += 1
    lines.append(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  stream {r.get('Stream_Id', r.get('Queue_Id', '?'))}  {name}")
    prev_end = max(prev_end or 0, e)
total = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print(f"window {total:.1f} us, {b - a} launches, idle (no kernel running, single-stream view) {idle:.1f} us")
for n, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  {v:8.1f} us  {cnt[n]:4d} x  {n}")
if len(sys.argv) > 3:
    print("\n".join(lines))

"""Reads a rocprofv3 kernel-trace CSV and writes {"factorisations": N}: the number of factorisations the trace holds (first launch of a factorisation = a
k_ds_assemble_level launch whose predecessor by start time is no kernel of the direct solve; the rule of trace_timeline.py / trace_gaps.py).  bench.py divides the
in-situ totals of a kernel class by it (the look-ahead issues the Schur / G launches of the upper levels in two parts: launch averages no longer compare with the replays).
usage: trace_counts.py <kernel_trace.csv> <out.json>"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = sum(1 for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_ds_assemble_level") and "k_ds_" not in rows[i - 1]["Kernel_Name"])
json.dump({"factorisations": n}, open(sys.argv[2], "w"))
print("factorisations in the trace:", n)

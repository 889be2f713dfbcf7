"""Summaries of a rocprofv3 --kernel-trace run (not part of the product): per-kernel totals and the launch sequence of one factorisation + solve.
usage: prof_summary.py <dir> <prefix> [factorisation index]"""
import csv, sys, re
d, pre = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = list(csv.DictReader(open(f"{d}/{pre}_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU ms", round(tot / 1e6, 1))
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):7d} tot {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f}%")
tr = list(csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if r['Kernel_Name'].startswith('k_ds_assemble_blocks')]
a, b = idx[which], idx[which + 1]
t0 = int(tr[a]['Start_Timestamp'])
prev = None; cnt = 0; totd = 0; start = 0; end = 0
def flush():
    if prev: print(f"{start:9.1f} {prev[0]:22s} {prev[1]:26s} x{cnt:3d} busy {totd:8.1f} us  span {end-start:8.1f} us")
for r in tr[a:b]:
    n = r['Kernel_Name'].split('(')[0][:22]
    key = (n, f"{r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}") if not n.startswith('k_ds_gemv') else (n, '')
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    if key != prev:
        flush(); prev = key; cnt = 0; totd = 0; start = s
    cnt += 1; totd += e - s; end = e
flush()

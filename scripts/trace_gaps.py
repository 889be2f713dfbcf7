"""Reads a rocprofv3 kernel-trace CSV; over the window between the starts of two factorisations `n` apart prints wall time, the union of
the kernels' busy intervals (all streams), the idle remainder, and the idle time grouped by (kernel before the gap -> kernel after it).
usage: trace_gaps.py <kernel_trace.csv> [first] [n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 0: a window in the MIDDLE of the run (the tail of the trace holds bench.py's class replays and parity legs, not time steps)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_ds_assemble_level") and "k_ds_" not in rows[i - 1]["Kernel_Name"]]
if first == 0: first = max(0, len(marks) // 2 - n // 2)
a, b = marks[first], marks[first + n]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
end = t0; endname = "start"; busy = 0
gaps = collections.defaultdict(float); cnt = collections.Counter()
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:36]
    if s > end:
        gaps[(endname, name)] += (s - end) / 1e3; cnt[(endname, name)] += 1
        busy += e - s
    else:
        busy += max(0, e - end)
    if e > end: end, endname = e, name
print(f"{n} factorisation periods: wall {(t1 - t0) / 1e3 / n:.1f} us each, busy (union over streams) {busy / 1e3 / n:.1f} us, idle {(t1 - t0 - busy) / 1e3 / n:.1f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {v / n:7.1f} us per period  {cnt[k] / n:5.1f} x {v / cnt[k]:6.1f} us   {k[0]} -> {k[1]}")
if len(sys.argv) > 4:   # the largest single gaps with their neighbourhood
    big = []
    end = t0
    for i in range(a, b):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        if s > end: big.append((s - end, i))
        end = max(end, e)
    for g, i in sorted(big, reverse=True)[:int(sys.argv[4])]:
        ctx = " | ".join(rows[j]["Kernel_Name"].split("(")[0][:28] for j in range(max(a, i - 3), min(b, i + 3)))
        print(f"gap {g / 1e3:8.1f} us at +{(int(rows[i]['Start_Timestamp']) - t0) / 1e6:8.2f} ms: {ctx}")

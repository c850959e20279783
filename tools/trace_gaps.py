"""Developer tool: from a rocprofv3 kernel trace CSV, the time between the end of each kernel and the start of the next one on the
same queue, by (kernel, next kernel) pair -- what a step loses between its launches.   python tools/trace_gaps.py <kernel_trace.csv> [min_calls]"""
import collections, csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", n).replace("void nl::", "").replace("nl::", "")[:48]
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if gap < 50:
        acc[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(gap)
    dur[short(a["Kernel_Name"])].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print("kernel -> next kernel: calls, median gap us (end -> start), median duration of the first")
for (a, b), g in sorted(acc.items(), key=lambda kv: -len(kv[1])):
    if len(g) >= mn:
        g.sort(); d = sorted(dur[a])
        print(f"  {a:48s} -> {b:48s} {len(g):5d}  gap {g[len(g) // 2]:6.2f}  dur {d[len(d) // 2]:7.2f}")

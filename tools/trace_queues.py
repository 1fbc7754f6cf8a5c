"""Reduce a rocprofv3 kernel trace (csv) of a bench run to per-queue totals: kernels, summed duration, and the union of the busy
intervals (what share of the wall time a queue had a kernel in flight).  Usage: trace_queues.py <kernel_trace.csv> [last_ms]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
t_end = max(int(r["End_Timestamp"]) for r in rows)
t0 = t_end - int(last_ms * 1e6)
q = defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e >= t0:
        q[r["Queue_Id"]].append((max(s, t0), e, r["Kernel_Name"]))
print("window: last %.0f ms of the trace" % last_ms)
for qid, iv in sorted(q.items(), key=lambda kv: -len(kv[1])):
    iv.sort()
    total = sum(e - s for s, e, _ in iv)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += (cur_e - cur_s) if cur_e is not None else 0
    names = defaultdict(float)
    for s, e, n in iv:
        names[n.split("(")[0][-40:]] += (e - s) / 1e6
    top = sorted(names.items(), key=lambda kv: -kv[1])[:5]
    print("queue %s: %6d kernels, sum %.1f ms, busy %.1f ms (%.0f %% of window); top: %s" % (
        qid, len(iv), total / 1e6, busy / 1e6, 100.0 * busy / (last_ms * 1e6), ", ".join("%s %.1f" % kv for kv in top)))

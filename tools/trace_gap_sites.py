"""Where the detector queue idles: per (previous kernel -> next kernel) pair, the summed gap between consecutive kernels of the
busiest queue in a rocprofv3 --kernel-trace CSV.   usage: python3 tools/trace_gap_sites.py t_kernel_trace.csv [passes]"""
import collections
import csv
import sys


def main(path, passes=1):
    byq = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows = sorted(max(byq.values(), key=len))
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    agg = collections.defaultdict(lambda: [0, 0])
    total = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - e0
        if 0 < g < 500000:                                        # longer = between passes
            agg[(short(n0), short(n1))][0] += 1
            agg[(short(n0), short(n1))][1] += g
            total += g
    print("kernels %d, summed gaps (<500 us) %.2f ms = %.2f ms per pass" % (len(rows), total / 1e6, total / 1e6 / passes))
    for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%8.1f us total  x%-4d %6.1f us avg   %s -> %s" % (t / 1e3, c, t / c / 1e3, a, b))




def one_pass(path, which=-2, aggregate=False):
    """The kernels of ONE detector pass (from a stem_pool_kernel to the next), in order: start offset, duration, gap before."""
    byq = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows = sorted(max(byq.values(), key=len))
    marks = [i for i, r in enumerate(rows) if "stem_pool_kernel" in r[2]]
    if which == "best":                                           # the pass with the least idle time = a graph replay
        idle = lambda k: sum(max(rows[i][0] - rows[i - 1][1], 0) for i in range(marks[k] + 1, marks[k + 1]))
        which = min(range(len(marks) - 1), key=idle)
    a, b = marks[which], marks[which + 1]
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    t0 = rows[a][0]
    gaps = 0
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for i in range(a, b):
        s, e, n = rows[i]
        g = s - rows[i - 1][1]
        gaps += max(g, 0) if i > a else 0
        if aggregate:
            r = agg[short(n)]
            r[0] += 1
            r[1] += e - s
            r[2] += max(g, 0) if i > a else 0
        else:
            print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, g / 1e3, short(n)))
    for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("x%-4d dur %8.1f us total %7.1f avg   gap before %7.1f us total   %s" % (c, t / 1e3, t / c / 1e3, g / 1e3, n))
    print("pass: %d kernels, span %.2f ms, busy %.2f ms, gaps %.2f ms" % (
        b - a, (rows[b][0] - t0) / 1e6, sum(e - s for s, e, _ in rows[a:b]) / 1e6, gaps / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] in ("pass", "passagg"):
        w = sys.argv[3] if len(sys.argv) > 3 else "-2"
        one_pass(sys.argv[1], w if w == "best" else int(w), sys.argv[2] == "passagg")
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)

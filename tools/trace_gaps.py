"""Analyse a rocprofv3 --kernel-trace CSV: per queue, busy time, gaps between consecutive kernels, and the longest /
most frequent kernels -- used to see what the tracker stream's serial chain waits for beside a saturated detector."""
import collections
import csv
import sys


def main(path, top=14):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                         r.get("Stream_Id", "?"), r["Kernel_Name"][:70]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print("kernels %d  span %.1f ms" % (len(rows), (t1 - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[(r[2], r[3])].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, *_ in rs)
        gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        small = [g for g in gaps if 0 <= g < 200000]
        print("queue/stream %s: %d kernels, busy %.1f ms, median gap %.1f us, mean gap(<200us) %.1f us, mean dur %.1f us"
              % (q, len(rs), busy / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0,
                 sum(small) / max(len(small), 1) / 1e3, busy / len(rs) / 1e3))
        agg = collections.defaultdict(lambda: [0, 0])
        for s, e, _, _, n in rs:
            agg[n][0] += 1
            agg[n][1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            print("    %-70s x%-5d %8.1f us avg  %7.2f ms" % (n, c, t / c / 1e3, t / 1e6))


def window(path, queue, start_frac=0.8, count=70):
    """Consecutive kernels of one queue: start offset, duration and gap to the previous one (us)."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Queue_Id") == queue:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    rows.sort()
    i0 = int(len(rows) * start_frac)
    base = rows[i0][0]
    for i in range(i0, min(i0 + count, len(rows))):
        s, e, n = rows[i]
        print("%9.1f  dur %7.1f  gap %7.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3, n))


def gaps_by_kernel(path, queue, top=14):
    """Per kernel name: how long the queue sat idle before its launches started (dispatch wait + host cadence)."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Queue_Id") == queue:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    rows.sort()
    agg = collections.defaultdict(list)
    for i in range(1, len(rows)):
        g = rows[i][0] - rows[i - 1][1]
        if g < 2000000:
            agg[rows[i][2]].append(g)
    for n, gs in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        gs.sort()
        print("%-60s x%-5d gap before: mean %7.1f us  median %6.1f  p90 %7.1f  max %8.1f  total %7.2f ms"
              % (n, len(gs), sum(gs) / len(gs) / 1e3, gs[len(gs) // 2] / 1e3, gs[int(len(gs) * 0.9)] / 1e3, gs[-1] / 1e3,
                 sum(gs) / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "gaps":
        gaps_by_kernel(sys.argv[1], sys.argv[2])
    elif len(sys.argv) > 2:
        window(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.8)
    else:
        main(sys.argv[1])

"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes to profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [round tag]

Per MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) under-reports wide coalesced streaming reads by exactly 2x on
gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-byte streaming stores."""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0].replace(" ", "")
            agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1) * 1024.0
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) * 1024.0
        out[k] = {"dispatches": len(fetch.get(k, [])), "fetch_size_bytes_raw": f, "fetch_bytes_x2": 2 * f,
                  "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w, "round": tag}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fjs:
        json.dump(out, fjs, indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
        print("%-50s %8.1f MB/launch (fetch x2 %.1f + write %.1f)" % (k[:50], v["hbm_bytes_per_launch"] / 1e6,
                                                                      v["fetch_bytes_x2"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()

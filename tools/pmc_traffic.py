"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes to profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [round tag] [gemm_api_grids.json]

Per MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) under-reports wide coalesced streaming reads by exactly 2x on
gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-byte streaming stores."""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter, api_grids):
    """{kernel name: values}; the dominant GEMM instantiation also under name + ' [gemm api]' / ' [pointwise conv]' by grid size"""
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            name = name.split("(")[0].replace(" ", "")
            agg[name].append(float(r["Counter_Value"]))
            if name.startswith("msda_fused_lanes_kernel<"):  # encoder tail / decoder launches of the lane kernel apart, by grid size
                agg[name + " [grid %d]" % int(r["Grid_Size"])].append(float(r["Counter_Value"]))
            if name.startswith("msda_window_kernel<"):
                agg["msda_window_kernel"].append(float(r["Counter_Value"]))
            if name.startswith("bneck_kernel<") or name.startswith("bneck2_kernel<"):   # every fused bottleneck launch together as well
                agg["bneck_kernel"].append(float(r["Counter_Value"]))
            if name.startswith("proj_ln_kernel<") or name.startswith("proj_ln2_kernel<"):   # both forms, all three FORMs
                agg["proj_ln_kernel"].append(float(r["Counter_Value"]))
            if name.startswith("conv3x3_patch_kernel<"):
                agg["conv3x3_patch_kernel"].append(float(r["Counter_Value"]))
            if name.startswith("gemm_f16x3_kernel<128,128,0,0") or name.startswith("gemm_bf16x6_kernel<128,128,0,0") or \
                    name.startswith("gemm_f32_kernel<128,128,64,64,0,0"):
                if api_grids:                                # the two populations of the instantiation, apart as well
                    agg[name + (" [gemm api]" if int(r["Grid_Size"]) in api_grids else " [pointwise conv]")].append(float(r["Counter_Value"]))
    return agg


def kernel_source_hash():
    """Same hash as bench.py: ties the counters to the build of the GEMM kernels they were taken on."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for name in ("gemm_f16x3.hip", "gemm_bf16x6.hip", "gemm_conv.hip", "gemm_k256.hip", "ffn_fused.hip", "proj_ln.hip", "msda.hip",
                 "dec_attn.hip", "dec_attn2.hip", "dec_tail.hip", "dec_tail2.hip", "bneck_fused.hip", "bneck2.hip", "conv3x3_patch.hip", "common.h"):
        with open(os.path.join(root, "gomatching_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    api_grids = set(json.load(open(sys.argv[4]))) if len(sys.argv) > 4 else set()      # work-items of the GEMM-API launches
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE", api_grids), per_kernel(sys.argv[2], "WRITE_SIZE", api_grids)
    out = {"_meta": {"kernel_source_hash": kernel_source_hash(), "round": tag,
                     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/gemm_shapes.py-free bench.py steps; "
                             "FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM).  The dominant GEMM instantiation appears three times: "
                             "its plain key = every launch of it (what bench.py's `roofline` covers), key + ' [gemm api]' = launches "
                             "whose grid is a GEMM-API shape of the transformer (grid sizes written by GOM_BENCH_WRITE_GRIDS), key + "
                             "' [pointwise conv]' = the rest (the backbone's pointwise convolutions)"}}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1) * 1024.0
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) * 1024.0
        out[k] = {"dispatches": len(fetch.get(k, [])), "fetch_size_bytes_raw": f, "fetch_bytes_x2": 2 * f,
                  "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w, "round": tag}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fjs:
        json.dump(out, fjs, indent=1, sort_keys=True)
    for k, v in sorted([kv for kv in out.items() if kv[0] != "_meta"], key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:10]:
        print("%-50s %8.1f MB/launch (fetch x2 %.1f + write %.1f)" % (k[:50], v["hbm_bytes_per_launch"] / 1e6,
                                                                      v["fetch_bytes_x2"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()

"""Reduce a rocprofv3 kernel trace of tools/gemm_shapes.py to one row per shape (profiles/r02_gemm_shapes.csv)."""
import csv
import json
import sys

trace, order = sys.argv[1], json.load(open(sys.argv[2]))
out = sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_gemm_shapes.csv"
rows = [r for r in csv.DictReader(open(trace))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ("gemm_f16x3_kernel", "gemm_k256_kernel", "proj_ln_kernel", "proj_ln2_kernel", "ffn_fused_kernel", "dec_tail_kernel", "dec_tail2_kernel", "dec_attn_kernel", "dec_attn2_kernel",
         "bneck2_kernel")
launches = []
for r in rows:
    if not any(n in r["Kernel_Name"] for n in names) or "split" in r["Kernel_Name"]:
        continue
    if "ffn_fused_kernel<false, 1>" in r["Kernel_Name"] and launches:
        # the half-height tail launch of a long fused-FFN call (csrc/ffn_fused.hip): one call = the two launches together
        launches[-1] = dict(launches[-1], End_Timestamp=r["End_Timestamp"])
        continue
    launches.append(r)
pos = 0
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["shape", "kernel", "launches_timed", "avg_us", "min_us", "tflops_fp32_equivalent", "frac_of_833"])
    for o in order:
        sel = []
        while len(sel) < o["launches"] and pos < len(launches):
            r = launches[pos]
            pos += 1
            if o["kernel"] in r["Kernel_Name"]:
                sel.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        sel = sel[2:]                                            # the two warm-up launches
        avg = sum(sel) / max(len(sel), 1)
        tf = o["flops"] / avg / 1e6 if avg else 0.0
        name = o["kernel"]
        w.writerow([o["label"], name, len(sel), "%.1f" % avg, "%.1f" % min(sel), "%.1f" % tf, "%.3f" % (tf / 833.3)])
        print("%-64s %-36s %8.1f us %6.1f TFLOP/s %.3f" % (o["label"], name[:36], avg, tf, tf / 833.3))

"""print the top kernels of gpurun_out/bench_details.json (developer tool): python tools/show_details.py [file] [rows]"""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_details.json"))
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 18
print(d["main"].get("workload", "")[:60], d["main"].get("ms_per_step"), "ms; launches", d["main"].get("launches_per_step"))
for k, v in d["conv_kernels"].items():
    tot = sum(r["ms"] for r in v.values())
    print("==", k, "kernel ms %.2f" % tot)
    for i, (kn, r) in enumerate(v.items()):
        if i < rows:
            print("  %-78s %3d x %8.1f us = %6.3f ms %5.0f TF %5.0f GB/s" % (kn[:78], r["launches"], r["us_per_launch"], r["ms"], r["tflops"] or 0, r["alg_gbs"] or 0))

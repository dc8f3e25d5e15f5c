"""Reads a rocprofv3 kernel-trace csv of mx_gemm_sweep.py and prints the mx256 kernel durations in dispatch order."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mx256" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
shapes = [(3072, 256), (3072, 512), (3072, 1024), (3072, 2048), (4096, 1024), (1024, 1024), (1024, 4096)]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 34816
i = 0
for (N, K) in shapes:
    for out in ("f32 out", "mx  out"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:i + 3]]
        i += 3
        if d:
            us = min(d)
            print(f"M={M} N={N} K={K} {out}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  ({rows[i-1]['Kernel_Name'][:40]})")

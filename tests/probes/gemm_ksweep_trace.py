"""Scratch: kernel durations (rocprofv3 --kernel-trace) of the K sweep, i.e. without the Python call overhead that
`gpu_gemm_ksweep.py` measures on top.  Run: rocprofv3 --kernel-trace --output-format csv -d OUT -o k -- python3
tests/probes/gpu_gemm_ksweep.py; then python3 tests/probes/gemm_ksweep_trace.py OUT/k_kernel_trace.csv"""
import csv, sys, json, collections

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the sweep launches 53 GEMMs per (M, N, K) cell (3 warm-up + 50 timed), cells in script order
shapes = [(1025, 3072), (1025, 1024), (1025, 4096), (1024, 3072), (128, 3072)]
Ks = (256, 512, 1024, 2048, 4096)
i = 0
for (M, N) in shapes:
    out = {}
    for K in Ks:
        cell = rows[i:i + 53]; i += 53
        d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in cell[3:])
        out[K] = round(d[len(d) // 2] / 1e3, 1)
    ks = sorted(out)
    slope = (out[4096] - out[256]) / ((4096 - 256) / 64)
    print(json.dumps(dict(M=M, N=N, us_by_K=out, us_per_64K=round(slope, 3), fixed_us=round(out[256] - 4 * slope, 1), kernel=cell[-1]["Kernel_Name"][:40])))

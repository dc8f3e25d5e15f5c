"""Scratch probe (not a test): L2 -> CU fill rate vs workgroups per CU, waves and loads in flight."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import valle_amd  # noqa
from valle_amd.engine import load_probe_library
load_probe_library()  # libvallex_probes.so: `python vall-e_amd/csrc/build.py --probes`
from valle_amd.engine import l2_fill

for region in (2 << 20, 32 << 20, 1 << 30):
    for grid, threads in ((256, 64), (256, 256), (512, 256), (1024, 256), (2048, 256)):
        for unroll in (2, 8, 16):
            r = l2_fill(grid, threads, unroll, region, 100)
            print(json.dumps(dict(region_mb=region >> 20, grid=grid, threads=threads, unroll=unroll, gbs=round(r["gbs"]), bpc=round(r["bytes_per_clk_per_cu"], 1), ghz=r["ghz"])), flush=True)

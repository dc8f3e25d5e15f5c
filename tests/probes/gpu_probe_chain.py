"""Scratch probe (not a test): cost of one dependent GEMV stage inside a single persistent launch."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import valle_amd  # noqa
from valle_amd.engine import load_probe_library
load_probe_library()  # libvallex_probes.so: `python vall-e_amd/csrc/build.py --probes`
from valle_amd.engine import stage_chain, launch_floor

for nwg, rows in ((256, 12), (256, 4)):
    for mode in (0, 2, 3, 4):
        r = stage_chain(nwg=nwg, stages=60, rows=rows, mode=mode, iters=20)
        print(json.dumps(dict(nwg=nwg, rows=rows, mode=mode, **r)), flush=True)
        if r["spin_timeout"]:
            sys.exit(1)
print(json.dumps(launch_floor(62, 256, 256, 100)))

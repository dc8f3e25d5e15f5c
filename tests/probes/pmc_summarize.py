"""Summarise rocprofv3 --pmc counter_collection.csv files of tests/probes/pmc_driver.py into per-AR-step sums.
usage: python tests/probes/pmc_summarize.py OUT.json NAME=dir [NAME=dir ...]"""
import collections
import csv
import glob
import json
import sys

out = {}
for spec in sys.argv[2:]:
    name, d = spec.split("=")
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "sample_embed" in r["Kernel_Name"]]
    steps = []
    for a, b in zip(idx, idx[1:]):
        if b - a != 62:
            continue
        per = collections.Counter()
        for r in rows[a:b]:
            n = r["Kernel_Name"]
            per["sample" if "sample" in n else "attn" if "attn_decode" in n else "gemv"] += float(r["Counter_Value"])
        steps.append(dict(per))
    tot = [sum(s.values()) for s in steps]
    out[name] = dict(counter=rows[0]["Counter_Name"], steps=len(steps), mean_per_step=sum(tot) / len(tot), first=tot[0], last=tot[-1],
                     mean_gemv=sum(s.get("gemv", 0) for s in steps) / len(steps), mean_attn=sum(s.get("attn", 0) for s in steps) / len(steps),
                     mean_sample=sum(s.get("sample", 0) for s in steps) / len(steps))
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))

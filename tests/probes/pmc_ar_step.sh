#!/bin/bash
# FETCH_SIZE and WRITE_SIZE of the batch-1 AR decode step (eager launches, torch-free driver), one rocprofv3 --pmc pass each, and the
# summary bench.py quotes as roofline.traffic.   usage: tests/probes/pmc_ar_step.sh OUTDIR [steps]   -> OUTDIR/pmc_ar_step.json
out=$1; steps=${2:-24}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o g -- python3 tests/probes/pmc_driver.py $steps > $out/$c.log 2>&1 || echo "pass $c failed (see $out/$c.log)"
done
find $out -name "*kernel_trace.csv" -delete
python3 - "$out" "$steps" <<'PY'
import collections, csv, glob, json, sys
out, nstep = sys.argv[1], int(sys.argv[2])
res = {"source": f"rocprofv3 --pmc <counter> --kernel-trace -- python3 tests/probes/pmc_driver.py {nstep} (torch-free C-ABI driver, eager launches; tests/probes/pmc_ar_step.sh), the default decode step",
       "gfx950_correction": "x2 for 16-byte-per-lane coalesced streaming reads (MI355X_MICROARCH.md, HBM section); counter unit KB"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        res[c] = None
        continue
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "sample_embed" in r["Kernel_Name"]]
    steps = []
    for a, b in zip(idx, idx[1:]):
        if b - a not in (62, 26):  # five launches per layer (rounds 1-2, VX_AR_TP=0) / the XCD-sharded step's two
            continue
        per = collections.Counter()
        for r in rows[a:b]:
            n = r["Kernel_Name"]
            per["sample" if "sample" in n else "attn" if ("attn_decode" in n or "tp_attn" in n) else "ffn" if "tp_ffn" in n else "gemv"] += float(r["Counter_Value"])
        steps.append(per)
    n = len(steps)
    res[c] = {"steps": n, "launches_per_step": b - a, "kb_per_step_mean": sum(sum(s.values()) for s in steps) / n,
              "kb_gemv_mean": sum(s["gemv"] for s in steps) / n, "kb_attn_mean": sum(s["attn"] for s in steps) / n,
              "kb_ffn_mean": sum(s["ffn"] for s in steps) / n, "kb_sample_mean": sum(s["sample"] for s in steps) / n}
f = res.get("FETCH_SIZE")
if f:
    ctx_first = 47 + 225 + 1
    ctx_mean = ctx_first + (f["steps"] - 1) / 2.0
    res.update(steps=f["steps"], ctx_first=ctx_first, ctx_last=ctx_first + f["steps"] - 1, ctx_mean=ctx_mean,
               hbm_bytes_per_step_corrected=f["kb_per_step_mean"] * 1024 * 2,
               algorithmic_bytes_at_same_ctx=304412672 + 49152 * ctx_mean)
    if f["launches_per_step"] == 62:
        res.update(gemv_bytes_per_step_corrected=(f["kb_gemv_mean"] + f["kb_sample_mean"]) * 1024 * 2,
                   attn_bytes_per_ctx_row_corrected=f["kb_attn_mean"] * 1024 * 2 / ctx_mean)
    res["ratio_traffic_over_algorithmic"] = res["hbm_bytes_per_step_corrected"] / res["algorithmic_bytes_at_same_ctx"]
    w = res.get("WRITE_SIZE")
    if w:
        res["write_bytes_per_step"] = w["kb_per_step_mean"] * 1024
json.dump(res, open(out + "/pmc_ar_step.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

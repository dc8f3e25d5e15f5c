#!/bin/bash
# PMC passes (separate rocprofv3 runs) + one kernel-stats pass over the batched NAR stages alone (nar_batch_driver.py).
#   usage: tests/probes/pmc_nar_batch.sh OUTDIR [driver arguments]      -> OUTDIR/summary.json (per kernel: mean counter values, mean duration)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o t -- python3 tests/probes/nar_batch_driver.py "$@" > $out/stats.log 2>&1 || echo "stats pass failed"
i=0
for c in ${PMC_SETS:-"SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_INST_CYCLES_VMEM" "TCC_HIT_sum,TCC_MISS_sum,TCC_EA0_RDREQ_sum,TCC_REQ_sum"}; do
  i=$((i+1)); c=$(echo $c | tr ',' ' ')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -o g -- python3 tests/probes/nar_batch_driver.py "$@" > $out/p$i.log 2>&1 || echo "pmc pass $i failed: $c"
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
short = lambda n: n.split("(")[0].replace("void vx::", "").replace("vx::", "")[:60]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for f in glob.glob(f"{out}/stats/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        res.setdefault(k, {})["dur_us_mean"] = sum(v) / len(v) / 1e3
        res[k]["launches"] = len(v)
for k, v in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("dur_us_mean", 0) * kv[1].get("launches", 0))[:8]:
    print(k, {a: round(b, 3) if b < 10 else round(b) for a, b in v.items()})
PY
find $out -name "*kernel_trace.csv" -size +8M -delete; find $out -name "*counter_collection.csv" -size +8M -delete

#!/bin/bash
# Instruction-cache behaviour of the decode step's kernels (eager launches, torch-free driver): one rocprofv3 --pmc pass.
#   usage: tests/probes/pmc_icache.sh OUTDIR [steps]
out=$1; steps=${2:-12}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_VALU\b\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INST_CYCLES[A-Z_]*\|SQC_ICACHE_MISSES_DUPLICATE" | sort -u > $out/counters.txt
cat $out/counters.txt
for c in ${PMC_SETS:-"SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"}; do
  c=$(echo $c | tr ',' ' '); tag=$(echo $c | tr ' ' '+')
  timeout -k 10 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$tag -o g -- python3 tests/probes/pmc_driver.py $steps > $out/$tag.log 2>&1 || echo "pass $tag failed"
done
find $out -name "*kernel_trace.csv" -delete
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void vx::", "")
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: round(sum(v) / len(v), 1) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
json.dump(res, open(out + "/icache.json", "w"), indent=1)
for k, v in res.items():
    print(k, v)
PY

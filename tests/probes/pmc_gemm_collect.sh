#!/bin/bash
# Scratch: PMC passes over one GEMM shape (separate rocprofv3 runs per counter group, kernel-trace only).
# usage: tests/probes/pmc_gemm_collect.sh OUTDIR M N K
out=$1; M=$2; N=$3; K=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o g --output-format csv -- python3 tests/probes/pmc_gemm_driver.py $M $N $K 3 > $out/p$i.log 2>&1 || echo "fail group $i: $c"
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys, json
res = {}
for d in sorted(glob.glob(sys.argv[1] + "/p*/")):
    try:
        rows = list(csv.DictReader(open(d + "g_counter_collection.csv")))
    except Exception as e:
        print("no data", d, e); continue
    agg = collections.defaultdict(list)
    for r in rows:
        if "mfma" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k] = sum(v) / len(v)
    kt = list(csv.DictReader(open(d + "g_kernel_trace.csv")))
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt if "mfma" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"]]
    if durs: res.setdefault("dur_us", []).append(round(sum(durs) / len(durs) / 1e3, 1))
print(json.dumps(res, indent=1))
json.dump(res, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY

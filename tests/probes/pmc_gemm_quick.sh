#!/bin/bash
# Three PMC passes (separate rocprofv3 runs, kernel-trace only) over one bf16 GEMM shape through vx_op_gemm: MFMA busy, waits, LDS.
# usage: tests/probes/pmc_gemm_quick.sh OUTDIR M N K [form]        -> OUTDIR/summary.json   (form: see pmc_gemm_driver.py)
out=$1; M=$2; N=$3; K=$4; form=${5:-f32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o g --output-format csv -- python3 tests/probes/pmc_gemm_driver.py $M $N $K 3 $form > $out/p$i.log 2>&1 || echo "fail group $i: $c"
done
python3 - "$out" "$M" "$N" "$K" "$form" <<'PY'
import csv, glob, collections, sys, json
res = {"shape": [int(v) for v in sys.argv[2:5]], "form": sys.argv[5]}
for d in sorted(glob.glob(sys.argv[1] + "/p*/")):
    try:
        rows = list(csv.DictReader(open(glob.glob(d + "**/*counter_collection.csv", recursive=True)[0])))
    except Exception as e:
        print("no data", d, e); continue
    agg = collections.defaultdict(list)
    for r in rows:
        if ("gemm" in r["Kernel_Name"] or "mfma256" in r["Kernel_Name"] or "mx256" in r["Kernel_Name"]) and "quant" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["kernel"] = r["Kernel_Name"][:90]
    for k, v in agg.items():
        res[k] = sum(v) / len(v)
    kt = list(csv.DictReader(open(glob.glob(d + "**/*kernel_trace.csv", recursive=True)[0])))
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt if ("gemm" in r["Kernel_Name"] or "mfma256" in r["Kernel_Name"] or "mx256" in r["Kernel_Name"]) and "quant" not in r["Kernel_Name"]]
    if durs: res.setdefault("dur_us", []).append(round(sum(durs) / len(durs) / 1e3, 1))
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
    # GRBM_GUI_ACTIVE counts per XCD (8): chip-wide SIMD-cycles = GUI / 8 x 1024 SIMDs (r01_notes.md)
    res["mfma_busy_frac"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] / 8 * 1024)
    res["clock_ghz"] = res["GRBM_GUI_ACTIVE"] / 8 / (sum(res["dur_us"]) / len(res["dur_us"])) / 1e3
    res["tflops"] = 2.0 * res["shape"][0] * res["shape"][1] * res["shape"][2] / (sum(res["dur_us"]) / len(res["dur_us"])) / 1e6
print(json.dumps(res, indent=1))
json.dump(res, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY

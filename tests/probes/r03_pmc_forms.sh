#!/bin/bash
# PMC matrix-pipe occupancy of the four GEMM forms the batched NAR pass launches at 34 816 rows (+ the MXFP8 QKV form), round-3 build
#   bash tests/probes/r03_pmc_forms.sh OUTDIR  -> OUTDIR/forms.json
out=$1; mkdir -p $out
for spec in "qkv 34816 3072 1024" "relu 34816 4096 1024" "resid 34816 1024 4096" "resid_o 34816 1024 1024"; do
  set -- $spec; form=$1; f=${form%_o}
  bash tests/probes/pmc_gemm_quick.sh $out/$form $2 $3 $4 $f > $out/$form.log 2>&1 || echo "failed $form"
done
python3 - "$out" <<'PY'
import json, sys, glob
o = sys.argv[1]
runs = []
for f in sorted(glob.glob(o + "/*/summary.json")):
    r = json.load(open(f)); r["run"] = f.split("/")[-2]; runs.append(r)
fl = sum(2.0 * r["shape"][0] * r["shape"][1] * r["shape"][2] for r in runs if "mfma_busy_frac" in r)
w = sum(2.0 * r["shape"][0] * r["shape"][1] * r["shape"][2] * r["mfma_busy_frac"] for r in runs if "mfma_busy_frac" in r)
res = {"what": "mfma256p_kernel in the forms the batched NAR pass launches (vx_op_gemm_rows), 34816 rows, random bf16 operands, torch-free driver, three separate rocprofv3 --pmc passes each (tests/probes/pmc_gemm_quick.sh); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); round-3 build (-amdgpu-mfma-vgpr-form)",
       "flop_weighted_mfma_busy_frac": w / fl if fl else None, "runs": runs}
json.dump(res, open(o + "/forms.json", "w"), indent=1)
for r in runs:
    print(r["run"], r["shape"], "busy %.3f" % r.get("mfma_busy_frac", -1), "TF %.0f" % r.get("tflops", -1), r.get("dur_us"))
print("flop-weighted busy", res["flop_weighted_mfma_busy_frac"])
PY

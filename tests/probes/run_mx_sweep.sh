R=$PWD; mkdir -p gpurun_out/r2; cd /tmp; export TMPDIR=/tmp
for a in 0 1; do
  VX_MX_ALG=$a timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r2/prof_mx$a -o t --output-format csv -- python3 $R/tests/probes/mx_gemm_sweep.py > $R/gpurun_out/r2/mx_sweep$a.log 2>&1
  f=$(find $R/gpurun_out/r2/prof_mx$a -name "*kernel_trace.csv" | head -1)
  echo "VX_MX_ALG=$a"; python3 $R/tests/probes/mx_gemm_sweep_report.py $f
done

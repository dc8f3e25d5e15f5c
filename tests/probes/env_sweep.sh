for kv in "X=1" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "GPU_FLUSH_ON_EXECUTION=1" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "AMD_DIRECT_DISPATCH=0"; do
  echo "== $kv"
  env $kv timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ar_step_us'], d['nar_7stage_ms'], d['prefill_ms'])" || exit 1
done

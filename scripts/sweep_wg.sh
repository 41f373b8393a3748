# diagnostic: k_evaluate workgroup size
for wg in 64 128 256; do
  echo -n "wg=$wg "; FOT_LANES=${FOT_LANES:-1} FOT_EVAL_WG=$wg timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])" || echo failed
done

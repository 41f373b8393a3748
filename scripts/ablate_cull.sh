# diagnostic (wrong results for ablate != 0): 1 = no wave ranges, 2 = no scatter pass, 4 = no classification pass, 8 = no profile boxes,
# 16 = stop after the per-instance constants, 32 = stop at entry (the launch alone), 64 = no gathers
for a in ${@:-0 1 2 4 7 15 79 16 32}; do
  echo -n "ablate=$a "; FOT_CULL_ABLATE=$a timeout -k 10 120 python bench.py --steps 50 --warmup 5 --overlap 1 --no-cpu-baseline --no-latency --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['kernel_ms']['k_cull'], d['kernel_ms']['k_evaluate'], d['ms_per_step'])" || echo failed
done

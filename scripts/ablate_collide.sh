for a in 0 1 2 3 4 7 8 12; do
  echo -n "ablate=$a "; FOT_COLLIDE_ABLATE=$a timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-latency --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['kernel_ms']['k_collide'], d['ms_per_step'])" || echo failed
done

# diagnostic: whole-step time against the number of internal lanes (sub-batches on separate streams)
for l in 1 2 3 4; do
  echo -n "lanes=$l "; FOT_LANES=$l timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])" || echo failed
done

# diagnostic: A/B one environment switch of libfot on the bench workload.  usage: ab_env.sh VAR [overlaps...]
var=$1; shift
for ov in ${@:-1 2}; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $var=1; else unset $var; fi
    echo -n "$var=$on overlap=$ov "; timeout -k 10 120 python bench.py --steps 100 --warmup 10 --overlap $ov --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" || echo failed
  done
done

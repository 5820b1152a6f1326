# usage: tools/bench_env.sh VAR v1 v2 ...   -- one short bench run per value of an environment variable
var=$1; shift
for v in "$@"; do echo "== $var=$v"; env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 $BENCH_ARGS | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'])" || exit 1; done

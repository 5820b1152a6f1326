for f in raytracer_amd/_var/*.so; do echo "== $f"; RT_HIP_LIB=$f timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 $BENCH_ARGS | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'])" || exit 1; done

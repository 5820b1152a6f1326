#!/bin/bash
# bench.py per library variant: frame time and stage times (development aid; run on the GPU box).
# usage: tools/ab_bench.sh [bunny|atrium] cur NAME ...   -- "cur" = the in-tree library, other names = raytracer_amd/_var/<name>.so
w=$1; shift
for name in "$@"; do
  lib=raytracer_amd/_var/$name.so; [ "$name" = cur ] && lib=raytracer_amd/librt_hip.so
  RT_HIP_LIB=$lib timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', '$w', 'Mrays/s %.0f  ms %.1f' % (d['value'], d['ms_per_step']), {k: round(v,1) for k,v in d['stage_ms_per_step'].items()}, [round(e['avg_launch_ms']*e['launches']/d['steps'],1) for e in [d['roofline']]+d['roofline_other_trace_kernels']])" || exit 1
done

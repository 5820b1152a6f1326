#!/bin/bash
# Throughput against rays per round (and with it the workspace) on the benchmark frame
# (development aid; run on the GPU box).
for mi in 2 4 8 16 32 64; do
  echo "== $mi Mi rays per round"
  RT_WAVEFRONT_RAYS=$((mi << 20)) timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workspace-gb 200 $BENCH_ARGS | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms', round(d['workspace_gb'],2), 'GB', d['stage_ms_per_step'])" || exit 1
done

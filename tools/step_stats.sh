#!/bin/bash
# Lane-occupancy and per-phase clock statistics of the closest-hit engine (development aid; run on
# the GPU box).  usage: tools/step_stats.sh 4 7 8   -- the variants raytracer_amd/_var/st<N>.so,
# each built with -DRT_STEP_STATS=<N>
for v in "$@"; do
  for sc in "bunny --size 512" "atrium --size 270" "soup --size 1080"; do
    echo "== st$v $sc"
    RT_HIP_LIB=raytracer_amd/_var/st$v.so timeout -k 10 150 python -u tools/time_render.py --scene $sc --n 4 --reps 2 | grep "render\|primary" | tail -1 || exit 1
  done
done

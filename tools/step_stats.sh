#!/bin/bash
# Lane-occupancy and per-phase clock statistics of the closest-hit engine (development aid; run on
# the GPU box).  Needs the variants built with tools/build_variant.sh-style flags:
#   raytracer_amd/_var/st3.so (-DRT_STEP_STATS=3), st4.so, st6.so
for v in 3 4 6; do
  for sc in "bunny --size 512" "atrium --size 270"; do
    echo "== st$v $sc"
    RT_HIP_LIB=raytracer_amd/_var/st$v.so timeout -k 10 150 python -u tools/time_render.py --scene $sc --n 4 --reps 1 | grep render || exit 1
  done
done

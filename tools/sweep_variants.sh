#!/bin/bash
# Time every library variant under raytracer_amd/_var on the bench scene and the soup scene
# (development aid; run on the GPU box).
for f in raytracer_amd/_var/*.so; do
  echo "== $f"
  RT_HIP_LIB=$f timeout -k 10 120 python -u tools/time_render.py --scene bunny --size 1024 --n 4 --reps 2 | grep render | tail -1 || exit 1
  RT_HIP_LIB=$f timeout -k 10 120 python -u tools/time_render.py --scene soup --size 2160 --reps 3 | grep primary | tail -1 || exit 1
done

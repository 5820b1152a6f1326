#!/bin/bash
# A/B timing of library variants on both benchmark scenes (development aid; run on the GPU box).
# usage: tools/ab.sh cur old ...   -- "cur" = the in-tree library, other names = raytracer_amd/_var/<name>.so
for name in "$@"; do
  lib=raytracer_amd/_var/$name.so; [ "$name" = cur ] && lib=raytracer_amd/librt_hip.so
  for sc in "bunny --size 1024" "atrium --size 540"; do
    echo "== $name $sc"
    RT_HIP_LIB=$lib timeout -k 10 150 python -u tools/time_render.py --scene $sc --n 4 --reps 3 2>&1 | grep render | tail -1 || exit 1
  done
done

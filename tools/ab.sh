#!/bin/bash
# A/B timing of library variants on the benchmark scenes (development aid; run on the GPU box).
# usage: tools/ab.sh cur prev ...   -- "cur" = the in-tree library, other names = raytracer_amd/_var/<name>.so
for name in "$@"; do
  lib=raytracer_amd/_var/$name.so; [ "$name" = cur ] && lib=raytracer_amd/librt_hip.so
  for sc in "bunny --size 1024" "atrium --size 540" "soup --size 2160"; do
    echo "== $name $sc"
    RT_HIP_LIB=$lib timeout -k 10 150 python -u tools/time_render.py --scene $sc --n 4 --reps 4 2>&1 | grep "render\|primary" | sort -t: -k2 -n | head -1 || exit 1
  done
done

#!/bin/bash
# Build a named variant of librt_hip.so with extra compiler flags (development aid):
#   tools/build_variant.sh NAME -DRT_NODE_MIN=16 ...   ->  raytracer_amd/_var/NAME.so
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$here/raytracer_amd/_var"
make -s -C "$here/raytracer_amd/csrc" OUT="$here/raytracer_amd/_var/$name.so" EXTRA="$*" 2>&1 | grep -v "hip-link" || true

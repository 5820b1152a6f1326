#!/bin/bash
# One rocprofv3 SQ-counter pass over tools/time_render.py for a library variant (development aid).
# usage: tools/pmc_sq.sh <name|cur> <scene args...>
set -e
name=$1; shift
lib=$GRAFT_REPO_ROOT/raytracer_amd/_var/$name.so; [ "$name" = cur ] && lib=$GRAFT_REPO_ROOT/raytracer_amd/librt_hip.so
out=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$name
export RT_HIP_LIB=$lib
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/time_render.py "$@" > $out.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/sum_pmc.py $out/*/p_counter_collection.csv 2>/dev/null || python3 $GRAFT_REPO_ROOT/tools/sum_pmc.py $(find $out -name "p_counter_collection.csv" | head -1)

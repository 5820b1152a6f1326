#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats, then separate PMC passes (HBM bytes, SQ
# counters).  usage: tools/profile_bench.sh [bunny|atrium]
# Outputs land in gpurun_out/prof_<workload>/ ; tools/summarize_profile.py turns them into profiles/.
set -e
W=${1:-bunny}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/sq -o b -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1
grep -h "\"metric\"" $O/*.log | cut -c1-160 | head -4

#!/bin/bash
# One counter pass over a short render (development aid; run on the GPU box):
#   tools/pmc_pass.sh NAME SCENE SIZE COUNTER...   ->  gpurun_out/pmc_NAME/ , per-kernel sums on stdout
set -e
name=$1; scene=$2; size=$3; shift 3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$name
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -o p -- python3 $R/tools/time_render.py --scene $scene --size $size --n 4 --reps 2 > $O/run.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    print(k, {c: f'{x:.4g}' for c, x in sorted(v.items())})
PY

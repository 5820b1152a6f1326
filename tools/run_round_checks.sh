set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests_na.log 2>&1
tail -3 gpurun_out/gputests_na.log
RT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 1 --warmup 0 --size 512 > gpurun_out/bench_n2_rehearsal.json 2> gpurun_out/bench_n2.err
cat gpurun_out/bench_n2_rehearsal.json | cut -c1-600
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_bunny.json 2> gpurun_out/bench_bunny.err
cut -c1-300 gpurun_out/bench_bunny.json
timeout -k 10 300 python bench.py --no-cpu-baseline --workload atrium --steps 1 > gpurun_out/bench_atrium.json 2> gpurun_out/bench_atrium.err
cut -c1-300 gpurun_out/bench_atrium.json

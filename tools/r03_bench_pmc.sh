#!/bin/bash
# HBM traffic of the bench command itself: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over a short run of bench.py at its
# default chunk; aggregated per kernel by tools/pmc_collect.py and wrapped with the command and chunk so bench.py can attach it as roofline.traffic.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-similarity --no-comparator --no-both-modes"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r03bench_$c -- python3 $R/$CMD > $R/gpurun_out/pmc_r03bench_$c.log 2>&1
  echo "pass $c rc=$?"
done
cd $R && python tools/pmc_collect.py gpurun_out/pmc_r03bench_FETCH_SIZE gpurun_out/pmc_r03bench_WRITE_SIZE > gpurun_out/r03_bench_pmc_kernels.json
python - <<PY
import json
k = json.load(open('gpurun_out/r03_bench_pmc_kernels.json'))
json.dump({'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 $CMD', 'chunk': 124, 'kernels': k}, open('gpurun_out/r03_bench_pmc.json', 'w'), indent=1)
for n, v in k.items():
    print(n, v.get('launches_per_pass'), v.get('us_per_launch_under_pmc'), v.get('traffic_bytes'))
PY
rm -rf gpurun_out/pmc_r03bench_FETCH_SIZE gpurun_out/pmc_r03bench_WRITE_SIZE

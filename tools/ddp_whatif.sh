#!/bin/bash
# One-GPU what-if of the data-parallel machinery (1-rank RCCL communicators, PLM_FORCE_REDUCER=1):  gpurun -- 'bash tools/ddp_whatif.sh'
# Prints ms/step of bench.py --no-extras for: no reducer, then the reducer with different dW-group budgets / bucket caps / reserves
# (PLM_BENCH_FIRST_ALT picks the data plane the timed region runs on: without it bench.py always times cap 0 first).
cd "$(dirname "$0")/.."
run() { echo -n "$1 :: "; shift; env "$@" python bench.py --no-extras --steps 20 --warmup 6 ${ARGS:-} 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); c=o.get('comm',{}); print(o['ms_per_step'], 'ms; n1', c.get('n1_ms_per_step'), 'exposed', c.get('exposed_comm_ms'), 'reserved_frac', c.get('reserved_launch_frac'), 'selected', c.get('selected'))"; }
for r in 1 2; do
run "no reducer                         " PLM_X=0
run "reducer cap 0, dW group 80 MB      " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0
run "reducer cap 0, dW group 160 MB     " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_DW_GROUP_MB=160
run "reducer cap 0, dW group 1000 MB    " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_DW_GROUP_MB=1000
ARGS="--bucket-mb 128" run "reducer cap 0, bucket 128 MiB      " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0
run "reducer cap 16 + tail, windows     " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_BENCH_FIRST_ALT=allreduce,16,1
run "reducer cap 16, model 60 GB/s      " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_BENCH_FIRST_ALT=allreduce,16,1 PLM_COMM_MODEL_GBPS=60
run "reducer cap 16, model 120 GB/s     " PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_BENCH_FIRST_ALT=allreduce,16,1 PLM_COMM_MODEL_GBPS=120
run "reducer cap 16, model 1 GB/s (always)" PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_BENCH_FIRST_ALT=allreduce,16,1 PLM_COMM_MODEL_GBPS=1
done

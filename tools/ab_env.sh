#!/bin/bash
# Same-box A/B of environment settings inside the training step:  gpurun -- 'bash tools/ab_env.sh 2 "" "PLM_NT_NO_HYBRID=1" "PLM_HEAD_CHUNK=16384"'
# Prints tokens/s of `bench.py --steps 20 --warmup 5 --no-extras` for every setting, N rounds, interleaved.
N=$1; shift
cd "$(dirname "$0")/.."
for i in $(seq "$N"); do
  for v in "$@"; do
    echo -n "[$v]  "
    env $v python bench.py --steps 20 --warmup 5 --no-extras ${BENCH_ARGS:-} 2>&1 | grep '"metric"' | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 tok\/s  \2 ms/'
  done
done

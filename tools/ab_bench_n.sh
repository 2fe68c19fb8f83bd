#!/bin/bash
# Same-box A/B of several builds of libplainlm_hip.so inside the training step:  gpurun -- 'bash tools/ab_bench_n.sh 3 tools/_lib_a.so tools/_lib_b.so tools/_lib_c.so'
N=$1; shift
cd "$(dirname "$0")/.."
cp plainlm_amd/libplainlm_hip.so /tmp/_plm_orig.so
trap 'cp /tmp/_plm_orig.so plainlm_amd/libplainlm_hip.so' EXIT
for i in $(seq "$N"); do
  for v in "$@"; do
    cp "$v" plainlm_amd/libplainlm_hip.so
    echo -n "$v  "
    python bench.py --steps 20 --warmup 5 --no-extras ${BENCH_ARGS:-} 2>&1 | grep '"metric"' | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 tok\/s  \2 ms/'
  done
done

#!/bin/bash
# Same-box A/B of two builds of libplainlm_hip.so inside the training step (the pool's boxes differ by +-3 %, and
# back-to-back kernel timings mis-rank schedules - see DESIGN.md section 8).
#
#   here:        build the two variants, e.g.
#                  git stash; make -C plainlm_amd/csrc; cp plainlm_amd/libplainlm_hip.so tools/_lib_a.so; git stash pop
#                  make -C plainlm_amd/csrc;             cp plainlm_amd/libplainlm_hip.so tools/_lib_b.so
#                (tools/_lib_*.so are git-ignored but travel with gpurun)
#   on the box:  gpurun -- 'bash tools/ab_bench.sh tools/_lib_a.so tools/_lib_b.so 3'
#
# Prints tokens/s of `bench.py --steps 20 --warmup 5 --no-extras` for a, b, a, b, ... and restores the original library.
set -e
A=$1; B=$2; N=${3:-2}
cd "$(dirname "$0")/.."
cp plainlm_amd/libplainlm_hip.so /tmp/_plm_orig.so
trap 'cp /tmp/_plm_orig.so plainlm_amd/libplainlm_hip.so' EXIT
for i in $(seq "$N"); do
  for v in "$A" "$B"; do
    cp "$v" plainlm_amd/libplainlm_hip.so
    echo -n "$v  "
    python bench.py --steps 20 --warmup 5 --no-extras ${BENCH_ARGS:-} 2>&1 | grep '"metric"' | sed 's/.*"value": \([0-9.]*\).*/\1 tok\/s/'
  done
done

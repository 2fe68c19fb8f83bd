#!/bin/bash
# Same-box A/B of builds of libplainlm_hip.so on tools/kbench.py sections:  gpurun -- 'bash tools/ab_kbench.sh 3 "--only attn" "attn" tools/_lib_a.so tools/_lib_b.so ...'
# (N rounds, kbench arguments, grep pattern for the lines to keep, libraries; see tools/ab_bench.sh for how to build the variants)
N=$1; ARGS=$2; PAT=$3; shift 3
cd "$(dirname "$0")/.."
cp plainlm_amd/libplainlm_hip.so /tmp/_plm_orig.so
trap 'cp /tmp/_plm_orig.so plainlm_amd/libplainlm_hip.so' EXIT
for i in $(seq "$N"); do
  for v in "$@"; do
    cp "$v" plainlm_amd/libplainlm_hip.so
    python tools/kbench.py $ARGS 2>&1 | grep -E "$PAT" | sed "s|^|$v  |"
  done
done

#!/bin/bash
# round-3 call A: attention variants correctness + timing, then the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
AT='attention or attn or rope'
for v in "22 22 21" "21 21 21" "12 12 21" "0 0 0"; do
  set -- $v
  echo "=== variants fwd=$1 dq=$2 dkdv=$3"
  PLM_ATTN_FWD=$1 PLM_ATTN_DQ=$2 PLM_ATTN_DKDV=$3 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "$AT" 2>&1 | tail -4
done
echo "=== attn_ab"
timeout 600 python tools/attn_ab.py 2>&1 | tee gpurun_out/r3a_attn_ab.txt
echo "=== attn_ab doc"
timeout 600 python tools/attn_ab.py --doc 2>&1 | tee gpurun_out/r3a_attn_ab_doc.txt
echo "=== attn_ab T=2048 nh=16 B=8"
timeout 600 python tools/attn_ab.py --B 8 --T 2048 --nh 16 2>&1 | tee gpurun_out/r3a_attn_ab_420m.txt
echo "=== full suite (default variants)"
timeout 2400 python -m pytest tests -m gpu -q --timeout 2400 -rf -x > gpurun_out/r3a_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3a_tests.log
echo "=== bench"
timeout 900 python bench.py --no-extras > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"; head -c 600 gpurun_out/r3a_bench.json

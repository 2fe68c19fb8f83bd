#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/c6
O=gpurun_out/c6
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -x -k "attention" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -15 $O/t.log
for a in 0 2 3; do
  echo -n "pipe ABL=$a: "; PLM_ATTN_ABL=$a timeout 300 python tools/kbench.py --only attn --iters 30 2>&1 | grep "attn bwd"
done
echo -n "old: "; PLM_ATTN_BWD_V1=1 timeout 300 python tools/kbench.py --only attn --iters 30 2>&1 | grep "attn bwd"
echo -n "T2048 pipe: "; timeout 300 python tools/kbench.py --only attn --iters 30 --B 8 --T 2048 2>&1 | grep "attn bwd"
echo -n "T2048 old: "; PLM_ATTN_BWD_V1=1 timeout 300 python tools/kbench.py --only attn --iters 30 --B 8 --T 2048 2>&1 | grep "attn bwd"

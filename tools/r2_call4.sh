#!/bin/bash
# round-2 GPU call 4: what do two waves of a SIMD share (micro-benchmark), and where does the ping-pong dK/dV kernel lose its time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/c4
O=gpurun_out/c4
export TMPDIR=/tmp
./tools/ubench/overlap > $O/overlap.txt 2>&1; echo "overlap rc=$?"; cat $O/overlap.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -x -k "attention" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -3 $O/t.log
for a in 0 1 2 3 4 5; do
  echo -n "ABL=$a: "; PLM_ATTN_ABL=$a timeout 300 python tools/kbench.py --only attn --iters 30 2>&1 | grep "attn bwd"
done
echo -n "old: "; PLM_ATTN_BWD_V1=1 timeout 300 python tools/kbench.py --only attn --iters 30 2>&1 | grep "attn bwd"

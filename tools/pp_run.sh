#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
for m in 0 4 8 16 24; do echo "== MAP=$m"; PLM_ATTN_MAP=$m PLM_ATTN_PP=0 timeout 300 python tools/kbench.py --only attn --iters 30 2>&1 | grep -i "attn fwd\|attn bwd"; done
echo "== tests MAP=8"; PLM_ATTN_MAP=8 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -3
} > gpurun_out/pp_run.log 2>&1

#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
AT='attention or attn'
for v in 43 44; do
  echo "=== variants fwd=$v"
  PLM_ATTN_FWD=$v PLM_ATTN_DQ=0 PLM_ATTN_DKDV=0 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "$AT" 2>&1 | tail -6
done
echo "=== attn_ab"
timeout 600 python tools/attn_ab.py --fwd 0,34,43,44,3243,1643,6443,12643,34,43,44 --dq 0 --dkdv 0 2>&1 | tee gpurun_out/r3d_attn_ab.txt
timeout 600 python tools/attn_ab.py --fwd 0,34,43,44 --dq 0 --dkdv 0 --B 8 --T 2048 --nh 16 2>&1 | tee -a gpurun_out/r3d_attn_ab.txt
timeout 600 python tools/attn_ab.py --fwd 0,34,43 --dq 0 --dkdv 0 --doc 2>&1 | tee -a gpurun_out/r3d_attn_ab.txt

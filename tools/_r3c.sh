#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
AT='attention or attn'
for v in 32 33 34; do
  echo "=== variants fwd=$v"
  PLM_ATTN_FWD=$v PLM_ATTN_DQ=0 PLM_ATTN_DKDV=0 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "$AT" 2>&1 | tail -4
done
echo "=== attn_ab"
timeout 900 python tools/attn_ab.py --fwd 0,22,32,33,34,3234,1634,6434,12634,22,34 --dq 0 --dkdv 0 2>&1 | tee gpurun_out/r3c_attn_ab.txt
timeout 900 python tools/attn_ab.py --fwd 0,22,32,33,34 --dq 0 --dkdv 0 --B 8 --T 2048 --nh 16 2>&1 | tee -a gpurun_out/r3c_attn_ab.txt

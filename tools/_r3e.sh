#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/attn_ab.py --fwd 0,34,43,12843,25643,38443,1643,3243,43,34 --dq 0 --dkdv 0 2>&1 | tee gpurun_out/r3e_attn_ab.txt

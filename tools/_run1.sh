cd /root/repo; mkdir -p gpurun_out
{
echo "== tests with PLM_DUO_DBG=16"
PLM_DUO_DBG=16 timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or swiglu or rope or fc1 or fc2 or qkv" 2>&1 | tail -3
for v in 0 16 0 16; do echo "== PLM_DUO_DBG=$v"; PLM_DUO_DBG=$v timeout 900 python tools/kbench.py --only gemm --iters 20 2>&1 | grep -E '"nt (qkv fwd|out fwd|fc1 fwd|fc2 fwd|head fwd|dX qkv|dX fc1|dX fc2|dX head)"|epilogue'; done
} > gpurun_out/run1.log 2>&1

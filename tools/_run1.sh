cd /root/repo; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "swiglu" 2>&1 | tail -3
timeout 600 python tools/kbench.py --only hbm --iters 30 2>&1 | grep -i "swiglu"
timeout 900 python tools/kbench.py --only gemm --iters 20 2>&1 | grep -i "swiglu\|nt dX fc2\|nt fc1 fwd\""
} > gpurun_out/run1.log 2>&1

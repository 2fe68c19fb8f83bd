#!/bin/bash
# Does the GPU performance guard fire?  Builds the library with -DPLM_DBG_DRAIN (every counted LDS-DMA wait of the persistent GEMM kernels becomes
# a draining s_waitcnt vmcnt(0): numerically identical, ~20 % slower - the defect hipcc produced on its own in rounds 2 and 3) into
# tools/_lib_drain.so, then (on a GPU box) runs tests/test_perf_guard_gpu.py against it: the test must FAIL on the GEMM sections.
#   here:        bash tools/perf_guard_selfcheck.sh build
#   on the box:  gpurun -- 'bash tools/perf_guard_selfcheck.sh run'
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  set -e
  d=tools/_drain_build; mkdir -p $d
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function"
  /opt/rocm/bin/hipcc $F -DPLM_DBG_DRAIN -c plainlm_amd/csrc/gemm_big.hip -o $d/gemm_big.o
  objs=$(ls plainlm_amd/csrc/*.o | grep -v gemm_big.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $d/gemm_big.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o tools/_lib_drain.so
  echo built tools/_lib_drain.so
else
  echo "== the shipped library: must pass"
  python -m pytest tests/test_perf_guard_gpu.py -q -m gpu -s 2>&1 | tail -25
  echo "== the draining build: must FAIL"
  PLM_PERF_GUARD_LIB=tools/_lib_drain.so python -m pytest tests/test_perf_guard_gpu.py -q -m gpu -s 2>&1 | tail -25
fi

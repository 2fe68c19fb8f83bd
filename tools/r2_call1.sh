#!/bin/bash
# round-2 GPU call 1: parity diagnostics + whole GPU suite + baseline bench / kernel trace on this box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/diag_engine_parity.py > gpurun_out/diag1.log 2>&1
echo "diag rc=$?"
python -m pytest tests -m gpu -q --timeout 2400 -rf > gpurun_out/t1.log 2>&1
echo "pytest rc=$?"
tail -40 gpurun_out/t1.log
python bench.py > gpurun_out/bench_r2_0.json 2> gpurun_out/bench_r2_0.err
echo "bench rc=$?"
cat gpurun_out/bench_r2_0.json | head -c 600

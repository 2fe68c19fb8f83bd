#!/bin/bash
# Whole GPU suite + default bench on a fresh box:  gpurun -- 'bash tools/gpu_check.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 2400 -rf > gpurun_out/gpu_check_tests.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/gpu_check_tests.log
python bench.py > gpurun_out/gpu_check_bench.json 2> gpurun_out/gpu_check_bench.err; echo "bench rc=$?"; head -c 400 gpurun_out/gpu_check_bench.json

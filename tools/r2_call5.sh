#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/c5
./tools/ubench/overlap > gpurun_out/c5/overlap.txt 2>&1; echo "overlap rc=$?"; cat gpurun_out/c5/overlap.txt

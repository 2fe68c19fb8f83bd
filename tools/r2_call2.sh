#!/bin/bash
# round-2 GPU call 2: SQ counters for every MFMA kernel, fixed tests, 2-rank plumbing run of bench.py on one device
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
python -m pytest tests/test_ddp_gpu.py tests/test_model_gpu.py -m gpu -q -s --timeout 2400 -k "two_rank or nan_check or engine_loss or 160m or 420m or checkpoint" > gpurun_out/t2.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|rel err|rel |drift|oracle:|final param" gpurun_out/t2.log | tail -40
python bench.py --gpus 2 --single-device --steps 3 --warmup 1 --no-extras > gpurun_out/bench_2rank_plumbing.json 2> gpurun_out/bench_2rank_plumbing.err
echo "2-rank plumbing rc=$?"; tail -c 900 gpurun_out/bench_2rank_plumbing.json
P=gpurun_out/pmc2
python3 tools/prof_kernels.py > $P/order.log 2>&1; echo "plain rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $P/mfma -o m --output-format csv -- python3 tools/prof_kernels.py > $P/mfma.log 2>&1; echo "mfma rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace -d $P/sq -o s --output-format csv -- python3 tools/prof_kernels.py > $P/sq.log 2>&1; echo "sq rc=$?"
python3 tools/pmc_report.py $P/order.log $P/report_sq.json $P/mfma $P/sq > $P/report_sq.txt 2>&1; echo "report rc=$?"
cat $P/report_sq.txt | cut -c1-400
find $P -name "*.csv" -size +3M -delete

#!/bin/bash
# rocprofv3 --pmc passes over one launch of every MFMA kernel at the shapes of the two other single-GPU configs (tools/prof_kernels.py with
# PLM_PROF_CONFIG):  gpurun -- 'bash tools/profile_configs_pmc.sh r05'   ->  gpurun_out/<tag>/pmc_{420m,160m_b8}.txt
TAG=${1:-r05}
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for cfg in 420m 160m_b8; do
  export PLM_PROF_CONFIG=$cfg
  python3 $R/tools/prof_kernels.py > $O/order_$cfg.log 2>&1; echo "order $cfg rc=$?"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$cfg -o f --output-format csv -- python3 $R/tools/prof_kernels.py > $O/fetch_$cfg.log 2>&1; echo "fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$cfg -o w --output-format csv -- python3 $R/tools/prof_kernels.py > $O/write_$cfg.log 2>&1; echo "write rc=$?"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O/mfma_$cfg -o m --output-format csv -- python3 $R/tools/prof_kernels.py > $O/mfma_$cfg.log 2>&1; echo "mfma rc=$?"
  python3 $R/tools/pmc_report.py $O/order_$cfg.log $O/pmc_$cfg.json $O/fetch_$cfg $O/write_$cfg $O/mfma_$cfg > $O/pmc_$cfg.txt 2>&1; echo "report rc=$?"
  head -40 $O/pmc_$cfg.txt | cut -c1-170
done
find $O -name "*.csv" -size +1M -delete

#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 -L > $R/gpurun_out/r3f_counters.txt 2>&1
cat > /tmp/drv.py <<PY
import os, sys, torch
sys.path.insert(0, '$R')
from plainlm_amd import ops
from plainlm_amd.transformer import rope_tables
B, T, nh, d = 32, 1024, 12, 768
qkv = torch.randn(B * T, 3 * d, device='cuda').to(torch.bfloat16)
for v in (0, 34, 43, 1643, 12843):
  os.environ['PLM_ATTN_FWD'] = str(v); ops.reload_env()
  for _ in range(3):
    ops.attn_fwd(qkv, B, T, nh)
torch.cuda.synchronize()
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
P4="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/r3f_p$i -o p --output-format csv -- python3 /tmp/drv.py > $R/gpurun_out/r3f_p$i.log 2>&1; echo "pass $i rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections
for i in (1,2,3,4):
  fs = glob.glob('gpurun_out/r3f_p%d/**/*counter_collection.csv' % i, recursive=True)
  if not fs: print('pass', i, 'no csv'); continue
  agg = collections.OrderedDict()
  for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'][:60]
    if 'attn' not in k: continue
    d = agg.setdefault(k, collections.OrderedDict())
    c = r['Counter_Name']; v = float(r['Counter_Value'])
    d.setdefault(c, []).append(v)
  for k, d in agg.items():
    print(k)
    print('   ', ' '.join('%s=%.3g' % (c, sum(v[-1:]) ) for c, v in d.items()))
PY
find gpurun_out -name "*.csv" -size +2M -delete

#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
AT='attention or attn or rope'
for v in "22 22 21" "21 21 21" "12 12 21"; do
  set -- $v
  echo "=== variants fwd=$1 dq=$2 dkdv=$3"
  PLM_ATTN_FWD=$1 PLM_ATTN_DQ=$2 PLM_ATTN_DKDV=$3 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "$AT" 2>&1 | tail -4
done
echo "=== attn_ab"
timeout 900 python tools/attn_ab.py --fwd 0,12,22,21,112,612,812,1612,2412,3212,6412,6512,7012,7812,12612,122,622,822,1622,2422,3222,6422,6522,7022,7822,12622 2>&1 | tee gpurun_out/r3b_attn_ab.txt
R=$PWD
cd /tmp
echo "=== kernel trace"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3b_trace -o t --output-format csv -- python3 $R/tools/attn_ab.py --fwd 0,12,22 --dq 0,22,21,12 --dkdv 0,21 > $R/gpurun_out/r3b_trace.log 2>&1; echo "trace rc=$?"
cd $R
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob('gpurun_out/r3b_trace/**/*kernel_stats.csv', recursive=True):
  rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:30]:
  print('%-100s calls %6s avg_us %9.1f' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3))
PY
find gpurun_out/r3b_trace -name "*.csv" -size +1M -delete

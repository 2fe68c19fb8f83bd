#!/bin/bash
# Kernel trace of bench.py with and without the 1-rank data plane (PLM_FORCE_REDUCER=1, no reserve):  gpurun -- 'bash tools/ddp_trace.sh'
# -> gpurun_out/r05d/trace_plain.txt, trace_reducer_cap0.txt (profiles/r05_ddp_trace_1gpu.txt is their digest).  The environment is exported in THIS shell:
# nothing but the program itself may stand behind rocprofv3's "--".
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
summ() {
python3 - "$1" "$2" <<PY
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
  rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(sys.argv[2], 'w') as out:
  out.write('# total kernel time %.2f ms over 13 steps = %.3f ms/step\n' % (tot / 1e6, tot / 13e6))
  for r in rows[:40]:
    out.write('%-90s calls %6s  total_ms %9.3f  avg_us %9.1f  %5.1f%%\n' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
}
rocprofv3 --kernel-trace --stats -d $O/t_plain -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras > $O/t_plain.log 2>&1
export PLM_FORCE_REDUCER=1 PLM_BENCH_AUTOTUNE=0 PLM_COMM_CUS=0
rocprofv3 --kernel-trace --stats -d $O/t_red -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras > $O/t_red.log 2>&1
summ $O/t_plain $O/trace_plain.txt; summ $O/t_red $O/trace_reducer_cap0.txt
find $O -name "*.csv" -size +1M -delete
head -25 $O/trace_plain.txt | cut -c1-170; head -30 $O/trace_reducer_cap0.txt | cut -c1-170

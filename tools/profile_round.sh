#!/bin/bash
# Evidence run of a round on one MI355X box:  gpurun -- 'bash tools/profile_round.sh r03'
# kernel trace of bench.py, PMC passes (HBM-side traffic, MFMA utilisation, issue / stall counters) over every MFMA kernel of the
# step, per-kernel micro-benchmarks, the bench line itself.  Summaries land in gpurun_out/<tag>/ ; copy what is judged into profiles/.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/kbench.py --iters 20 --json $O/kbench.jsonl > $O/kbench.log 2>&1; echo "kbench rc=$?"
python tools/engine_bench.py > $O/engine_bench.txt 2>&1; echo "engine_bench rc=$?"
python bench.py --config 420m --no-extras > $O/bench_420m.json 2> $O/bench_420m.err; echo "bench 420m rc=$?"
python bench.py --doc-mask --no-extras > $O/bench_docmask.json 2> $O/bench_docmask.err; echo "bench doc-mask rc=$?"
python bench.py --doc-mask --micro-batch 8 --no-extras > $O/bench_docmask_b8.json 2> $O/bench_docmask_b8.err; echo "bench doc-mask B=8 rc=$?"
PLM_FORCE_REDUCER=1 python bench.py --no-extras > $O/bench_force_reducer.json 2> $O/bench_force_reducer.err; echo "bench 1-GPU reducer what-if rc=$?"
PLM_FORCE_REDUCER=1 PLM_COMM_MODEL_GBPS=60 python bench.py --no-extras > $O/bench_force_reducer_60gbps.json 2> $O/bench_force_reducer_60gbps.err; echo "bench 1-GPU reducer what-if, windows at 60 GB/s rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras > $R/$O/trace.log 2>&1; echo "trace rc=$?"
python3 $R/tools/prof_kernels.py > $R/$O/order.log 2>&1; echo "order rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/fetch -o f --output-format csv -- python3 $R/tools/prof_kernels.py > $R/$O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/write -o w --output-format csv -- python3 $R/tools/prof_kernels.py > $R/$O/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $R/$O/mfma -o m --output-format csv -- python3 $R/tools/prof_kernels.py > $R/$O/mfma.log 2>&1; echo "mfma rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC --kernel-trace -d $R/$O/sq -o s --output-format csv -- python3 $R/tools/prof_kernels.py > $R/$O/sq.log 2>&1; echo "sq rc=$?"
cd $R
python3 tools/pmc_report.py $O/order.log $O/pmc.json $O/fetch $O/write $O/mfma $O/sq > $O/pmc.txt 2>&1; echo "report rc=$?"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob('$O/trace/**/*kernel_stats.csv', recursive=True):
  rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('$O/kernel_trace_summary.txt', 'w') as out:
  out.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-extras  (13 steps)\n# name, calls, total ms, avg us, share\n')
  for r in rows[:40]:
    out.write('%-90s calls %6s  total_ms %9.3f  avg_us %9.1f  %5.1f%%\n' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print(open('$O/kernel_trace_summary.txt').read()[:3000])
PY
find $O -name "*.csv" -size +1M -delete
head -c 600 $O/bench.json; echo; head -30 $O/pmc.txt | cut -c1-200

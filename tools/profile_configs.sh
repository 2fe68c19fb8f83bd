#!/bin/bash
# Evidence run for the two single-GPU configs beside the headline:  gpurun -- 'bash tools/profile_configs.sh r05'
#   420m       = BASELINE configs[3], tr_420M_x8gpu.yaml:20-24,34 (24L d=1024 nh=16 T=2048, B=8 per GPU)
#   docmask_b8 = BASELINE configs[4] at the reference's own micro-batch, config_doc_mask.yaml:35-36 (160M, B=8, document masks)
# For each: the bench line WITH roofline, a rocprofv3 kernel trace, and kbench at its shapes.  Summaries land in gpurun_out/<tag>/.
TAG=${1:-r05}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python bench.py --config 420m > $O/bench_420m.json 2> $O/bench_420m.err; echo "bench 420m rc=$?"
python bench.py --doc-mask --micro-batch 8 > $O/bench_docmask_b8.json 2> $O/bench_docmask_b8.err; echo "bench doc-mask B=8 rc=$?"
# the reference's configs AS SHIPPED: seq_len 2048 for the 160M model (config/config.yaml:9,33: micro-batch 32; config_doc_mask.yaml:9,35: micro-batch 8 + document masks)
python bench.py --seq-len 2048 --no-extras > $O/bench_160m_t2048.json 2> $O/bench_160m_t2048.err; echo "bench 160M T=2048 B=32 rc=$?"
python bench.py --seq-len 2048 --doc-mask --micro-batch 8 --no-extras > $O/bench_160m_t2048_docmask_b8.json 2> $O/bench_160m_t2048_docmask_b8.err; echo "bench 160M T=2048 doc masks B=8 rc=$?"
python tools/kbench.py --iters 10 --T 2048 --only gemm,attn,hbm --json $O/kbench_160m_t2048.jsonl > $O/kbench_160m_t2048.log 2>&1; echo "kbench 160M T=2048 rc=$?"
python tools/kbench.py --iters 20 --config 420m --only gemm,attn,hbm --json $O/kbench_420m.jsonl > $O/kbench_420m.log 2>&1; echo "kbench 420m rc=$?"
python tools/kbench.py --iters 20 --B 8 --doc-mask --only gemm,attn,hbm --json $O/kbench_docmask_b8.jsonl > $O/kbench_docmask_b8.log 2>&1; echo "kbench B=8 rc=$?"
summ() {  # $1 = trace dir, $2 = output file, $3 = command line text
python3 - "$1" "$2" "$3" <<PY
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
  rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(sys.argv[2], 'w') as out:
  out.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py %s --steps 10 --warmup 3 --no-extras  (13 steps; total kernel time %.1f ms)\n# name, calls, total ms, avg us, share\n' % (sys.argv[3], tot / 1e6))
  for r in rows[:45]:
    out.write('%-100s calls %6s  total_ms %9.3f  avg_us %9.1f  %5.1f%%\n' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print(open(sys.argv[2]).read()[:2500])
PY
}
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/trace_420m -o bench --output-format csv -- python3 $R/bench.py --config 420m --steps 10 --warmup 3 --no-extras > $R/$O/trace_420m.log 2>&1; echo "trace 420m rc=$?"
rocprofv3 --kernel-trace --stats -d $R/$O/trace_docmask_b8 -o bench --output-format csv -- python3 $R/bench.py --doc-mask --micro-batch 8 --steps 10 --warmup 3 --no-extras > $R/$O/trace_docmask_b8.log 2>&1; echo "trace docmask_b8 rc=$?"
cd $R
summ $O/trace_420m $O/kernel_trace_420m.txt "--config 420m"
summ $O/trace_docmask_b8 $O/kernel_trace_docmask_b8.txt "--doc-mask --micro-batch 8"
find $O -name "*.csv" -size +1M -delete
head -c 400 $O/bench_420m.json; echo; head -c 400 $O/bench_docmask_b8.json; echo

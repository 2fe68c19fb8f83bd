#!/bin/bash
# round-2 GPU call 3: 8-wave ping-pong dK/dV kernel - correctness, then A/B against the 4-wave kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/c3
O=gpurun_out/c3
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x -k "attention or loss_and_grads or docmask or two_rank" > $O/t.log 2>&1
echo "pytest rc=$?"; tail -5 $O/t.log
timeout 300 python -m pytest tests/test_ddp_gpu.py -m gpu -q --timeout 600 > $O/t_ddp.log 2>&1; echo "ddp rc=$?"; tail -3 $O/t_ddp.log
for v in new old new old; do
  if [ $v = old ]; then export PLM_ATTN_BWD_V1=1; else unset PLM_ATTN_BWD_V1; fi
  echo "== $v"; timeout 300 python tools/kbench.py --only attn --iters 30 2>&1 | grep -i attn
done
unset PLM_ATTN_BWD_V1
echo "== T=2048 B=8 new / old"
timeout 300 python tools/kbench.py --only attn --iters 30 --B 8 --T 2048 2>&1 | grep -i attn
PLM_ATTN_BWD_V1=1 timeout 300 python tools/kbench.py --only attn --iters 30 --B 8 --T 2048 2>&1 | grep -i attn
cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o k --output-format csv -- python3 $OLDPWD/tools/kbench.py --only attn --iters 20 > $OLDPWD/$O/prof.log 2>&1; cd $OLDPWD
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/c3/prof/**/*kernel_stats.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'attn' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
for v in new old new old; do
  if [ $v = old ]; then export PLM_ATTN_BWD_V1=1; else unset PLM_ATTN_BWD_V1; fi
  echo -n "bench $v: "; timeout 600 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
done
find $O/prof -name "*.csv" -size +2M -delete

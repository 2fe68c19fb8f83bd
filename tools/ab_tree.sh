#!/bin/bash
# Same-box A/B of two TREES (when the C ABI or the Python side differs between them, swapping the .so alone is not enough):
#   here:       rm -rf _ab_head && mkdir _ab_head && git archive <rev> | tar -x -C _ab_head && make -C _ab_head/plainlm_amd/csrc
#               (_ab_head/ is git-ignored but travels with gpurun)
#   on the box: gpurun -- 'bash tools/ab_tree.sh 3 --doc-mask --micro-batch 8'
# Prints ms per step of `bench.py --steps 20 --warmup 5 --no-extras <args>` for _ab_head, this tree, _ab_head, ...
N=$1; shift
cd "$(dirname "$0")/.."
for i in $(seq "$N"); do
  for t in _ab_head .; do
    echo -n "$t  "
    (cd $t && python bench.py --steps 20 --warmup 5 --no-extras "$@" 2>/dev/null | grep '"metric"' | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\2 ms  \1 tok\/s/')
  done
done

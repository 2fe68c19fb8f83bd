"""Square-GEMM yardstick (4096^3 / 8192^3, uniform random [-1, 1) operands): every NT kernel variant vs the vendor library."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from plainlm_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

BF = torch.bfloat16
VARIANTS = ((0, 'auto'), (3, 'plain256'), (4, 'deep256'), (-1, 'vendor'))


def main():
  sizes = [int(v) for v in sys.argv[1:]] or [4096, 8192]
  for n in sizes:
    A = (torch.rand(n, n, device='cuda') * 2 - 1).to(BF)
    B = (torch.rand(n, n, device='cuda') * 2 - 1).to(BF)
    out = torch.empty(n, n, device='cuda', dtype=BF)
    fl = 2.0 * n ** 3
    res = {}
    for rep in range(2):
      for v, name in VARIANTS:
        if v < 0:
          Bt = B.t().contiguous()
          t = timeit(lambda: torch.matmul(A, Bt, out=out), 30)
        else:
          t = timeit(lambda: ops.gemm_nt(A, B, out=out, variant=v), 30)
        res.setdefault(name, []).append(round(fl / t / 1e9, 1))
    print(n, json.dumps(res), flush=True)


if __name__ == '__main__':
  main()

"""Launch every GEMM shape of the 160M step twice (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes).
The analysis side (tools/pmc_traffic_report.py) maps counter rows to shapes by dispatch order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from plainlm_amd import ops  # noqa: E402

BF = torch.bfloat16
B, T, d, h, V = 32, 1024, 768, 2048, 50280
M = B * T
NT = {'nt qkv fwd': (M, 3 * d, d), 'nt out fwd': (M, d, d), 'nt fc1 fwd': (M, 2 * h, d), 'nt fc2 fwd': (M, d, h), 'nt head fwd': (M, V, d),
      'nt dX qkv': (M, d, 3 * d), 'nt dX fc1': (M, d, 2 * h), 'nt dX fc2': (M, h, d), 'nt dX head': (M, d, 50304)}
TN = {'tn dW qkv': (3 * d, d, M), 'tn dW out': (d, d, M), 'tn dW fc1': (2 * h, d, M), 'tn dW fc2': (d, h, M), 'tn dW head': (V, d, M)}


def main():
  order = []
  for name, (m, n, k) in NT.items():
    A = torch.randn(m, k, device='cuda').to(BF)
    Bm = torch.randn(n, k, device='cuda').to(BF)
    out = torch.empty(m, n, device='cuda', dtype=BF)
    torch.cuda.synchronize()
    for _ in range(2):
      ops.gemm_nt(A, Bm, out=out)
    torch.cuda.synchronize()
    order.append((name, m, n, k, 2.0 * (m * k + n * k + m * n)))
    del A, Bm, out
  for name, (m, n, k) in TN.items():
    A = torch.randn(k, m, device='cuda').to(BF)
    Bm = torch.randn(k, n, device='cuda').to(BF)
    out = torch.zeros(m, n, device='cuda')
    torch.cuda.synchronize()
    for _ in range(2):
      ops.gemm_tn(A, Bm, out=out, accumulate=True)
    torch.cuda.synchronize()
    order.append((name, m, n, k, 2.0 * (m * k + n * k) + 8.0 * m * n))
    del A, Bm, out
  import json
  print('ORDER ' + json.dumps(order))


if __name__ == '__main__':
  main()

"""Tiny driver for counter collection: a few launches of the dW (tn) and fwd (nt) GEMMs at 160M shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops
BF = torch.bfloat16
M, d = 32768, 768
A = torch.randn(M, 3 * d, device='cuda').to(BF)
X = torch.randn(M, d, device='cuda').to(BF)
out = torch.zeros(3 * d, d, device='cuda')
W = torch.randn(3 * d, d, device='cuda').to(BF)
y = torch.empty(M, 3 * d, device='cuda', dtype=BF)
for _ in range(3):
  ops.gemm_tn(A, X, out=out, accumulate=True)     # big-tile tn (split-K)
  os.environ['PLM_TN_NO_BIG'] = '1'
  ops.gemm_tn(A, X, out=out, accumulate=True)     # 128x128 tn dma
  del os.environ['PLM_TN_NO_BIG']
  ops.gemm_nt(X, W, out=y, variant=3)
  ops.gemm_nt(X, W, out=y, variant=2)
torch.cuda.synchronize()

"""Tiny driver for counter collection on the attention kernels at the 160M shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops
from plainlm_amd.transformer import rope_tables
B, T, nh, d = 32, 1024, 12, 768
cos, sin = (t.cuda() for t in rope_tables(64, T))
qkv = torch.randn(B * T, 3 * d, device='cuda').to(torch.bfloat16)
dout = torch.randn(B * T, d, device='cuda').to(torch.bfloat16)
for _ in range(3):
  out, lse = ops.attn_fwd(qkv, B, T, nh)
  ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh)
torch.cuda.synchronize()

"""dK/dV ping-pong kernel: packed-fp32 build (PLM_ATTN_PP=1) against the scalar build (=3) bit by bit, and both against the generation-two kernel (=0)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
  import torch
  from plainlm_amd import ops
  from plainlm_amd.transformer import rope_tables
  B, T, nh = 4, 1024, 12
  torch.manual_seed(1)
  dev = 'cuda'
  cos, sin = (t.to(dev) for t in rope_tables(64, T))
  qkv = torch.randn(B * T, 3 * nh * 64, device=dev).bfloat16()
  dout = torch.randn(B * T, nh * 64, device=dev).bfloat16()
  out, lse = ops.attn_fwd(qkv, B, T, nh)
  r = ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh)
  torch.save(r.cpu(), sys.argv[1])
else:
  import torch
  res = {}
  for v in ('0', '1', '3'):
    subprocess.run([sys.executable, __file__, f'/tmp/pp_{v}.pt'], env=dict(os.environ, PLM_ATTN_PP=v), check=True)
    res[v] = torch.load(f'/tmp/pp_{v}.pt').float()
  print('packed vs scalar: bit-identical =', bool((res['1'] == res['3']).all()), ' max abs diff', float((res['1'] - res['3']).abs().max()))
  print('ping-pong vs generation two: max abs diff', float((res['1'] - res['0']).abs().max()), ' rel (fro)', float((res['1'] - res['0']).norm() / res['0'].norm()))

"""Yardstick only (never on the product path): vendor bf16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) vs this repo's kernels
at the 160M shapes, same process, same random data.  Answers "what does the platform reach on THESE shapes"."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops
BF = torch.bfloat16
M, d, h, V = 32768, 768, 2048, 50280

def timeit(fn, iters=20, warm=3):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / iters

rows = []
for name, (m, n, k) in {'qkv fwd': (M, 3 * d, d), 'out fwd': (M, d, d), 'fc1 fwd': (M, 2 * h, d), 'fc2 fwd': (M, d, h), 'head fwd': (M, V, d),
                        'dX qkv': (M, d, 3 * d), 'dX fc1': (M, d, 2 * h), 'dX fc2': (M, h, d), 'dX head': (M, d, 50304)}.items():
  A = torch.randn(m, k, device='cuda').to(BF); B = torch.randn(n, k, device='cuda').to(BF)
  out = torch.empty(m, n, device='cuda', dtype=BF)
  t_ref = timeit(lambda: torch.matmul(A, B.t(), out=out))
  t_own = timeit(lambda: ops.gemm_nt(A, B, out=out))
  fl = 2.0 * m * n * k
  r = {'gemm': 'nt ' + name, 'vendor_TF': round(fl / t_ref / 1e9, 1), 'ours_TF': round(fl / t_own / 1e9, 1)}
  rows.append(r); print(json.dumps(r), flush=True)
  del A, B, out
for name, (m, n, k) in {'dW qkv': (3 * d, d, M), 'dW out': (d, d, M), 'dW fc1': (2 * h, d, M), 'dW fc2': (d, h, M), 'dW head': (V, d, M)}.items():
  A = torch.randn(k, m, device='cuda').to(BF); B = torch.randn(k, n, device='cuda').to(BF)
  out32 = torch.zeros(m, n, device='cuda'); outb = torch.empty(m, n, device='cuda', dtype=BF)
  t_ref = timeit(lambda: torch.matmul(A.t(), B, out=outb))      # vendor: bf16 output, no accumulate
  t_own = timeit(lambda: ops.gemm_tn(A, B, out=out32, accumulate=True))  # ours: fp32 accumulate into the gradient
  fl = 2.0 * m * n * k
  r = {'gemm': 'tn ' + name, 'vendor_TF': round(fl / t_ref / 1e9, 1), 'ours_TF': round(fl / t_own / 1e9, 1)}
  rows.append(r); print(json.dumps(r), flush=True)
  del A, B, out32, outb
if len(sys.argv) > 1:
  json.dump(rows, open(sys.argv[1], 'w'), indent=1)

# ---- attention yardstick: torch SDPA (vendor flash kernels) vs ours, causal, B=32 nh=12 T=1024 hd=64 ----
import torch.nn.functional as F
from plainlm_amd.transformer import rope_tables
Bq, T, nh = 32, 1024, 12
q, k, v = (torch.randn(Bq, nh, T, 64, device='cuda', dtype=BF, requires_grad=True) for _ in range(3))
do = torch.randn(Bq, nh, T, 64, device='cuda', dtype=BF)
def sdpa_fwd(): return F.scaled_dot_product_attention(q, k, v, is_causal=True)
def sdpa_fb():
  o = F.scaled_dot_product_attention(q, k, v, is_causal=True); o.backward(do); q.grad = k.grad = v.grad = None
fl = 4.0 * Bq * nh * 64 * T * (T + 1) / 2
t_f = timeit(sdpa_fwd); t_fb = timeit(sdpa_fb)
cos, sin = (t.cuda() for t in rope_tables(64, T))
qkv = torch.randn(Bq * T, 3 * nh * 64, device='cuda').to(BF); dout = torch.randn(Bq * T, nh * 64, device='cuda').to(BF)
out, lse = ops.attn_fwd(qkv, Bq, T, nh)
o_f = timeit(lambda: ops.attn_fwd(qkv, Bq, T, nh)); o_b = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, cos, sin, Bq, T, nh))
r = {'attention': 'causal B32 nh12 T1024 hd64', 'vendor_fwd_TF': round(fl / t_f / 1e9, 1), 'vendor_fwd+bwd_ms': round(t_fb, 3),
     'ours_fwd_TF': round(fl / o_f / 1e9, 1), 'ours_fwd+bwd_ms': round(o_f + o_b, 3), 'note': 'ours includes inverse RoPE in bwd and reads q/k/v strided from the projection buffer'}
print(json.dumps(r)); rows.append(r)
if len(sys.argv) > 1:
  json.dump(rows, open(sys.argv[1], 'w'), indent=1)

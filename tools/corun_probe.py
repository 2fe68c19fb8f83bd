"""Do an HBM-bound kernel and an MFMA-bound GEMM overlap when they are launched as SEPARATE kernels on two HIP streams (GEMM first, on the
high-priority stream)?  Partners chosen so that the streaming kernel's waves fit into the registers the persistent GEMM workgroups leave free.
Usage (GPU box): python tools/corun_probe.py   ->  profiles/r04_corun_probe.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops
BF = torch.bfloat16
M, d, h, V = 32768, 768, 2048, 50280
dev = 'cuda'
# MFMA-bound partners: TN dW head (221 regs: 64 free per SIMD), NT dX head 256x192 (202 regs: 96 free), NT head fwd 256x256 (230: 48 free)
dl = torch.randn(M, 50304, device=dev).to(BF)
y = torch.randn(M, d, device=dev).to(BF)
Wt = (torch.randn(d, 50304, device=dev) * 0.02).to(BF)
W = (torch.randn(V, d, device=dev) * 0.02).to(BF)
dW = torch.zeros(V, d, device=dev)
dy = torch.empty(M, d, device=dev, dtype=BF)
lg = torch.empty(M, 50304, device=dev, dtype=BF)
gemms = {'tn dW head (64 regs free)': lambda: ops.gemm_tn(dl[:, :V], y, out=dW), 'nt dX head 256x192 (96 free)': lambda: ops.gemm_nt(dl, Wt, out=dy),
         'nt head fwd 256x256 (48 free)': lambda: ops.gemm_nt(y, W, out=lg[:, :V]), 'nt head fwd 256x192 (96 free)': lambda: ops.gemm_nt(y, W, out=lg[:, :V], variant=5)}
# HBM-bound partners: swiglu_bwd (39 regs, 256 threads), rmsnorm_fwd (34 regs)
u = torch.randn(M, 2 * h, device=dev).to(BF); do = torch.randn(M, h, device=dev).to(BF)
x = torch.randn(M, d, device=dev); w = torch.ones(d, device=dev)
def hb1():
  for _ in range(12): ops.swiglu_bwd(do, u)
def hb2():
  for _ in range(30): ops.rmsnorm_fwd(x, w, 1e-6, write_xout=True)
hbs = {'12 x swiglu_bwd (39 regs)': hb1, '30 x rmsnorm_fwd (34 regs)': hb2}
lo, hi_ = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
print('priority range', lo, hi_)
sB = torch.cuda.Stream(priority=0)
sA = torch.cuda.Stream(priority=-1)
def t(fa, fb, it=5):
  def once():
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    if fa is not None:
      with torch.cuda.stream(sA): fa()
    if fb is not None:
      with torch.cuda.stream(sB): fb()
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
  once(); once()
  return min(once() for _ in range(it))
for gn, g in gemms.items():
  ta = t(g, None)
  for hn, hfn in hbs.items():
    tb = t(None, hfn); tab = t(g, hfn)
    print(f'{gn:32s} {ta:.3f} ms | {hn:28s} {tb:.3f} ms | both on two streams {tab:.3f} ms = {tab / max(ta, tb):.2f} x max, {tab / (ta + tb):.2f} x sum', flush=True)

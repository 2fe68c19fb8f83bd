"""A/B of the attention kernel variants (PLM_ATTN_FWD / PLM_ATTN_DQ / PLM_ATTN_DKDV) at the step's shape: each variant is first
compared with the fp32 oracle on a small case and with variant 0 (the first-generation kernels) at full size, then timed.
Usage: python tools/attn_ab.py [--B 32] [--T 1024] [--nh 12] [--iters 20] [--fwd 0,22,21,12] [--dq 0,22,21,12] [--dkdv 0,21]
Under `rocprofv3 --kernel-trace --stats` the per-kernel averages separate dQ from dK/dV."""

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops  # noqa: E402
from plainlm_amd.transformer import rope_tables  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters, warmup=8):
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3  # us


def relmax(a, b):
  return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--B', type=int, default=32)
  ap.add_argument('--T', type=int, default=1024)
  ap.add_argument('--nh', type=int, default=12)
  ap.add_argument('--iters', type=int, default=40)
  ap.add_argument('--fwd', default='0,22,21,12')
  ap.add_argument('--dq', default='0,22,21,12')
  ap.add_argument('--dkdv', default='0,21')
  ap.add_argument('--doc', action='store_true')
  ap.add_argument('--reps', type=int, default=3, help='passes over the forward variants; the best time of each is reported')
  a = ap.parse_args()
  dev = 'cuda'
  B, T, nh = a.B, a.T, a.nh
  d, M = nh * 64, B * T
  torch.manual_seed(0)
  cos, sin = (t.to(dev) for t in rope_tables(64, T))
  qkv = torch.randn(M, 3 * d, device=dev).to(BF)
  dout = torch.randn(M, d, device=dev).to(BF)
  ops.rope_qk_(qkv, cos, sin, B, T, nh)
  ds = None
  if a.doc:
    import numpy as np
    from plainlm_amd.engine import doc_start_from_lengths
    rng = np.random.default_rng(1)
    docs = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
    ds = doc_start_from_lengths(docs, T).to(dev)
  fl = 2.0 * 2 * B * nh * T * (T + 1) / 2 * 64  # causal-counted QK^T + PV

  def setv(f, q, k):
    os.environ['PLM_ATTN_FWD'], os.environ['PLM_ATTN_DQ'], os.environ['PLM_ATTN_DKDV'] = str(f), str(q), str(k)
    ops.reload_env()

  setv(0, 0, 0)
  out0, lse0 = ops.attn_fwd(qkv, B, T, nh, ds)
  dqkv0 = ops.attn_bwd(qkv, out0, dout, lse0, cos, sin, B, T, nh, ds)
  torch.cuda.synchronize()
  for _ in range(300):  # clocks / power state settle over ~50 ms of load: the first variants of a cold run read 10 % slow
    ops.attn_fwd(qkv, B, T, nh, ds)
  fv = [int(x) for x in a.fwd.split(',') if x]
  best = {}
  for rep in range(a.reps):
    for f in fv:
      setv(f, 0, 0)
      us = timeit(lambda: ops.attn_fwd(qkv, B, T, nh, ds), a.iters)
      best[f] = min(best.get(f, 1e9), us)
  for f in fv:
    setv(f, 0, 0)
    out, lse = ops.attn_fwd(qkv, B, T, nh, ds)
    us = best[f]
    print(f'fwd variant {f:2d}: {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s   out vs v0 {relmax(out, out0):.2e}  lse max|diff| {(lse - lse0).abs().max().item():.2e}', flush=True)
  for q in [int(x) for x in a.dq.split(',') if x]:
    setv(0, q, 0)
    g = ops.attn_bwd(qkv, out0, dout, lse0, cos, sin, B, T, nh, ds)
    us = timeit(lambda: ops.attn_bwd(qkv, out0, dout, lse0, cos, sin, B, T, nh, ds), a.iters)
    print(f'bwd dq variant {q:2d} (+ dkdv v0): {us:7.1f} us   dq vs v0 {relmax(g[:, :d], dqkv0[:, :d]):.2e}  dk {relmax(g[:, d:2 * d], dqkv0[:, d:2 * d]):.2e}', flush=True)
  for k in [int(x) for x in a.dkdv.split(',') if x]:
    setv(0, 0, k)
    g = ops.attn_bwd(qkv, out0, dout, lse0, cos, sin, B, T, nh, ds)
    us = timeit(lambda: ops.attn_bwd(qkv, out0, dout, lse0, cos, sin, B, T, nh, ds), a.iters)
    print(f'bwd dkdv variant {k:2d} (+ dq v0): {us:7.1f} us   dk vs v0 {relmax(g[:, d:2 * d], dqkv0[:, d:2 * d]):.2e}  dv {relmax(g[:, 2 * d:], dqkv0[:, 2 * d:]):.2e}', flush=True)


if __name__ == '__main__':
  main()

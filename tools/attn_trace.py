"""Per-workgroup timeline of the document-mask attention kernels (forward, dQ, dK/dV).

Builds a copy of the library with -DPLM_ATTN_TRACE (tools/_lib_trace.so; the shipped .so never carries the instrumentation), runs one
forward + backward at (B, T, nh) with the bench's document lengths and prints, per kernel: the span of the launch, the ramp of workgroup
starts, per-workgroup duration against its cost (fit: prologue + per-tile time), the load of the busiest CUs, and where the grid's
first workgroups were placed.

  python tools/attn_trace.py [--B 8] [--T 1024] [--nh 12] [--split-min 8] [--build-only]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, 'tools', '_lib_trace.so')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-munsafe-fp-atomics', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-Wno-unused-function',
         '-DPLM_ATTN_TRACE']


def build():
  csrc = os.path.join(ROOT, 'plainlm_amd', 'csrc')
  objs = []
  out = os.path.join(ROOT, 'tools', '_trace_build')
  os.makedirs(out, exist_ok=True)
  procs = []
  for f in ('misc.hip', 'elementwise.hip', 'ce.hip', 'gemm.hip', 'gemm_big.hip', 'attn.hip', 'attn_causal.hip', 'attn_doc.hip'):
    o = os.path.join(out, f.replace('.hip', '.o'))
    objs.append(o)
    src = os.path.join(csrc, f)
    if os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(os.path.join(csrc, x)) for x in os.listdir(csrc) if x.endswith(('.hip', '.h'))):
      continue
    extra = ['-fno-slp-vectorize'] if f in ('attn_causal.hip', 'attn_doc.hip') else []
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc'] + FLAGS + extra + ['-c', src, '-o', o]))
  for p in procs:
    if p.wait() != 0:
      sys.exit('trace build failed')
  objs.append(os.path.join(csrc, 'comm.o'))
  subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '--offload-arch=gfx950'] + objs + ['-L/opt/rocm/lib', '-lrccl', '-Wl,-rpath,/opt/rocm/lib', '-o', TRACE_LIB],
                 check=True)


def cu_key(hw, xcc):
  return ((int(xcc) & 15) << 8) | (((int(hw) >> 13) & 7) << 5) | (((int(hw) >> 12) & 1) << 4) | ((int(hw) >> 8) & 15)


def report(name, rec):
  t0, t1, t2, t3 = (rec[:, i].astype(np.int64) for i in range(4))
  base = t0.min()
  us = lambda x: (x - base) / 100.0  # 100 MHz wall clock
  cu = np.array([cu_key(h, x) for h, x in zip(rec[:, 4], rec[:, 5])])
  kind = (rec[:, 7].astype(np.int64) >> 30) & 1
  cost = rec[:, 7].astype(np.int64) & ((1 << 30) - 1)
  dur = (t3 - t0) / 100.0
  print(f'== {name}: {len(rec)} workgroups on {len(set(cu))} CUs; launch span {us(t3.max()):.1f} us; starts: median {np.median(us(t0)):.1f} us, last {us(t0.max()):.1f} us')
  A = np.stack([np.ones_like(cost, dtype=float), cost.astype(float)], 1)
  fit = np.linalg.lstsq(A, dur, rcond=None)[0]
  print(f'   duration ~ {fit[0]:.2f} us + {fit[1]:.2f} us per tile step (cost: min {cost.min()} mean {cost.mean():.2f} max {cost.max()}; {int(kind.sum())} split items)')
  loads = {}
  for c, k, e in zip(cu, cost, us(t3)):
    n, s, last = loads.get(c, (0, 0, 0.0))
    loads[c] = (n + 1, s + int(k), max(last, e))
  per = np.array([v[1] for v in loads.values()])
  cnt = np.array([v[0] for v in loads.values()])
  if t1.min() > 0:
    early = us(t0) < 2.0
    for tag, m in (('first round', early), ('later rounds', ~early)):
      if m.any():
        print(f'   {tag}: start -> loop median {np.median((t1 - t0)[m]) / 100.0:.2f} us, loop {np.median((t2 - t1)[m]) / 100.0:.2f} us '
              f'({np.median(((t2 - t1) / np.maximum(cost, 1))[m]) / 100.0:.2f} per tile step), loop end -> end {np.median((t3 - t2)[m]) / 100.0:.2f} us')
  print(f'   per CU: workgroups min {cnt.min()} max {cnt.max()}; summed cost min {per.min()} mean {per.mean():.1f} max {per.max()}')
  worst = sorted(loads.items(), key=lambda kv: -kv[1][2])[:5]
  for c, (n, s, last) in worst:
    idx = np.nonzero(cu == c)[0]
    desc = ', '.join(f'wg {i} cost {cost[i]}{"s" if kind[i] else ""} [{us(t0[i]):.1f}-{us(t3[i]):.1f}]' for i in idx)
    print(f'   CU {c:#05x}: {n} workgroups, cost {s}, done at {last:.1f} us: {desc}')
  order = np.argsort(us(t3))[::-1][:5]
  print('   last to finish: ' + '; '.join(f'wg {i} (item {int(rec[i, 6])}, cost {cost[i]}{"s" if kind[i] else ""}, CU {cu[i]:#05x}) {us(t0[i]):.1f}-{us(t3[i]):.1f}' for i in order))
  first = ' '.join(f'{cu[i]:#05x}' for i in range(min(16, len(cu))))
  print(f'   CUs of workgroups 0..15: {first}')
  share = [len(set(cu[i::256][:3])) for i in range(min(256, len(cu)))]
  print(f'   workgroups i, i+256, i+512 on ONE CU for {sum(1 for s in share if s == 1)} of {len(share)} i')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--B', type=int, default=8)
  ap.add_argument('--T', type=int, default=1024)
  ap.add_argument('--nh', type=int, default=12)
  ap.add_argument('--split-min', type=int, default=8)
  ap.add_argument('--build-only', action='store_true')
  a = ap.parse_args()
  if a.build_only or not os.path.exists(TRACE_LIB):
    build()
    if a.build_only:
      return
  os.environ['PLM_ATTN_DOC_SPLIT_MIN'] = str(a.split_min)
  import torch
  from plainlm_amd import _lib
  _lib.LIB_PATH = TRACE_LIB
  from plainlm_amd import ops
  from plainlm_amd.engine import doc_start_from_lengths
  lib = _lib.load()
  B, T, nh = a.B, a.T, a.nh
  d = nh * 64
  dev = 'cuda'
  rng = np.random.default_rng(7)
  docs = []
  for _ in range(B):
    lens, tot = [], 0
    while tot < T + 1:
      n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
      lens.append(n)
      tot += n
    docs.append(lens)
  ds = doc_start_from_lengths(docs, T).to(dev)
  plan = ops.attn_doc_plan(ds, nh)
  hdr = plan[:2].cpu().numpy()
  qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
  dout = torch.randn(B * T, d, device=dev).to(torch.bfloat16)
  cos = torch.ones(T, 32, device=dev)
  sin = torch.zeros(T, 32, device=dev)
  buf = torch.zeros(3 * 65536 * 8, dtype=torch.int64, device=dev)
  for fn in ('plm_dbg_attn_trace_doc', 'plm_dbg_attn_trace_causal'):
    f = getattr(lib, fn)
    f.restype, f.argtypes = C.c_int, [C.c_void_p]
    assert f(C.c_void_p(buf.data_ptr())) == 0
  for _ in range(3):  # warm: the last pass is the one read back
    out, lse = ops.attn_fwd(qkv, B, T, nh, ds, plan)
    ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh, ds, plan)
  torch.cuda.synchronize()
  rec = buf.cpu().numpy().view(np.uint64).reshape(3, 65536, 8)
  for k, name in enumerate(('forward', 'backward dQ', 'backward dK/dV')):
    report(name, rec[k, :int(hdr[0 if k < 2 else 1]) * nh])


if __name__ == '__main__':
  main()

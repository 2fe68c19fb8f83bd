"""Timing-only ablation: what would a ONE-pass attention backward cost? (price the 5-product form before building it.)

Today's backward is two deterministic passes (dQ, then dK/dV: 7 matrix products per block where one pass needs 5).  A one-pass kernel is the dK/dV
kernel plus, per 32 x 32 block: dS handed from the key-owning lanes to query-owning lanes through LDS, 4 more MFMAs for the block's partial dQ, and
the accumulation of that partial dQ across key tiles (fp32 atomics - which also ends run-to-run bit equality - or an ordered semaphore chain, which
can only be slower).  This tool builds the library with -DPLM_ATTN_TRACE -DPLM_ATTN_ONEPASS_ABLATION (tools/_lib_onepass.so), runs the backward of a
batch whose rows are ONE document each (= the causal mask, through the document-mask kernels, whose per-workgroup trace gives each kernel's span) and
prints: dQ span, dK/dV span, and the dK/dV span with mode 1 (LDS hand-over + MFMAs) and mode 2 (+ atomics).  One-pass estimate = dK/dV span in mode 2;
it wins only if that is clearly below dQ + dK/dV.

  python tools/attn_onepass_ablation.py [--B 32] [--T 1024] [--nh 12] [--build-only]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', '_lib_onepass.so')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-munsafe-fp-atomics', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-Wno-unused-function',
         '-DPLM_ATTN_TRACE', '-DPLM_ATTN_ONEPASS_ABLATION', '-fno-slp-vectorize']


def build():
  csrc = os.path.join(ROOT, 'plainlm_amd', 'csrc')
  out = os.path.join(ROOT, 'tools', '_trace_build')
  os.makedirs(out, exist_ok=True)
  o = os.path.join(out, 'attn_causal_onepass.o')
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', os.path.join(csrc, 'attn_causal.hip'), '-o', o, '-save-temps=obj'], check=True, cwd=out)
  for f in os.listdir(out):  # register / spill report of the ablated kernel
    if f.startswith('attn_causal') and f.endswith('gfx950.s'):
      txt = open(os.path.join(out, f)).read()
      for blk in txt.split('  - .agpr_count:')[1:]:
        if 'dkdv_doc' in blk:
          import re
          g = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1))
          print('ablated dK/dV kernel: vgpr', g('vgpr_count'), 'spills', g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
  objs = [os.path.join(out, f) for f in ('misc.o', 'elementwise.o', 'ce.o', 'gemm.o', 'gemm_big.o', 'attn.o', 'attn_doc.o')]
  if not all(os.path.exists(x) for x in objs):
    sys.exit('run tools/attn_trace.py --build-only first (the other objects of the traced library)')
  subprocess.run(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '--offload-arch=gfx950'] + objs + [o, os.path.join(csrc, 'comm.o'), '-L/opt/rocm/lib', '-lrccl',
                  '-Wl,-rpath,/opt/rocm/lib', '-o', LIB], check=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--B', type=int, default=32)
  ap.add_argument('--T', type=int, default=1024)
  ap.add_argument('--nh', type=int, default=12)
  ap.add_argument('--build-only', action='store_true')
  a = ap.parse_args()
  if a.build_only or not os.path.exists(LIB):
    build()
    if a.build_only:
      return
  os.environ['PLM_ATTN_DOC_SPLIT_MIN'] = '0'
  import torch
  from plainlm_amd import _lib
  _lib.LIB_PATH = LIB
  from plainlm_amd import ops
  lib = _lib.load()
  B, T, nh = a.B, a.T, a.nh
  d, dev = nh * 64, 'cuda'
  ds = torch.zeros(B, T, dtype=torch.int32, device=dev)  # one document per row: the causal mask
  plan = ops.attn_doc_plan(ds, nh)
  n_wg = int(plan[1].item()) * nh
  qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
  dout = torch.randn(B * T, d, device=dev).to(torch.bfloat16)
  cos, sin = torch.ones(T, 32, device=dev), torch.zeros(T, 32, device=dev)
  buf = torch.zeros(3 * 65536 * 8, dtype=torch.int64, device=dev)
  dqbuf = torch.zeros(B * T, d, device=dev)
  for fn in ('plm_dbg_attn_trace_doc', 'plm_dbg_attn_trace_causal'):
    f = getattr(lib, fn)
    f.restype, f.argtypes = C.c_int, [C.c_void_p]
    assert f(C.c_void_p(buf.data_ptr())) == 0
  lib.plm_dbg_attn_onepass.restype, lib.plm_dbg_attn_onepass.argtypes = C.c_int, [C.c_void_p, C.c_int]
  out, lse = ops.attn_fwd(qkv, B, T, nh, ds, plan)
  res = {}
  for mode in (0, 1, 2, 0):
    assert lib.plm_dbg_attn_onepass(C.c_void_p(dqbuf.data_ptr()), mode) == 0
    spans = []
    for _ in range(5):
      ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh, ds, plan)
      torch.cuda.synchronize()
      rec = buf.cpu().numpy().view(np.uint64).reshape(3, 65536, 8)
      spans.append([(rec[k, :n_wg, 3].astype(np.int64).max() - rec[k, :n_wg, 0].astype(np.int64).min()) / 100.0 for k in (1, 2)])
    res.setdefault(mode, []).append(np.median(np.array(spans), axis=0))
  base = np.mean(res[0], axis=0)
  print(f'(B, T, heads) = ({B}, {T}, {nh}), one document per row (causal), {n_wg} workgroups per kernel; spans in us (median of 5 launches)')
  print(f'two-pass backward today:   dQ {base[0]:.1f} + dK/dV {base[1]:.1f} = {base.sum():.1f}')
  for mode, what in ((1, 'dK/dV + dS through LDS + 4 MFMAs per block'), (2, 'dK/dV + ... + fp32 atomics of the partial dQ (one-pass estimate)')):
    s = res[mode][0][1]
    print(f'mode {mode}: {what}: {s:.1f}  ({100.0 * (s - base.sum()) / base.sum():+.1f} % against the two passes)')


if __name__ == '__main__':
  main()

"""Per-kernel micro-benchmarks at the shapes of a BASELINE config: 160M (default: B=32, T=1024, d=768, h=2048, V=50280; --B 8 = the
reference's document-mask micro-batch, config_doc_mask.yaml:35) or --config 420m (B=8, T=2048, d=1024, nh=16, h=2816; tr_420M_x8gpu.yaml).
Prints achieved TFLOP/s (MFMA kernels) or GB/s of ALGORITHMIC bytes (HBM kernels).
Usage: python tools/kbench.py [--iters 20] [--only gemm,attn,...] [--config 420m] [--B 8] [--doc-mask] [--variants]
"""

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops, _lib  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters, warmup=12):  # enough launches for clocks / caches to settle (the first rows used to read 4-8 % slow)
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters  # ms


def timeit_instep(fns, iters, between, warmup=6):
  """In-step conditions: rotate over several operand sets (nothing stays cache-resident from the previous launch of the same
  shape) and run an HBM-bound kernel between GEMM launches (as the norm / activation kernels of the step do); only the GEMM
  launches are timed, one event pair each."""
  for i in range(warmup):
    between()
    fns[i % len(fns)]()
  torch.cuda.synchronize()
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for i, (s, e) in enumerate(evs):
    between()
    s.record()
    fns[i % len(fns)]()
    e.record()
  torch.cuda.synchronize()
  return sum(s.elapsed_time(e) for s, e in evs) / iters


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--only', default='')
  ap.add_argument('--config', default='160m', choices=['160m', '420m'])
  ap.add_argument('--B', type=int, default=0)
  ap.add_argument('--T', type=int, default=0)
  ap.add_argument('--doc-mask', action='store_true', help='attention with document masks (mean document length 256) beside the causal kernels')
  ap.add_argument('--json', default='')
  ap.add_argument('--cu-reserve', type=int, default=0, help='CUs the persistent GEMMs leave free (what a data-parallel run sets during backward)')
  ap.add_argument('--variants', action='store_true', help='also time every NT kernel variant per shape')
  ap.add_argument('--instep', action='store_true', help='NT variants under in-step conditions: rotating operand sets + an HBM-bound kernel between launches')
  a = ap.parse_args()
  only = set(a.only.split(',')) if a.only else None
  if a.cu_reserve:
    ops.set_cu_reserve(a.cu_reserve)
  B, T, d, nh, h, V = {'160m': (32, 1024, 768, 12, 2048, 50280), '420m': (8, 2048, 1024, 16, 2816, 50280)}[a.config]
  B, T = a.B or B, a.T or T
  M = B * T
  n_params = 12 * (4 * d * d + 3 * d * h) * (1 if a.config == '160m' else 2) + 2 * V * d
  dev = 'cuda'
  rows = []

  def rec(name, ms, flops=None, bytes_=None):
    r = {'kernel': name, 'ms': round(ms, 4)}
    if flops:
      r['TFLOP/s'] = round(flops / ms / 1e9, 1)
    if bytes_:
      r['GB/s'] = round(bytes_ / ms / 1e6, 1)
    rows.append(r)
    print(json.dumps(r), flush=True)

  def want(k):
    return only is None or k in only

  if want('gemm'):
    for name, (m, n, k) in {
      'nt qkv fwd': (M, 3 * d, d), 'nt out fwd': (M, d, d), 'nt fc1 fwd': (M, 2 * h, d), 'nt fc2 fwd': (M, d, h),
      'nt head fwd': (M, V, d), 'nt dX qkv': (M, d, 3 * d), 'nt dX fc1': (M, d, 2 * h), 'nt dX fc2': (M, h, d),
      'nt dX head': (M, d, 50304)}.items():
      A = torch.randn(m, k, device=dev).to(BF)
      Bm = torch.randn(n, k, device=dev).to(BF)
      out = torch.empty(m, n, device=dev, dtype=BF)
      rec(name, timeit(lambda: ops.gemm_nt(A, Bm, out=out), a.iters), flops=2.0 * m * n * k)
      if _lib.load().plm_gemm_nt_workspace_bytes(m, n, k) > 0:
        os.environ['PLM_NT_NO_HYBRID'] = '1'
        ops.reload_env()
        rec(name + ' [no hybrid]', timeit(lambda: ops.gemm_nt(A, Bm, out=out), a.iters), flops=2.0 * m * n * k)
        del os.environ['PLM_NT_NO_HYBRID']
        ops.reload_env()
      if k % 64 == 0 and a.variants:
        for v, vn in ((2, 'dma128'), (4, 'deep256x256'), (5, 'deep256x192'), (6, 'deep256x128'), (7, 'deep128x192')):
          rec(f'{name} [{vn}]', timeit(lambda: ops.gemm_nt(A, Bm, out=out, variant=v), a.iters), flops=2.0 * m * n * k)
        rec(f'{name} [auto again]', timeit(lambda: ops.gemm_nt(A, Bm, out=out), a.iters), flops=2.0 * m * n * k)
      if k % 64 == 0 and a.instep and n < 10000:
        sets = [(torch.randn(m, k, device=dev).to(BF), torch.randn(n, k, device=dev).to(BF) * 0.02, torch.empty(m, n, device=dev, dtype=BF))
                for _ in range(4)]
        big = torch.randn(M, 4096, device=dev).to(BF)
        between = lambda: ops.swiglu_fwd(big)  # 268 MB read + 134 MB written: evicts L2 / most of the Infinity Cache
        for v, vn in ((0, 'auto'), (4, 'deep256x256'), (5, 'deep256x192'), (6, 'deep256x128'), (7, 'deep128x192')):
          fns = [(lambda X=X, W=W, O=O: ops.gemm_nt(X, W, out=O, variant=v)) for X, W, O in sets]
          rec(f'{name} [in-step {vn}]', timeit_instep(fns, a.iters, between), flops=2.0 * m * n * k)
        del sets, big
      del A, Bm, out
    # fc1 + SwiGLU: two launches vs the GEMM epilogue
    A = torch.randn(M, d, device=dev).to(BF)
    W1 = (torch.randn(2 * h, d, device=dev) * 0.02).to(BF)
    rec('nt fc1 fwd + swiglu fwd (2 launches)', timeit(lambda: ops.swiglu_fwd(ops.gemm_nt(A, W1)), a.iters), flops=2.0 * M * 2 * h * d)
    rec('nt fc1 fwd + swiglu (GEMM epilogue)', timeit(lambda: ops.fc1_swiglu(A, W1), a.iters), flops=2.0 * M * 2 * h * d)
    del A, W1
    dYm = torch.randn(M, d, device=dev).to(BF)
    W2T = (torch.randn(h, d, device=dev) * 0.02).to(BF)
    Um = torch.randn(M, 2 * h, device=dev).to(BF)
    rec('nt dX fc2 + swiglu bwd (2 launches)', timeit(lambda: ops.swiglu_bwd(ops.gemm_nt(dYm, W2T), Um), a.iters), flops=2.0 * M * h * d)
    rec('nt dX fc2 + swiglu bwd (GEMM epilogue)', timeit(lambda: ops.fc2_dx_swiglu_bwd(dYm, W2T, Um), a.iters), flops=2.0 * M * h * d)
    del dYm, W2T, Um
    for name, (m, n, k) in {'tn dW qkv': (3 * d, d, M), 'tn dW out': (d, d, M), 'tn dW fc1': (2 * h, d, M),
                             'tn dW fc2': (d, h, M), 'tn dW head': (V, d, M)}.items():
      A = torch.randn(k, m, device=dev).to(BF)
      Bm = torch.randn(k, n, device=dev).to(BF)
      out = torch.zeros(m, n, device=dev)
      rec(name, timeit(lambda: ops.gemm_tn(A, Bm, out=out, accumulate=True), a.iters), flops=2.0 * m * n * k)
      if a.variants:
        os.environ['PLM_TN_NO_BIG'] = '1'
        ops.reload_env()
        rec(name + ' [dma128]', timeit(lambda: ops.gemm_tn(A, Bm, out=out, accumulate=True), a.iters), flops=2.0 * m * n * k)
        del os.environ['PLM_TN_NO_BIG']
        ops.reload_env()
      del A, Bm, out

  if want('gemm'):
    shp = [(d, h), (2 * h, d), (d, d), (3 * d, d)]  # fc2, fc1, w_out, w_qkv: the dW GEMMs of one block
    As = [torch.randn(M, m, device=dev).to(BF) for m, _ in shp]
    Bs = [torch.randn(M, n, device=dev).to(BF) for _, n in shp]
    outs = [torch.zeros(m, n, device=dev) for m, n in shp]
    fl = sum(2.0 * m * n * M for m, n in shp)
    rec('tn dW block, 4 launches', timeit(lambda: [ops.gemm_tn(a_, b_, out=o_, accumulate=False) for a_, b_, o_ in zip(As, Bs, outs)], a.iters), flops=fl)
    rec('tn dW block, grouped (1 block = 4 problems)', timeit(lambda: ops.gemm_tn_grouped([(a_, b_, o_, False, None) for a_, b_, o_ in zip(As, Bs, outs)]), a.iters), flops=fl)
    for nb in (3, 6, 12):  # what the engine issues: 3 blocks per launch under DDP, all 12 without a gradient consumer
      outs_n = [torch.zeros(m, n, device=dev) for m, n in shp * nb]
      probs = [(a_, b_, o_, False, None) for a_, b_, o_ in zip(As * nb, Bs * nb, outs_n)]
      rec(f'tn dW, grouped x{nb} blocks (per block)', timeit(lambda: ops.gemm_tn_grouped(probs), a.iters) / nb, flops=fl)
      del outs_n, probs
    del As, Bs, outs

  if want('attn'):
    from plainlm_amd.transformer import rope_tables
    cos, sin = (t.to(dev) for t in rope_tables(64, T))
    qkv = torch.randn(M, 3 * d, device=dev).to(BF)
    dout = torch.randn(M, d, device=dev).to(BF)
    ops.rope_qk_(qkv, cos, sin, B, T, nh)
    out, lse = ops.attn_fwd(qkv, B, T, nh)
    fl = 2.0 * 2 * B * nh * T * (T + 1) / 2 * 64  # causal-counted QK^T + PV
    xq = torch.randn(M, d, device=dev).to(BF)
    wq = (torch.randn(3 * d, d, device=dev) * 0.02).to(BF)
    rec('nt qkv fwd + rope (2 launches)', timeit(lambda: ops.rope_qk_(ops.gemm_nt(xq, wq), cos, sin, B, T, nh), a.iters), flops=2.0 * M * 3 * d * d)
    rec('nt qkv fwd + rope (GEMM epilogue)', timeit(lambda: ops.qkv_rope(xq, wq, cos, sin, B, T, nh), a.iters), flops=2.0 * M * 3 * d * d)
    del xq, wq
    rec('rope qk (in place)', timeit(lambda: ops.rope_qk_(qkv, cos, sin, B, T, nh), a.iters), bytes_=8.0 * M * d)
    rec('attn fwd', timeit(lambda: ops.attn_fwd(qkv, B, T, nh), a.iters), flops=fl)
    rec('attn bwd', timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh), a.iters), flops=2.0 * fl)
    if a.doc_mask:
      import numpy as np
      from plainlm_amd.engine import doc_start_from_lengths
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
      pos = torch.arange(T, device=dev)[None, :]
      fl_m = float((4.0 * nh * 64 * (pos - ds + 1)).sum())  # visible (query, key) pairs x 2 matmuls x 2 flop x head_dim
      rec('attn doc plan (once per batch)', timeit(lambda: ops.attn_doc_plan(ds, nh), a.iters))
      # the shipped kernels, and the same without heavy tiles split into 64-row items (same process, same box)
      for tag, env in (('', {}), (' [no split]', {'PLM_ATTN_DOC_SPLIT_MIN': '0'})):
        for k in ('PLM_ATTN_DOC_SPLIT_MIN',):
          os.environ.pop(k, None)
        os.environ.update(env)
        _lib.load().plm_reload_env()
        plan = ops.attn_doc_plan(ds, nh)
        out_m, lse_m = ops.attn_fwd(qkv, B, T, nh, ds, plan)
        rec('attn fwd (doc masks, mean length 256; flops of the visible pairs)' + tag, timeit(lambda: ops.attn_fwd(qkv, B, T, nh, ds, plan), a.iters), flops=fl_m)
        rec('attn bwd (doc masks)' + tag, timeit(lambda: ops.attn_bwd(qkv, out_m, dout, lse_m, cos, sin, B, T, nh, ds, plan), a.iters), flops=2.0 * fl_m)
      for k in ('PLM_ATTN_DOC_SPLIT_MIN',):
        os.environ.pop(k, None)
      _lib.load().plm_reload_env()

  if want('hbm'):
    x = torch.randn(M, d, device=dev)
    w = torch.ones(d, device=dev)
    br = torch.randn(M, d, device=dev).to(BF)
    rec('rmsnorm fwd', timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6), a.iters), bytes_=6.0 * M * d)
    rec('add+rmsnorm fwd', timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, branch=br), a.iters), bytes_=12.0 * M * d)
    _, y, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
    rec('rmsnorm bwd', timeit(lambda: ops.rmsnorm_bwd(y, x, w, rstd, gin=x, want_bf16=True), a.iters), bytes_=16.0 * M * d)
    u = torch.randn(M, 2 * h, device=dev).to(BF)
    g = torch.randn(M, h, device=dev).to(BF)
    rec('swiglu fwd', timeit(lambda: ops.swiglu_fwd(u), a.iters), bytes_=6.0 * M * h)
    rec('swiglu bwd', timeit(lambda: ops.swiglu_bwd(g, u), a.iters), bytes_=10.0 * M * h)
    del u, g
    logits = torch.randn(M, V, device=dev).to(BF)
    tg = torch.randint(0, V, (M,), device=dev)
    rec('ce fwd+bwd', timeit(lambda: ops.ce_fwd_bwd_(logits, tg, 1.0 / M), max(3, a.iters // 4)), bytes_=4.0 * M * V)
    del logits
    ids = torch.randint(0, V, (M,), device=dev)
    W = torch.randn(V, d, device=dev)
    rec('embed fwd', timeit(lambda: ops.embed_fwd(ids, W), a.iters), bytes_=8.0 * M * d + 8.0 * M)
    dW = torch.zeros(V, d, device=dev)
    rec('embed bwd (atomic)', timeit(lambda: ops.embed_bwd(ids, x, dW), a.iters), bytes_=12.0 * M * d)
    rec('embed bwd (sorted, writes all of dW)', timeit(lambda: ops.embed_bwd_sorted(ids, x, dW, False), a.iters), bytes_=4.0 * M * d + 4.0 * V * d)
    P = torch.randn(2304, 768, device=dev)
    rec('cast+transpose qkv', timeit(lambda: ops.cast_bf16_t(P), a.iters), bytes_=8.0 * P.numel())
    flat = torch.randn(n_params, device=dev)
    rec(f'sumsq {n_params / 1e6:.0f}M', timeit(lambda: ops.sumsq(flat), a.iters), bytes_=4.0 * flat.numel())
    m_, v_, g_ = torch.zeros_like(flat), torch.zeros_like(flat), torch.randn_like(flat)
    rec(f'adamw {n_params / 1e6:.0f}M', timeit(lambda: ops.adamw_(flat, g_, m_, v_, 1e-3, 0.9, 0.95, 1e-8, 0.1, 1), a.iters), bytes_=28.0 * flat.numel())

  if a.json:
    with open(a.json, 'w') as f:
      json.dump(rows, f, indent=1)


if __name__ == '__main__':
  main()

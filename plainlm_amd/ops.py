"""Tensor-level wrappers over the C ABI: validate, allocate outputs with torch (the caching
allocator owns all memory), pass raw device pointers + the current HIP stream.

Every function here launches hand-written gfx950 kernels; none has a torch fallback.
"""

import ctypes as C
import os

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32


def _stream():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-launch timing used by bench.py's roofline leg: when PROFILE is a list, every MFMA kernel launch - and every launch of the
# HBM-bound kernels, whose family names start with 'hbm:' and whose work figure is ALGORITHMIC BYTES instead of
# flops - is bracketed by HIP events on the launch stream and (family, work, start, end) appended.
PROFILE = None
# Called with (family, flops) right before every MFMA kernel launch when set (ddp.GradReducer: it keeps an estimate of the GPU time
# enqueued so far and reserves CUs for RCCL only while a gradient bucket's collective is expected to be running).
LAUNCH_HOOK = None


class _Timed:
  __slots__ = ('family', 'flops', 'start')

  def __init__(self, family, flops):
    self.family, self.flops, self.start = family, flops, None

  def __enter__(self):
    if PROFILE is not None:
      self.start = torch.cuda.Event(enable_timing=True)
      self.start.record()
    return self

  def __exit__(self, *exc):
    if self.start is not None:
      end = torch.cuda.Event(enable_timing=True)
      end.record()
      PROFILE.append((self.family, self.flops, self.start, end))
    return False


def _hook(family, flops):
  """First thing in every MFMA op, BEFORE its workspace query: the hook may change the CU reserve, and plans (tile shape, stream-K /
  split-K workspace) are functions of it."""
  if LAUNCH_HOOK is not None:
    LAUNCH_HOOK(family, flops)


def _p(t):
  return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need(t, dtype, name, dim=None):
  if not t.is_cuda:
    raise RuntimeError(f'{name}: tensor must live on the GPU (plainlm_amd has no CPU path)')
  if t.dtype != dtype:
    raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
  if not t.is_contiguous():
    raise ValueError(f'{name}: tensor must be contiguous')
  if dim is not None and t.dim() != dim:
    raise ValueError(f'{name}: expected {dim} dims, got {t.dim()}')
  return t


# ---- casts ------------------------------------------------------------------
def cast_bf16(src, out=None):
  _need(src, F32, 'cast_bf16.src')
  out = torch.empty(src.shape, dtype=BF16, device=src.device) if out is None else out
  _lib.check(_lib.load().plm_cast_f32_bf16(_p(src), _p(out), src.numel(), _stream()), 'plm_cast_f32_bf16')
  return out


def cast_bf16_t(src, out=None, out_t=None):
  """src fp32 [R, C] -> (bf16 [R, C], bf16 [C, R]).  ``out_t`` may be a zero-initialised [C, R_pad] buffer
  (R_pad >= R): only its first R columns are written, the pad stays zero (K-padding for the dX GEMM)."""
  _need(src, F32, 'cast_bf16_t.src', 2)
  R, Cc = src.shape
  out = torch.empty((R, Cc), dtype=BF16, device=src.device) if out is None else out
  out_t = torch.empty((Cc, R), dtype=BF16, device=src.device) if out_t is None else out_t
  if out_t.dtype != BF16 or out_t.dim() != 2 or out_t.shape[0] != Cc or out_t.shape[1] < R or out_t.stride(1) != 1:
    raise ValueError('cast_bf16_t.out_t: need bf16 [C, >=R]')
  _lib.check(_lib.load().plm_cast_f32_bf16_t(_p(src), _p(out), _p(out_t), R, Cc, out_t.stride(0), _stream()),
             'plm_cast_f32_bf16_t')
  return out, out_t


def cast_bf16_t_multi(items):
  """[(src fp32 [R, C], out bf16 [R, C], out_t bf16 [C, >= R])]: cast_bf16_t for all of them in one launch."""
  if not items:
    return
  arr = (_lib.CastItem * len(items))()
  for i, (src, out, out_t) in enumerate(items):
    _need(src, F32, 'cast_bf16_t_multi.src', 2)
    R, Cc = src.shape
    if out.dtype != BF16 or out.shape != (R, Cc) or not out.is_contiguous() or not out.is_cuda:
      raise ValueError('cast_bf16_t_multi.out: need contiguous bf16 [R, C] on the GPU')
    if out_t.dtype != BF16 or out_t.dim() != 2 or out_t.shape[0] != Cc or out_t.shape[1] < R or out_t.stride(1) != 1 or not out_t.is_cuda:
      raise ValueError('cast_bf16_t_multi.out_t: need bf16 [C, >=R] on the GPU')
    arr[i] = _lib.CastItem(src.data_ptr(), out.data_ptr(), out_t.data_ptr(), R, Cc, out_t.stride(0))
  with _Timed('hbm:cast_bf16_t_multi', 8.0 * sum(src.numel() for src, _, _ in items)):  # one fp32 read, the bf16 copy and its transpose written
    _lib.check(_lib.load().plm_cast_f32_bf16_t_multi(arr, len(items), _stream()), 'plm_cast_f32_bf16_t_multi')


# ---- embedding ----------------------------------------------------------------
def embed_fwd(ids, W):
  _need(ids, torch.int64, 'embed_fwd.ids')
  _need(W, F32, 'embed_fwd.W', 2)
  M = ids.numel()
  out = torch.empty((M, W.shape[1]), dtype=F32, device=W.device)
  with _Timed('hbm:embed_fwd', 8.0 * M * W.shape[1] + 8.0 * M):
    _lib.check(_lib.load().plm_embed_fwd(_p(ids), _p(W), _p(out), M, W.shape[1], W.shape[0], _stream()), 'plm_embed_fwd')
  return out


def embed_bwd(ids, dout, dW):
  """dW[ids[m]] += dout[m]  (dW must hold the value to accumulate onto)."""
  _need(ids, torch.int64, 'embed_bwd.ids')
  _need(dout, F32, 'embed_bwd.dout', 2)
  _need(dW, F32, 'embed_bwd.dW', 2)
  _lib.check(_lib.load().plm_embed_bwd(_p(ids), _p(dout), _p(dW), ids.numel(), dW.shape[1], dW.shape[0], _stream()),
             'plm_embed_bwd')
  return dW


_embed_ws = {}


def embed_bwd_sorted(ids, dout, dW, accumulate):
  """dW[v] (+)= sum of dout rows of the tokens with id v, in token order, without atomics (bit-reproducible); with
  accumulate=False every row of dW is written (zeros for unused ids).  Returns False when the shape is not supported
  (V >= 65536) - the caller then falls back to embed_bwd.  The launch sorts at most 65536 tokens (16-bit positions in its keys); longer
  batches go through it in slices of that many tokens, every slice after the first accumulating onto the previous ones - still in token
  order, still without atomics."""
  ids = ids.reshape(-1)
  _need(ids, torch.int64, 'embed_bwd_sorted.ids')
  _need(dout, F32, 'embed_bwd_sorted.dout', 2)
  _need(dW, F32, 'embed_bwd_sorted.dW', 2)
  lib = _lib.load()
  M, lim = ids.numel(), 65536
  if M > lim:
    if lib.plm_embed_bwd_workspace_bytes(lim, dW.shape[0]) == 0:
      return False
    for lo in range(0, M, lim):
      embed_bwd_sorted(ids[lo:lo + lim], dout[lo:lo + lim], dW, accumulate or lo > 0)
    return True
  nbytes = lib.plm_embed_bwd_workspace_bytes(ids.numel(), dW.shape[0])
  if nbytes == 0:
    return False
  ws = _embed_ws.get(dout.device)
  if ws is None or ws.numel() < nbytes:
    ws = _embed_ws[dout.device] = torch.empty(nbytes, dtype=torch.uint8, device=dout.device)
  with _Timed('hbm:embed_bwd', 4.0 * dout.numel() + (8.0 if accumulate else 4.0) * dW.numel() + 8.0 * ids.numel()):
    _lib.check(lib.plm_embed_bwd_sorted(_p(ids), _p(dout), _p(dW), ids.numel(), dW.shape[1], dW.shape[0], int(bool(accumulate)),
                                        _p(ws), nbytes, _stream()), 'plm_embed_bwd_sorted')
  return True


# ---- rmsnorm --------------------------------------------------------------------
def rmsnorm_fwd(x, w, eps, branch=None, write_xout=False):
  """returns (xout fp32 or None, y bf16, rstd fp32).  r = x + branch; y = bf16(r * rstd * w)."""
  _need(x, F32, 'rmsnorm_fwd.x', 2)
  _need(w, F32, 'rmsnorm_fwd.w', 1)
  M, d = x.shape
  if branch is not None:
    _need(branch, BF16, 'rmsnorm_fwd.branch', 2)
  xout = torch.empty_like(x) if (write_xout or branch is not None) else None
  y = torch.empty((M, d), dtype=BF16, device=x.device)
  rstd = torch.empty((M,), dtype=F32, device=x.device)
  with _Timed('hbm:rmsnorm_fwd', (6.0 + (2.0 if branch is not None else 0.0) + (4.0 if xout is not None else 0.0)) * M * d):
    _lib.check(_lib.load().plm_rmsnorm_fwd(_p(x), _p(branch), _p(xout), _p(w), _p(y), _p(rstd), M, d, float(eps), _stream()),
               'plm_rmsnorm_fwd')
  return xout, y, rstd


def rmsnorm_bwd(dy, x, w, rstd, gin=None, want_bf16=False, dw_out=None, dw_accumulate=False, defer_dw=False):
  """returns (dx fp32, dx_bf16 or None, dw fp32[d]).  dx = gin + d(rmsnorm).
  defer_dw: skip the column sum and return the [nblk, d] per-block partials in place of dw (the caller reduces a whole
  backward pass's norms with ONE colsum_multi launch)."""
  _need(dy, BF16, 'rmsnorm_bwd.dy', 2)
  _need(x, F32, 'rmsnorm_bwd.x', 2)
  M, d = x.shape
  lib = _lib.load()
  if gin is not None:
    _need(gin, F32, 'rmsnorm_bwd.gin', 2)
  dx = torch.empty_like(x)
  dxb = torch.empty((M, d), dtype=BF16, device=x.device) if want_bf16 else None
  nblk = lib.plm_rmsnorm_bwd_blocks(M)
  part = torch.empty((nblk, d), dtype=F32, device=x.device)
  with _Timed('hbm:rmsnorm_bwd', (10.0 + (4.0 if gin is not None else 0.0) + (2.0 if want_bf16 else 0.0)) * M * d):
    _lib.check(lib.plm_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(gin), _p(dx), _p(dxb), _p(part), M, d, _stream()),
               'plm_rmsnorm_bwd')
  if defer_dw:
    return dx, dxb, part
  if dw_out is None:
    dw_out = torch.empty((d,), dtype=F32, device=x.device)
    dw_accumulate = False
  _lib.check(lib.plm_colsum_f32(_p(part), _p(dw_out), nblk, d, int(bool(dw_accumulate)), _stream()), 'plm_colsum_f32')
  return dx, dxb, dw_out


def colsum_multi(items):
  """[(part fp32 [rows, d], out fp32 [d], accumulate)] with one common rows x d: out (+)= column sums, ONE launch."""
  if not items:
    return
  rows, d = items[0][0].shape
  arr = (_lib.ColsumItem * len(items))()
  for i, (part, out, acc) in enumerate(items):
    _need(part, F32, 'colsum_multi.part', 2)
    _need(out, F32, 'colsum_multi.out')
    if tuple(part.shape) != (rows, d) or out.numel() != d:
      raise ValueError('colsum_multi: every item needs the same [rows, d] partials and a [d] output')
    arr[i] = _lib.ColsumItem(part.data_ptr(), out.data_ptr(), int(bool(acc)))
  _lib.check(_lib.load().plm_colsum_f32_multi(arr, len(items), rows, d, _stream()), 'plm_colsum_f32_multi')


# ---- swiglu -----------------------------------------------------------------------
def swiglu_fwd(u):
  _need(u, BF16, 'swiglu_fwd.u', 2)
  M, h2 = u.shape
  out = torch.empty((M, h2 // 2), dtype=BF16, device=u.device)
  _lib.check(_lib.load().plm_swiglu_fwd(_p(u), _p(out), M, h2 // 2, _stream()), 'plm_swiglu_fwd')
  return out


def swiglu_bwd(dout, u):
  _need(dout, BF16, 'swiglu_bwd.dout', 2)
  _need(u, BF16, 'swiglu_bwd.u', 2)
  du = torch.empty_like(u)
  _lib.check(_lib.load().plm_swiglu_bwd(_p(dout), _p(u), _p(du), u.shape[0], u.shape[1] // 2, _stream()), 'plm_swiglu_bwd')
  return du


ACT_KINDS = {'silu': 0, 'relu_sq': 1}


def act_fwd(u, kind):
  """bf16 act(u) for the plain MLP classes (components.py:31-40 silu, :59-70 relu squared)."""
  _need(u, BF16, 'act_fwd.u', 2)
  out = torch.empty_like(u)
  _lib.check(_lib.load().plm_act_fwd(_p(u), _p(out), u.numel(), ACT_KINDS[kind], _stream()), 'plm_act_fwd')
  return out


def act_bwd(dout, u, kind):
  _need(dout, BF16, 'act_bwd.dout', 2)
  _need(u, BF16, 'act_bwd.u', 2)
  du = torch.empty_like(u)
  _lib.check(_lib.load().plm_act_bwd(_p(dout), _p(u), _p(du), u.numel(), ACT_KINDS[kind], _stream()), 'plm_act_bwd')
  return du


# ---- GEMMs ------------------------------------------------------------------------
_tn_ws = {}
_nt_ws_cache = {}


_cu_reserve = 0


def set_cu_reserve(n):
  """Leave n CUs out of the persistent GEMM grids (RCCL's kernels run there while gradient buckets are in flight); every
  plan (tile shape, hybrid stream-K, split-K) is recomputed per launch from the current value."""
  global _cu_reserve
  _lib.check(_lib.load().plm_set_cu_reserve(int(n)), 'plm_set_cu_reserve')
  _cu_reserve = int(n)


def cu_reserve():
  return _cu_reserve


_env_epoch = 0


def reload_env():
  """Have the library read its PLM_* environment switches again (it reads them once, at its first call; tests and A/B tools
  that change os.environ afterwards call this)."""
  global _env_epoch
  _lib.load().plm_reload_env()
  _env_epoch += 1


def _nt_ws_bytes(lib, M, N, K):
  """Workspace query of the hybrid NT schedule, cached per shape (the plan depends on the shape, the library's environment
  switches - see reload_env - and the CU reserve: all part of the key)."""
  key = (M, N, K, _cu_reserve, _env_epoch)
  v = _nt_ws_cache.get(key)
  if v is None:
    v = _nt_ws_cache[key] = int(lib.plm_gemm_nt_workspace_bytes(M, N, K))
  return v


def _tn_workspace(nbytes, device):
  """One grow-only split-K slab buffer per device (owned by torch's allocator)."""
  buf = _tn_ws.get(device)
  if buf is None or buf.numel() < nbytes:
    buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    _tn_ws[device] = buf
  return buf


def gemm_nt(A, B, out=None, out_dtype=BF16, accumulate=False, alpha=None, variant=0):
  """C[M,N] = alpha * A[M,K] @ B[N,K]^T ; A, B bf16 (row stride may exceed K).
  variant: 0 auto | 1 128x128 register-staged | 2 128x128 LDS-DMA | 3 persistent 256x256, plain 4-phase ring |
  4 / 5 / 6 / 7 persistent 256x256 / 256x192 / 256x128 / 128x192, deep-prefetch ring with offset wave groups (what auto picks from)."""
  for t, n in ((A, 'A'), (B, 'B')):
    if not t.is_cuda or t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1:
      raise ValueError(f'gemm_nt.{n}: need a 2-D bf16 GPU tensor with unit inner stride')
  M, K = A.shape
  N = B.shape[0]
  if B.shape[1] != K:
    raise ValueError(f'gemm_nt: inner dims differ ({K} vs {B.shape[1]})')
  if out is None:
    out = torch.empty((M, N), dtype=out_dtype, device=A.device)
  if out.dim() != 2 or out.shape != (M, N) or out.stride(1) != 1:
    raise ValueError('gemm_nt.out: bad shape/stride')
  cd = {BF16: 0, F32: 1}[out.dtype]
  if alpha is not None:
    _need(alpha, F32, 'gemm_nt.alpha')
  lib = _lib.load()
  _hook('gemm_nt', 2.0 * M * N * K)
  nbytes = _nt_ws_bytes(lib, M, N, K) if (variant == 0 and cd == 0) else 0
  ws = _tn_workspace(nbytes, A.device) if nbytes else None  # the split-K slab buffer is shared with gemm_tn (same stream)
  with _Timed('gemm_nt', 2.0 * M * N * K):
    _lib.check(lib.plm_gemm_bf16_nt_ws(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), M, N, K, cd,
                                       int(bool(accumulate)), _p(alpha), int(variant), _p(ws), nbytes, _stream()), 'plm_gemm_bf16_nt')
  return out


def gemm_tn(A, B, out=None, accumulate=False, alpha=None):
  """C[M,N] (+)= alpha * A[K,M]^T @ B[K,N] ; A, B bf16, C fp32."""
  for t, n in ((A, 'A'), (B, 'B')):
    if not t.is_cuda or t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1:
      raise ValueError(f'gemm_tn.{n}: need a 2-D bf16 GPU tensor with unit inner stride')
  K, M = A.shape
  N = B.shape[1]
  if B.shape[0] != K:
    raise ValueError(f'gemm_tn: contraction dims differ ({K} vs {B.shape[0]})')
  if out is None:
    out = torch.empty((M, N), dtype=F32, device=A.device)
    accumulate = False
  if out.dtype != F32 or out.shape != (M, N) or out.stride(1) != 1:
    raise ValueError('gemm_tn.out: need fp32 [M, N] with unit inner stride')
  lib = _lib.load()
  _hook('gemm_tn', 2.0 * M * N * K)
  nbytes = lib.plm_gemm_tn_workspace_bytes(M, N, K)
  ws = _tn_workspace(nbytes, A.device) if nbytes else None
  if alpha is not None:
    _need(alpha, F32, 'gemm_tn.alpha')
  with _Timed('gemm_tn', 2.0 * M * N * K):
    _lib.check(lib.plm_gemm_bf16_tn(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), M, N, K,
                                    int(bool(accumulate)), _p(alpha), _p(ws), nbytes, _stream()), 'plm_gemm_bf16_tn')
  return out


TN_GROUP_MAX = 48  # PLM_TN_GROUP_MAX of csrc/gemm_big.hip


def gemm_tn_grouped(problems):
  """Several dW-style GEMMs with the same contraction length as ONE launch over the union of their output tiles (whole-K tiles for
  the full rounds of the persistent grid, split-K + one reduce for the remainder):
  problems = [(A[K,M] bf16, B[K,N] bf16, out[M,N] fp32, accumulate, alpha or None), ...] (at most TN_GROUP_MAX = 48);
  out (+)= alpha * A^T @ B for each.  Returns False when the shapes cannot be grouped (the caller then uses gemm_tn)."""
  n = len(problems)
  if n < 1 or n > TN_GROUP_MAX:
    return False
  K = problems[0][0].shape[0]
  arr = (_lib.TnProblem * n)()
  Ms, Ns = (C.c_int64 * n)(), (C.c_int64 * n)()
  for i, (A, B, out, accumulate, alpha) in enumerate(problems):
    for t, nm in ((A, 'A'), (B, 'B')):
      if not t.is_cuda or t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f'gemm_tn_grouped.{nm}: need a 2-D bf16 GPU tensor with unit inner stride')
    if A.shape[0] != K or B.shape[0] != K:
      return False
    M, N = A.shape[1], B.shape[1]
    if out.dtype != F32 or out.shape != (M, N) or out.stride(1) != 1:
      raise ValueError('gemm_tn_grouped.out: need fp32 [M, N] with unit inner stride')
    if alpha is not None:
      _need(alpha, F32, 'gemm_tn_grouped.alpha')
    Ms[i], Ns[i] = M, N
    arr[i] = _lib.TnProblem(_p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), M, N, int(bool(accumulate)), _p(alpha))
  lib = _lib.load()
  if lib.plm_gemm_tn_grouped_workspace_bytes(Ms, Ns, n, K) == 0:  # unsupported shapes (independent of the CU reserve): the caller's
    return False                                                  # per-problem gemm_tn calls report to the launch hook themselves
  flops = sum(2.0 * a.shape[1] * b.shape[1] * K for a, b, *_ in problems)
  _hook('gemm_tn', flops)  # may change the CU reserve: the plan is taken after it
  nbytes = lib.plm_gemm_tn_grouped_workspace_bytes(Ms, Ns, n, K)
  ws = _tn_workspace(nbytes, problems[0][0].device)
  with _Timed('gemm_tn', flops):
    _lib.check(lib.plm_gemm_bf16_tn_grouped(arr, n, K, _p(ws), nbytes, _stream()), 'plm_gemm_bf16_tn_grouped')
  return True


def fc1_swiglu(x, w_fc1):
  """fc1 of the SwiGLU MLP with the activation in the GEMM epilogue: returns (u [M, 2h] = x @ w_fc1^T as gate | up, act [M, h])."""
  _need(x, BF16, 'fc1_swiglu.x', 2)
  _need(w_fc1, BF16, 'fc1_swiglu.w', 2)
  M, K = x.shape
  N = w_fc1.shape[0]
  if w_fc1.shape[1] != K or N % 16 != 0 or x.stride(1) != 1 or w_fc1.stride(1) != 1:
    raise ValueError('fc1_swiglu: need x [M, K], w [2h, K] with unit inner strides and h % 8 == 0')
  h = N // 2
  u = torch.empty((M, N), dtype=BF16, device=x.device)
  act = torch.empty((M, h), dtype=BF16, device=x.device)
  _hook('gemm_nt_fused', 2.0 * M * N * K)
  with _Timed('gemm_nt_fused', 2.0 * M * N * K):
    _lib.check(_lib.load().plm_fc1_swiglu_bf16(_p(x), x.stride(0), _p(w_fc1), w_fc1.stride(0), _p(u), _p(act), M, h, K, _stream()),
               'plm_fc1_swiglu_bf16')
  return u, act


def fc2_dx_swiglu_bwd(dy, w2t, u):
  """du [M, 2h] = swiglu_bwd(dy @ w2t^T, u): the dX GEMM of fc2 (w2t = bf16 W_fc2^T [h, K(+pad)]) with the SwiGLU backward in its
  epilogue; d(act) is never stored for qualifying shapes."""
  _need(dy, BF16, 'fc2_dx_swiglu_bwd.dy', 2)
  _need(w2t, BF16, 'fc2_dx_swiglu_bwd.w2t', 2)
  _need(u, BF16, 'fc2_dx_swiglu_bwd.u', 2)
  M, K = dy.shape
  h = w2t.shape[0]
  if w2t.shape[1] != K or u.shape != (M, 2 * h) or not u.is_contiguous() or dy.stride(1) != 1 or w2t.stride(1) != 1:
    raise ValueError('fc2_dx_swiglu_bwd: need dy [M, K], w2t [h, K], contiguous u [M, 2h]')
  du = torch.empty((M, 2 * h), dtype=BF16, device=dy.device)
  lib = _lib.load()
  _hook('gemm_nt_fused', 2.0 * M * h * K)
  with _Timed('gemm_nt_fused', 2.0 * M * h * K):
    # first without the d(act) scratch: the one-launch path never touches it; the library answers PLM_E_WORKSPACE (-4) when this
    # shape takes the two-launch path, and only then is the buffer allocated
    rc = lib.plm_fc2_dx_swiglu_bwd_bf16(_p(dy), dy.stride(0), _p(w2t), w2t.stride(0), _p(u), _p(du), _p(None), M, h, K, _stream())
    if rc == -4:
      scratch = torch.empty((M, h), dtype=BF16, device=dy.device)
      rc = lib.plm_fc2_dx_swiglu_bwd_bf16(_p(dy), dy.stride(0), _p(w2t), w2t.stride(0), _p(u), _p(du), _p(scratch), M, h, K, _stream())
    _lib.check(rc, 'plm_fc2_dx_swiglu_bwd_bf16')
  return du


# ---- attention ----------------------------------------------------------------------
def rope_qk_(qkv, rope_cos, rope_sin, B, T, nh):
  """Rotate the q and k column blocks of a projection output in place (qkv_rope's fallback; the step uses the fused epilogue)."""
  _need(qkv, BF16, 'rope_qk.qkv', 2)
  hd = qkv.shape[1] // (3 * nh)
  _need(rope_cos, F32, 'rope_qk.rope_cos', 2)
  _need(rope_sin, F32, 'rope_qk.rope_sin', 2)
  if rope_cos.shape[0] < T or rope_cos.shape[1] != hd // 2:
    raise ValueError('rope_qk: RoPE table too short for T or wrong head_dim')
  _lib.check(_lib.load().plm_rope_qk(_p(qkv), _p(rope_cos), _p(rope_sin), B, T, nh, hd, _stream()), 'plm_rope_qk')
  return qkv


def qkv_rope(x, w_qkv, rope_cos, rope_sin, B, T, nh):
  """qkv[M, 3d] = x @ w_qkv^T with q | k rotated: one launch, the rotation in the GEMM's store-side epilogue (shapes that do
  not qualify: GEMM + the in-place rope_qk_ pass, same bits)."""
  _need(x, BF16, 'qkv_rope.x', 2)
  _need(w_qkv, BF16, 'qkv_rope.w', 2)
  M, K = x.shape
  N = w_qkv.shape[0]
  hd = N // (3 * nh)
  out = torch.empty((M, N), dtype=BF16, device=x.device)
  _hook('gemm_nt_fused', 2.0 * M * N * K)
  with _Timed('gemm_nt_fused', 2.0 * M * N * K):
    _lib.check(_lib.load().plm_qkv_rope_bf16(_p(x), x.stride(0), _p(w_qkv), w_qkv.stride(0), _p(out), N, M, K, _p(rope_cos),
                                             _p(rope_sin), B, T, nh, hd, _stream()), 'plm_qkv_rope_bf16')
  return out


def doc_start_from_mask(mask):
  """bool [B, T, T] (True = may attend: the reference's mask, engine/engine.py:21-23) -> (doc_start int32 [B, T], status int32 [1]): status
  becomes 1 when some row is not exactly True on [doc_start, i] - the caller decides when to look (it is a device tensor)."""
  if mask.dtype != torch.bool or mask.dim() != 3 or mask.shape[1] != mask.shape[2] or not mask.is_cuda:
    raise ValueError('doc_start_from_mask: need a bool [B, T, T] tensor on the GPU')
  mask = mask.contiguous()
  B, T = mask.shape[0], mask.shape[1]
  ds = torch.empty((B, T), dtype=torch.int32, device=mask.device)
  status = torch.zeros((1,), dtype=torch.int32, device=mask.device)
  _lib.check(_lib.load().plm_attn_doc_start_from_mask(_p(mask), _p(ds), _p(status), B, T, _stream()), 'plm_attn_doc_start_from_mask')
  return ds, status


def attn_doc_plan(doc_start, nh):
  """Plan of a document-masked batch for attention with nh heads (include/plainlm_hip.h: doc_end[B,T] + the sorted item lists): built ONCE
  per batch, shared by every layer's forward / backward launches."""
  _need(doc_start, torch.int32, 'attn_doc_plan.doc_start', 2)
  B, T = doc_start.shape
  lib = _lib.load()
  plan = torch.empty((lib.plm_attn_doc_plan_bytes(B, T) // 4,), dtype=torch.int32, device=doc_start.device)
  _lib.check(lib.plm_attn_doc_plan(_p(doc_start), _p(plan), B, T, nh, _stream()), 'plm_attn_doc_plan')
  return plan


def attn_flops(B, T, nh, hd, doc_start=None):
  """Algorithmic flops of the forward pass: 2 matmuls x 2 flop x hd per visible (query, key) pair - causal T(T+1)/2 pairs per head and
  sequence, with a document mask the pairs doc_start[i] <= j <= i (one small device reduction, cached on the tensor)."""
  if doc_start is None:
    return 4.0 * B * nh * hd * T * (T + 1) / 2
  pairs = getattr(doc_start, '_plm_pairs', None)
  if pairs is None:
    pos = torch.arange(T, device=doc_start.device, dtype=torch.int64)[None, :]
    pairs = float((pos - doc_start.to(torch.int64) + 1).sum().item())
    try:
      doc_start._plm_pairs = pairs
    except AttributeError:
      pass
  return 4.0 * nh * hd * pairs


def attn_fwd(qkv_rot, B, T, nh, doc_start=None, plan=None):
  """qkv_rot: projection output with q, k already rotated (rope_qk_).  doc_start: int32 [B,T] document mask; plan: attn_doc_plan(doc_start)
  (built here when absent - pass it when the same batch goes through several layers)."""
  _need(qkv_rot, BF16, 'attn_fwd.qkv', 2)
  hd = qkv_rot.shape[1] // (3 * nh)
  if doc_start is not None:
    _need(doc_start, torch.int32, 'attn_fwd.doc_start', 2)
    if plan is None:
      plan = attn_doc_plan(doc_start, nh)
  out = torch.empty((B * T, nh * hd), dtype=BF16, device=qkv_rot.device)
  lse = torch.empty((B, nh, T), dtype=F32, device=qkv_rot.device)
  _hook('attn_fwd', attn_flops(B, T, nh, hd))  # the reducer's clock only needs an estimate: no device reduction on the launch path
  with _Timed('attn_fwd', attn_flops(B, T, nh, hd, doc_start) if PROFILE is not None else 0.0):
    _lib.check(_lib.load().plm_attn_fwd(_p(qkv_rot), _p(doc_start), _p(plan), _p(out), _p(lse), B, T, nh, hd, _stream()), 'plm_attn_fwd')
  return out, lse


def attn_bwd(qkv, out, dout, lse, rope_cos, rope_sin, B, T, nh, doc_start=None, plan=None):
  _need(dout, BF16, 'attn_bwd.dout', 2)
  hd = qkv.shape[1] // (3 * nh)
  if doc_start is not None and plan is None:
    plan = attn_doc_plan(doc_start, nh)
  dqkv = torch.empty_like(qkv)
  delta = torch.empty((B, nh, T), dtype=F32, device=qkv.device)
  _hook('attn_bwd', 2.0 * attn_flops(B, T, nh, hd))
  with _Timed('attn_bwd', 2.0 * attn_flops(B, T, nh, hd, doc_start) if PROFILE is not None else 0.0):
    _lib.check(_lib.load().plm_attn_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(rope_cos), _p(rope_sin), _p(doc_start), _p(plan),
                                        _p(dqkv), _p(delta), B, T, nh, hd, _stream()), 'plm_attn_bwd')
  return dqkv


# ---- cross entropy --------------------------------------------------------------------
def ce_fwd_bwd_(logits, targets, grad_scale, V=None):
  """In place: logits[:, :V] <- (softmax - onehot) * grad_scale, logits[:, V:] <- 0 (logits is bf16 [M, ld], ld >= V).
  Returns per-row losses fp32[M]."""
  _need(logits, BF16, 'ce.logits', 2)
  _need(targets, torch.int64, 'ce.targets', 1)
  M, ld = logits.shape
  V = ld if V is None else V
  rows = torch.empty((M,), dtype=F32, device=logits.device)
  with _Timed('hbm:ce_fwd_bwd', 4.0 * M * V):  # one read + one write of the bf16 logits
    _lib.check(_lib.load().plm_ce_fwd_bwd(_p(logits), _p(targets), _p(rows), M, V, ld, float(grad_scale), _stream()),
               'plm_ce_fwd_bwd')
  return rows


def mean(x):
  _need(x, F32, 'mean.x')
  out = torch.empty((), dtype=F32, device=x.device)
  _lib.check(_lib.load().plm_mean_f32(_p(x), _p(out), x.numel(), _stream()), 'plm_mean_f32')
  return out


def scale_bf16_(x, alpha):
  """x <- bf16(x * alpha) in place; alpha is a 0-d fp32 device tensor."""
  _need(x, BF16, 'scale_bf16.x')
  _need(alpha, F32, 'scale_bf16.alpha')
  _lib.check(_lib.load().plm_scale_bf16(_p(x), x.numel(), _p(alpha), _stream()), 'plm_scale_bf16')
  return x


def axpy_f32_(out, x, alpha=None, accumulate=True):
  """out <- (accumulate ? out : 0) + alpha * x; alpha a 0-d fp32 device tensor (None = 1)."""
  _need(out, F32, 'axpy.out')
  _need(x, F32, 'axpy.x')
  if out.numel() != x.numel():
    raise ValueError('axpy_f32_: sizes differ')
  if alpha is not None:
    _need(alpha, F32, 'axpy.alpha')
  _lib.check(_lib.load().plm_axpy_f32(_p(out), _p(x), x.numel(), _p(alpha), int(bool(accumulate)), _stream()), 'plm_axpy_f32')
  return out


# ---- optimizer tail ---------------------------------------------------------------------
def sumsq(x, scratch=None):
  _need(x, F32, 'sumsq.x')
  scratch = torch.empty(4096, dtype=F32, device=x.device) if scratch is None else scratch
  out = torch.empty((), dtype=F32, device=x.device)
  _lib.check(_lib.load().plm_sumsq_f32(_p(x), x.numel(), _p(scratch), _p(out), _stream()), 'plm_sumsq_f32')
  return out


def adamw_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, clip_coef=None):
  for t, n in ((p, 'p'), (g, 'g'), (m, 'm'), (v, 'v')):
    _need(t, F32, 'adamw.' + n)
  bc1 = 1.0 - beta1 ** step
  bc2 = 1.0 - beta2 ** step
  _lib.check(_lib.load().plm_adamw_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                                       _p(clip_coef), _stream()), 'plm_adamw_f32')


def adamw_cast_multi_(items, lr, beta1, beta2, eps, weight_decay, step, clip_coef=None, table=None):
  """AdamW on a list of Linear weights that also writes their bf16 shadows: items = [(p, g, m, v fp32 [rows, cols], dst bf16 [rows, cols],
  dst_t bf16 [cols, >= rows])].  Returns the ctypes item table; pass it back as `table` on later steps (the pointers do not move)."""
  if table is None:
    table = (_lib.AdamwItem * len(items))()
    for i, (p, g, m, v, dst, dst_t) in enumerate(items):
      for t, n in ((p, 'p'), (g, 'g'), (m, 'm'), (v, 'v')):
        _need(t, F32, 'adamw_cast_multi.' + n, 2)
      R, Cc = p.shape
      if dst.dtype != BF16 or tuple(dst.shape) != (R, Cc) or not dst.is_contiguous() or not dst.is_cuda:
        raise ValueError('adamw_cast_multi.dst: need contiguous bf16 [rows, cols] on the GPU')
      if dst_t.dtype != BF16 or dst_t.dim() != 2 or dst_t.shape[0] != Cc or dst_t.shape[1] < R or dst_t.stride(1) != 1 or not dst_t.is_cuda:
        raise ValueError('adamw_cast_multi.dst_t: need bf16 [cols, >= rows] on the GPU')
      table[i] = _lib.AdamwItem(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), dst.data_ptr(), dst_t.data_ptr(), R, Cc, dst_t.stride(0))
  bc1 = 1.0 - beta1 ** step
  bc2 = 1.0 - beta2 ** step
  _lib.check(_lib.load().plm_adamw_cast_multi(table, len(table), lr, beta1, beta2, eps, weight_decay, bc1, bc2, _p(clip_coef), _stream()),
             'plm_adamw_cast_multi')
  return table


# ---- probes -------------------------------------------------------------------------------
def probe_ds_read_tr16():
  out = torch.empty(256, dtype=torch.int32, device='cuda')
  _lib.check(_lib.load().plm_probe_ds_read_tr16(_p(out), _stream()), 'plm_probe_ds_read_tr16')
  return out.view(64, 4)


def probe_mfma32(A, B):
  _need(A, F32, 'probe.A', 2)
  _need(B, F32, 'probe.B', 2)
  out = torch.empty((32, 32), dtype=F32, device=A.device)
  _lib.check(_lib.load().plm_probe_mfma32(_p(A), _p(B), _p(out), _stream()), 'plm_probe_mfma32')
  return out

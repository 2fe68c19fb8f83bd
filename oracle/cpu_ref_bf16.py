"""bf16-emulating mode of the CPU oracle.  TEST INFRASTRUCTURE ONLY (same rules as cpu_ref.py).

``cpu_ref.py`` restates the reference's CPU path (pure fp32, engine/engine.py:75 ``nullcontext`` on CPU).  The
reference's GPU path is the bf16-autocast flow, and that is what the HIP kernels
implement.  A bf16 trajectory and an fp32 trajectory separate once the optimizer has moved the weights (AdamW's
early updates are sign-like), so "GPU loss vs fp32 oracle" alone cannot tell inherent bf16 drift from a kernel that is
wrong by 1e-3.  This file closes that gap: the SAME algorithm as cpu_ref.py (it reuses its tables, masks, schedule
and optimizer), evaluated in fp32 on the CPU, but rounded to bf16 (round-to-nearest-even) at exactly the points
where plainlm_amd/functional.py + the kernels round:

  * every Linear consumes bf16 activations and bf16 weight shadows, accumulates in fp32, emits bf16
    (forward and dX); weight gradients stay fp32 (gemm_tn out, functional.LinearFn);
  * RMSNorm: fp32 statistics on the fp32 residual stream, bf16(x * rstd * w) out; backward emits the fp32
    residual gradient plus its bf16 copy for the branch (elementwise.hip rmsnorm_fwd/bwd_kernel);
  * RoPE: fp32 rotation of the bf16 projection, bf16 result; inverse rotation of dQ/dK in fp32 (attn.hip);
  * attention: fp32 scores, online softmax over 64-key tiles with the probabilities rounded to bf16 for P.V
    while the row sum accumulates the un-rounded fp32 values; backward recomputes P from the base-2 LSE,
    rounds P and dS to bf16 (attn.hip attn_fwd/bwd kernels);
  * SwiGLU: the bf16 chain documented in elementwise.hip (s = bf16(silu(x)), out = bf16(s*z), ...);
  * cross-entropy on bf16 logits, dlogits = bf16((softmax - onehot) / M) (ce.hip), upstream scale applied by
    the two head GEMMs (functional.HeadLossFn).

What it cannot reproduce bit for bit: fp32 summation order inside the MFMA GEMMs / reductions and the 1-ulp
hardware exp2 / rcp / rsqrt.  Those perturb a value BEFORE a bf16 rounding by ~1e-7 relative, so they flip a
rounding for ~1e-4 of the elements - two orders of magnitude below the every-element bf16 noise that separates
either trajectory from the fp32 one.  It is pinned transitively: with rounding disabled (``round_bf16=False``) it
must agree with cpu_ref.py - itself pinned against the reference's golden vectors - to fp32 round-off
(tests/test_oracle_golden.py::test_bf16_mode_without_rounding_equals_fp32_oracle).
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import cpu_ref as O

Tensor = torch.Tensor
LOG2E = 1.4426950408889634


class _Round:
  """rb(x): fp32 -> bf16 (RNE) -> fp32, or the identity when emulation is off."""

  def __init__(self, on: bool):
    self.on = on

  def __call__(self, x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).float() if self.on else x


def _rms_fwd(r: Tensor, w: Tensor, eps: float, rb):
  rstd = torch.rsqrt(r.pow(2).mean(-1, keepdim=True) + eps)
  return rb((r * rstd) * w), rstd


def _rms_bwd(dy: Tensor, x: Tensor, w: Tensor, rstd: Tensor, gin: Optional[Tensor]):
  """elementwise.hip rmsnorm_bwd_kernel: a = dy*w; coef = sum(a*x) * rstd^3 / d; dx = rstd*a - x*coef (+ gin)."""
  d = x.shape[-1]
  a = dy * w
  coef = (a * x).sum(-1, keepdim=True) * rstd * rstd * rstd / d
  dx = rstd * a - x * coef
  if gin is not None:
    dx = dx + gin
  dw = (dy * (x * rstd)).sum(0)
  return dx, dw


def _rope(x: Tensor, cos: Tensor, sin: Tensor, sgn: float, rb) -> Tensor:
  """x [B, T, nh, hd] (bf16-exact fp32); interleaved pairs, fp32 math, bf16 result.  sgn = -1: inverse rotation."""
  B, T, nh, hd = x.shape
  xf = x.reshape(B, T, nh, hd // 2, 2)
  a, b = xf[..., 0], xf[..., 1]
  c = cos[:T].reshape(1, T, 1, hd // 2)
  s = sin[:T].reshape(1, T, 1, hd // 2) * sgn
  return rb(torch.stack([a * c - b * s, b * c + a * s], dim=-1).reshape(B, T, nh, hd))


def _attn_fwd(q: Tensor, k: Tensor, v: Tensor, allow: Tensor, rb, key_tile: int = 64):
  """q, k, v [B, nh, T, hd]; allow bool [B or 1, 1, T, T].  Online softmax over key tiles like attn_fwd_kernel:
  running max m, l += sum of fp32 p, O += bf16(p) V.  Returns (O bf16-rounded [B, nh, T, hd], base-2 LSE)."""
  B, nh, T, hd = q.shape
  c2 = (1.0 / math.sqrt(hd)) * LOG2E
  m = torch.full((B, nh, T, 1), float('-inf'))
  l = torch.zeros(B, nh, T, 1)
  o = torch.zeros(B, nh, T, hd)
  for k0 in range(0, T, key_tile):
    k1 = min(T, k0 + key_tile)
    s = torch.matmul(q, k[:, :, k0:k1].transpose(-1, -2))
    s = s.masked_fill(~allow[..., k0:k1], float('-inf'))
    m_new = torch.maximum(m, s.max(-1, keepdim=True).values)
    m_safe = torch.where(torch.isinf(m_new), torch.zeros_like(m_new), m_new)
    alpha = torch.exp2((m - m_safe) * c2)
    p = torch.exp2(s * c2 - m_safe * c2)
    l = l * alpha + p.sum(-1, keepdim=True)
    o = o * alpha + torch.matmul(rb(p), v[:, :, k0:k1])
    m = m_new
  return rb(o / l), m * c2 + torch.log2(l)


def _attn_bwd(q, k, v, o, do, lse2, allow, rb):
  """attn_bwd_*_kernel: delta = rowsum(dO*O); P = exp2(S*c2 - LSE2); dV = bf16(P)^T dO; dS = bf16(P*(dP - delta));
  dQ = scale * dS K; dK = scale * dS^T Q (all fp32 accumulation; scale = 1/sqrt(hd) is a power of two for hd = 64)."""
  hd = q.shape[-1]
  scale = 1.0 / math.sqrt(hd)
  c2 = scale * LOG2E
  delta = (do * o).sum(-1, keepdim=True)
  s = torch.matmul(q, k.transpose(-1, -2))
  p = torch.exp2(s * c2 - lse2).masked_fill(~allow, 0.0)
  dp = torch.matmul(do, v.transpose(-1, -2))
  ds = rb(p * (dp - delta))
  dv = torch.matmul(rb(p).transpose(-1, -2), do)
  dq = torch.matmul(ds, k) * scale
  dk = torch.matmul(ds.transpose(-1, -2), q) * scale
  return dq, dk, dv


def _sigmoid(x):
  return 1.0 / (1.0 + torch.exp2(-LOG2E * x))


def loss_and_grads(params: Dict[str, Tensor], cfg: O.OracleConfig, ids: Tensor, targets: Tensor,
                   doc_start: Optional[Tensor] = None, scale: float = 1.0, round_bf16: bool = True):
  """Same contract as cpu_ref.loss_and_grads - (un-scaled mean loss, {name: d(scale*loss)/d param} in fp32) - with the
  forward AND the hand-written backward of plainlm_amd/functional.py restated op by op (models/transformer.py:39-114,
  engine/engine.py:109-120)."""
  rb = _Round(round_bf16)
  B, T = ids.shape
  d, nh, hd, h, L, V = cfg.dim, cfg.n_heads, cfg.head_dim, cfg.hidden, cfg.n_layers, cfg.vocab_size
  M = B * T
  eps = cfg.rmsnorm_eps
  cos, sin = O.rope_table(hd, cfg.seq_len, cfg.rope_theta)
  if doc_start is None:
    allow = torch.ones(T, T, dtype=torch.bool).tril().view(1, 1, T, T)
  else:
    allow = O.mask_from_doc_start(doc_start).view(B, 1, T, T)
  W = {n: (rb(t) if t.dim() == 2 and not n.startswith('embed') else t) for n, t in params.items()}  # bf16 weight shadows
  head_name = 'embed_tokens.weight' if cfg.tie_embeddings else 'lm_head.weight'
  Wh = rb(params[head_name])

  # ---------------- forward ----------------
  flat_ids = ids.reshape(-1)
  x = params['embed_tokens.weight'][flat_ids]  # fp32 [M, d]
  branch = None
  saved = []
  for i in range(L):
    p = f'layers.{i}.'
    r1 = x if branch is None else x + branch
    n1, rstd1 = _rms_fwd(r1, params[p + 'attn_norm.weight'], eps, rb)
    qkv = rb(n1 @ W[p + 'attn.w_qkv.weight'].t())
    q, k, v = (t.reshape(B, T, nh, hd) for t in qkv.split(d, dim=1))
    qr, kr = _rope(q, cos, sin, 1.0, rb), _rope(k, cos, sin, 1.0, rb)
    qh, kh, vh = (t.transpose(1, 2) for t in (qr, kr, v))
    oh, lse2 = _attn_fwd(qh, kh, vh, allow, rb)
    a = oh.transpose(1, 2).reshape(M, d)
    ao = rb(a @ W[p + 'attn.w_out.weight'].t())
    r2 = r1 + ao
    n2, rstd2 = _rms_fwd(r2, params[p + 'mlp_norm.weight'], eps, rb)
    u = rb(n2 @ W[p + 'mlp.fc1.weight'].t())
    ux, uz = u[:, :h], u[:, h:]
    s = rb(ux * _sigmoid(ux))
    g = rb(s * uz)
    mo = rb(g @ W[p + 'mlp.fc2.weight'].t())
    saved.append(dict(r1=r1, rstd1=rstd1, n1=n1, qh=qh, kh=kh, vh=vh, oh=oh, lse2=lse2, a=a, r2=r2, rstd2=rstd2, n2=n2,
                      ux=ux, uz=uz, s=s, g=g, first=branch is None))
    x, branch = r2, mo
  rf = x + branch
  y, rstdf = _rms_fwd(rf, params['out_norm.weight'], eps, rb)
  logits = rb(y @ Wh.t())  # [M, V]
  tg = targets.reshape(-1)
  lse = torch.logsumexp(logits, dim=-1)
  loss = (lse - logits.gather(-1, tg.view(-1, 1)).squeeze(-1)).mean()

  # ---------------- backward ----------------
  grads: Dict[str, Tensor] = {}

  def add(name, gval):
    grads[name] = gval if name not in grads else grads[name] + gval

  prob = torch.exp(logits - lse.view(-1, 1))
  prob[torch.arange(M), tg] -= 1.0
  dlog = rb(prob * (1.0 / M))              # ce.hip: bf16((softmax - onehot) * grad_scale)
  dy = rb(scale * (dlog @ Wh))              # head dX, upstream gradient folded in as the GEMM's alpha
  add(head_name, scale * (dlog.t() @ y))    # head dW, fp32
  dx, dw = _rms_bwd(dy, rf, params['out_norm.weight'], rstdf, None)
  add('out_norm.weight', dw)
  dres, dbranch = dx, rb(dx)               # AddNormFn.backward: (g_xout, bf16 copy for the branch)
  for i in range(L - 1, -1, -1):
    p = f'layers.{i}.'
    sv = saved[i]
    # fc2
    add(p + 'mlp.fc2.weight', dbranch.t() @ sv['g'])
    dg = rb(dbranch @ W[p + 'mlp.fc2.weight'])
    # swiglu (elementwise.hip swiglu_bwd_kernel)
    sig = _sigmoid(sv['ux'])
    dsg = rb(dg * sv['uz'])
    dz = rb(dg * sv['s'])
    dxg = rb(dsg * (sig * (1.0 + sv['ux'] * (1.0 - sig))))
    du = torch.cat([dxg, dz], dim=1)
    # fc1
    add(p + 'mlp.fc1.weight', du.t() @ sv['n2'])
    dn2 = rb(du @ W[p + 'mlp.fc1.weight'])
    # mlp_norm (+ residual gradient)
    dx, dw = _rms_bwd(dn2, sv['r2'], params[p + 'mlp_norm.weight'], sv['rstd2'], dres)
    add(p + 'mlp_norm.weight', dw)
    dres, dao = dx, rb(dx)
    # w_out
    add(p + 'attn.w_out.weight', dao.t() @ sv['a'])
    da = rb(dao @ W[p + 'attn.w_out.weight'])
    doh = da.reshape(B, T, nh, hd).transpose(1, 2)
    dq, dk, dv = _attn_bwd(sv['qh'], sv['kh'], sv['vh'], sv['oh'], doh, sv['lse2'], allow, rb)
    dq = _rope(dq.transpose(1, 2), cos, sin, -1.0, rb)  # gradient w.r.t. the PRE-rotation projection
    dk = _rope(dk.transpose(1, 2), cos, sin, -1.0, rb)
    dv = rb(dv.transpose(1, 2))
    dqkv = torch.cat([t.reshape(M, d) for t in (dq, dk, dv)], dim=1)
    # w_qkv
    add(p + 'attn.w_qkv.weight', dqkv.t() @ sv['n1'])
    dn1 = rb(dqkv @ W[p + 'attn.w_qkv.weight'])
    if sv['first']:  # NormFn: the embedding output feeds the norm and the residual stream
      dx, dw = _rms_bwd(dn1, sv['r1'], params[p + 'attn_norm.weight'], sv['rstd1'], None)
      add(p + 'attn_norm.weight', dw)
      dres = dres + dx
      dbranch = None
    else:
      dx, dw = _rms_bwd(dn1, sv['r1'], params[p + 'attn_norm.weight'], sv['rstd1'], dres)
      add(p + 'attn_norm.weight', dw)
      dres, dbranch = dx, rb(dx)
  demb = torch.zeros_like(params['embed_tokens.weight'])
  demb.index_add_(0, flat_ids, dres)
  add('embed_tokens.weight', demb)
  return loss.detach(), grads


class OracleEngineBF16(O.OracleEngine):
  """cpu_ref.OracleEngine (engine/engine.py:93-141: accumulation, clip, AdamW, warmup-cosine - all fp32 on both sides)
  with the bf16-emulating fwd+bwd."""

  def __init__(self, *a, round_bf16: bool = True, **k):
    super().__init__(*a, **k)
    self.round_bf16 = round_bf16

  def _loss_and_grads(self, ids, tgt, ds):
    return loss_and_grads(self.params, self.cfg, ids, tgt, ds, scale=1.0 / self.accum, round_bf16=self.round_bf16)

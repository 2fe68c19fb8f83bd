"""CPU fp32 oracle for the plainLM hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain PyTorch eager, fp32,
CPU) of the arithmetic that the reference executes on its training hot path.
It exists to CHECK the HIP kernels; it is never on the product path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md §4), so this oracle is pinned against outputs of the reference
itself, imported in the build container by ``tests/golden/make_golden.py``;
the resulting vectors are committed under ``tests/golden/`` and
``tests/test_oracle_golden.py`` re-checks this file against them on every run.

Every function cites the reference lines it restates (paths relative to the
reference root).  The hot-path arithmetic itself lives in third-party PyTorch
(``torch>=2.6.0``, pyproject.toml:16; 2.10.0 in the build container), so the
"published algorithm" restated here is torch's CPU fp32 semantics for
``F.scaled_dot_product_attention``, ``nn.Linear``, ``nn.Embedding``,
``F.silu``, ``CrossEntropyLoss`` and ``torch.optim.AdamW``.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from fractions import Fraction
from typing import Dict, List, Optional, Sequence

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class OracleConfig:
  """Shape parameters of the decoder (models/transformer.py:13-23)."""

  vocab_size: int
  seq_len: int
  dim: int
  n_layers: int
  n_heads: int
  expand: float = 8.0 / 3.0
  rmsnorm_eps: float = 1e-6
  tie_embeddings: bool = False
  rope_theta: float = 500000.0  # models/transformer.py:99
  mlp: str = 'glu'              # models/transformer.py:26 MLP_CLASSES: 'glu' (every shipped config) | 'mlp' | 'mlp_relu_sq'

  @property
  def head_dim(self) -> int:
    return self.dim // self.n_heads

  @property
  def hidden(self) -> int:
    return glu_hidden_dim(self.dim, self.expand)


def glu_hidden_dim(dim: int, expand: float, multiple_of: int = 256) -> int:
  """models/transformer.py:74 (int(expand*dim)) then models/components.py:48 (round up)."""
  h = int(expand * dim)
  return multiple_of * ((h + multiple_of - 1) // multiple_of)


def parse_expand(expand) -> float:
  """models/construct.py:15 — the YAML carries a string fraction such as '8/3'."""
  return float(Fraction(expand))


def param_names(cfg: OracleConfig) -> List[str]:
  """Parameter names in ``named_parameters()`` order (probed)."""
  names = ['embed_tokens.weight']
  for i in range(cfg.n_layers):
    names += [
      f'layers.{i}.attn.w_qkv.weight',
      f'layers.{i}.attn.w_out.weight',
      f'layers.{i}.attn_norm.weight',
      f'layers.{i}.mlp.fc1.weight',
      f'layers.{i}.mlp.fc2.weight',
      f'layers.{i}.mlp_norm.weight',
    ]
  names.append('out_norm.weight')
  if not cfg.tie_embeddings:
    names.append('lm_head.weight')
  return names


def param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
  d, h, V = cfg.dim, cfg.hidden, cfg.vocab_size
  shapes = {'embed_tokens.weight': (V, d), 'out_norm.weight': (d,)}
  if not cfg.tie_embeddings:
    shapes['lm_head.weight'] = (V, d)
  for i in range(cfg.n_layers):
    shapes[f'layers.{i}.attn.w_qkv.weight'] = (3 * d, d)
    shapes[f'layers.{i}.attn.w_out.weight'] = (d, d)
    shapes[f'layers.{i}.attn_norm.weight'] = (d,)
    shapes[f'layers.{i}.mlp.fc1.weight'] = ((2 * h if cfg.mlp == 'glu' else h), d)  # components.py:50 (GLU: gate | up) vs :35, :65
    shapes[f'layers.{i}.mlp.fc2.weight'] = (d, h)
    shapes[f'layers.{i}.mlp_norm.weight'] = (d,)
  return shapes


def init_params(cfg: OracleConfig, seed: int = 0) -> Dict[str, Tensor]:
  """Same distributions as models/transformer.py:116-129: N(0, 0.02) for every
  Linear/Embedding weight, N(0, 0.02/sqrt(2L)) for w_out and fc2, ones for norms.
  The RNG stream is NOT the reference's; parity runs share tensors instead."""
  g = torch.Generator().manual_seed(seed)
  out = {}
  for name, shape in ((n, param_shapes(cfg)[n]) for n in param_names(cfg)):
    if 'norm' in name:
      out[name] = torch.ones(shape)
    else:
      std = 0.02
      if name.endswith('fc2.weight') or name.endswith('w_out.weight'):
        std = 0.02 / math.sqrt(2 * cfg.n_layers)
      out[name] = torch.randn(shape, generator=g) * std
  return out


# --------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------
def rope_table(head_dim: int, seq_len: int, theta: float = 500000.0):
  """cos/sin tables [T, hd/2] fp32 (models/embeddings.py:8-12).
  angle[t, i] = t * theta**(-(2i)/hd)."""
  expo = torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim
  inv = 1.0 / (theta**expo)
  pos = torch.arange(seq_len, dtype=torch.float32)
  ang = torch.outer(pos, inv).float()
  return torch.cos(ang), torch.sin(ang)


def rope_apply(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
  """Interleaved-pair rotation (models/embeddings.py:21-30).
  x: [B, T, nh, hd]; pair i = (x[..., 2i], x[..., 2i+1]);
  out = (a*cos - b*sin, b*cos + a*sin), computed in fp32, cast back to x.dtype."""
  B, T, nh, hd = x.shape
  xf = x.float().reshape(B, T, nh, hd // 2, 2)
  a, b = xf[..., 0], xf[..., 1]
  c = cos[:T].reshape(1, T, 1, hd // 2)
  s = sin[:T].reshape(1, T, 1, hd // 2)
  out = torch.stack([a * c - b * s, b * c + a * s], dim=-1).reshape(B, T, nh, hd)
  return out.to(x.dtype)


def rmsnorm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
  """models/components.py:22-28: x * rsqrt(mean(x^2) + eps) in fp32, cast to x.dtype, times w."""
  xf = x.float()
  y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
  return y.to(x.dtype) * w


def doc_start_from_lengths(docs_lengths: Sequence[Sequence[int]], seq_len: int) -> Tensor:
  """Per-token index of the first token of its document, int32 [B, T].

  Replaces the materialised mask of data/datasets/data_prep_utils.py:7-23 as
  cropped by engine/engine.py:21-23: lengths sum to T+1, the mask is the
  block-diagonal of lower-triangular blocks, cropped to [:T, :T].  Token i may
  attend to j iff doc_start[i] <= j <= i."""
  rows = []
  for lens in docs_lengths:
    if sum(lens) != seq_len + 1:
      raise ValueError('Sum of doc_boundaries does not match max_seq_length.')
    row, start = [], 0
    for n in lens:
      row += [start] * n
      start += n
    rows.append(row[:seq_len])
  return torch.tensor(rows, dtype=torch.int32)


def mask_from_doc_start(doc_start: Tensor) -> Tensor:
  """bool [B, T, T], True = may attend (the reference's mask layout)."""
  B, T = doc_start.shape
  j = torch.arange(T).view(1, 1, T)
  i = torch.arange(T).view(1, T, 1)
  return (j <= i) & (j >= doc_start.view(B, T, 1).long())


def attention(q: Tensor, k: Tensor, v: Tensor, doc_start: Optional[Tensor] = None) -> Tensor:
  """softmax(q k^T / sqrt(hd) + mask) v with q,k,v [B, T, nh, hd]
  (models/transformer.py:49-65; default SDPA scale 1/sqrt(hd), causal when no
  mask, boolean mask True = attend otherwise).  Returns [B, T, nh*hd]."""
  B, T, nh, hd = q.shape
  qh, kh, vh = (t.transpose(1, 2) for t in (q, k, v))  # [B, nh, T, hd]
  s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(hd)
  if doc_start is None:
    allow = torch.ones(T, T, dtype=torch.bool).tril().view(1, 1, T, T)
  else:
    allow = mask_from_doc_start(doc_start).view(B, 1, T, T)
  s = s.masked_fill(~allow, float('-inf'))
  p = torch.softmax(s, dim=-1)
  o = torch.matmul(p, vh)  # [B, nh, T, hd]
  return o.transpose(1, 2).reshape(B, T, nh * hd)


def swiglu(u: Tensor, hidden: int) -> Tensor:
  """models/components.py:55-56: split fc1 output into x = [:h], z = [h:]; silu(x)*z."""
  x, z = u[..., :hidden], u[..., hidden:]
  return torch.nn.functional.silu(x) * z


def mlp_act(u: Tensor, hidden: int, kind: str) -> Tensor:
  """The activation between fc1 and fc2 of the three MLP classes (models/components.py): 'glu' :55-56 silu(x) * z, 'mlp' :40 silu(u),
  'mlp_relu_sq' :70 relu(u)^2."""
  if kind == 'glu':
    return swiglu(u, hidden)
  if kind == 'mlp':
    return torch.nn.functional.silu(u)
  if kind == 'mlp_relu_sq':
    return torch.relu(u).pow(2)
  raise ValueError(f'unknown mlp class {kind!r}')


def cross_entropy(logits: Tensor, targets: Tensor) -> Tensor:
  """engine/engine.py:81,111: mean over tokens of logsumexp(l) - l[target], fp32."""
  lf = logits.float()
  lse = torch.logsumexp(lf, dim=-1)
  picked = lf.gather(-1, targets.view(-1, 1)).squeeze(-1)
  return (lse - picked).mean()


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
def forward(params: Dict[str, Tensor], cfg: OracleConfig, ids: Tensor, doc_start: Optional[Tensor] = None) -> Tensor:
  """models/transformer.py:108-114 and :79-83, :39-67.  ids int64 [B, T] -> logits [B, T, V]."""
  B, T = ids.shape
  d, nh, hd, h = cfg.dim, cfg.n_heads, cfg.head_dim, cfg.hidden
  cos, sin = rope_table(hd, cfg.seq_len, cfg.rope_theta)
  x = params['embed_tokens.weight'][ids]  # transformer.py:110
  for i in range(cfg.n_layers):
    p = f'layers.{i}.'
    n1 = rmsnorm(x, params[p + 'attn_norm.weight'], cfg.rmsnorm_eps)
    qkv = n1 @ params[p + 'attn.w_qkv.weight'].t()
    q, k, v = (t.reshape(B, T, nh, hd) for t in qkv.split(d, dim=2))
    q, k = rope_apply(q, cos, sin), rope_apply(k, cos, sin)
    a = attention(q, k, v, doc_start)
    x = x + a @ params[p + 'attn.w_out.weight'].t()
    n2 = rmsnorm(x, params[p + 'mlp_norm.weight'], cfg.rmsnorm_eps)
    u = n2 @ params[p + 'mlp.fc1.weight'].t()
    x = x + mlp_act(u, h, cfg.mlp) @ params[p + 'mlp.fc2.weight'].t()
  xn = rmsnorm(x, params['out_norm.weight'], cfg.rmsnorm_eps)
  head = params['embed_tokens.weight'] if cfg.tie_embeddings else params['lm_head.weight']
  return xn @ head.t()


def loss_fn(params, cfg: OracleConfig, ids: Tensor, targets: Tensor, doc_start: Optional[Tensor] = None) -> Tensor:
  logits = forward(params, cfg, ids, doc_start)
  return cross_entropy(logits.reshape(-1, cfg.vocab_size), targets.reshape(-1))


def loss_and_grads(params, cfg: OracleConfig, ids, targets, doc_start=None, scale: float = 1.0):
  """fwd+bwd through torch autograd on the fp32 restatement. Returns (loss, {name: grad})."""
  leaves = {n: t.detach().clone().requires_grad_(True) for n, t in params.items()}
  loss = loss_fn(leaves, cfg, ids, targets, doc_start)
  (loss * scale).backward()
  return loss.detach(), {n: t.grad for n, t in leaves.items()}


# --------------------------------------------------------------------------
# engine (one optimizer step incl. accumulation, clip, AdamW, LR schedule)
# --------------------------------------------------------------------------
def warmup_cosine_lr(t: int, lr_start: float, lr_max: float, lr_end: float, warmup_steps: int, T: int) -> float:
  """optim/lr_schedule.py:42-49."""
  if t <= warmup_steps:
    return lr_start + (lr_max - lr_start) / warmup_steps * t
  if t <= T:
    prog = (t - warmup_steps) / (T - warmup_steps)
    return lr_end + 0.5 * (lr_max - lr_end) * (1 + math.cos(math.pi * prog))
  return lr_end


def decay_mask(names: Sequence[str]) -> Dict[str, bool]:
  """models/construct.py:54-58: weight decay for names without 'bias'/'norm'."""
  return {n: ('bias' not in n and 'norm' not in n) for n in names}


class OracleEngine:
  """Restates engine/engine.py:93-141 for CPU fp32 with AdamW + warmup-cosine
  (optim/init_optim.py:14-21, :94-102; optim/lr_schedule.py:29-54).

  ``step(batch)`` takes ``{'input_ids': int64 [B, T+1], ('docs_lengths': ...)}``
  and returns the un-divided micro-batch loss, exactly like the reference."""

  def __init__(self, params, cfg: OracleConfig, *, lr, weight_decay, beta1, beta2, grad_clip, accum,
               steps_budget, warmup_steps, lr_start=0.0, lr_end=1e-5, eps=1e-8, intra_doc_masking=False):
    self.params = {n: t.detach().clone() for n, t in params.items()}
    self.cfg = cfg
    self.lr_max, self.wd, self.b1, self.b2, self.eps = lr, weight_decay, beta1, beta2, eps
    self.grad_clip, self.accum = grad_clip, accum
    self.T = steps_budget
    self.warmup = warmup_steps if isinstance(warmup_steps, int) else int(warmup_steps * steps_budget)
    self.lr_start, self.lr_end = lr_start, lr_end
    self.intra_doc_masking = intra_doc_masking
    self.lr = lr_start  # lr_schedule.py:40 sets lr_start at construction
    self.sched_iter = 0
    self.opt_step = 0
    self.m = {n: torch.zeros_like(t) for n, t in self.params.items()}
    self.v = {n: torch.zeros_like(t) for n, t in self.params.items()}
    self.grads = None
    self.accumulated = 0
    self.decay = decay_mask(list(self.params))

  def step(self, batch) -> Tensor:
    T = self.cfg.seq_len
    ids = batch['input_ids'][:, :T]
    tgt = batch['input_ids'][:, 1:T + 1]
    ds = doc_start_from_lengths(batch['docs_lengths'], T) if self.intra_doc_masking else None
    self.accumulated += 1
    loss, g = self._loss_and_grads(ids, tgt, ds)
    if self.grads is None:
      self.grads = g
    else:
      for n in g:
        self.grads[n] += g[n]
    if self.accumulated == self.accum:
      self.accumulated = 0
      self._optimizer_step()
    return loss

  def _loss_and_grads(self, ids, tgt, ds):
    """fwd+bwd of one micro-batch with the 1/accum factor of engine.py:118 (overridden by the bf16-emulating mode)."""
    return loss_and_grads(self.params, self.cfg, ids, tgt, ds, scale=1.0 / self.accum)

  def _optimizer_step(self):
    g = self.grads
    if self.grad_clip:
      # torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
      total = torch.sqrt(sum((t.double() ** 2).sum() for t in g.values())).float()
      coef = torch.clamp(self.grad_clip / (total + 1e-6), max=1.0)
      for t in g.values():
        t.mul_(coef)
    self.opt_step += 1
    bc1 = 1 - self.b1 ** self.opt_step
    bc2 = 1 - self.b2 ** self.opt_step
    for n, p in self.params.items():
      wd = self.wd if self.decay[n] else 0.0
      p.mul_(1 - self.lr * wd)
      self.m[n].mul_(self.b1).add_(g[n], alpha=1 - self.b1)
      self.v[n].mul_(self.b2).addcmul_(g[n], g[n], value=1 - self.b2)
      denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(self.eps)
      p.addcdiv_(self.m[n], denom, value=-self.lr / bc1)
    self.grads = None
    self.sched_iter += 1
    self.lr = warmup_cosine_lr(self.sched_iter, self.lr_start, self.lr_max, self.lr_end, self.warmup, self.T)

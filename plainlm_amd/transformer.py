"""Llama-style decoder with the reference's module API, running on hand-written gfx950 kernels.

Drop-in contract: same constructor config, same parameter names /
shapes / dtypes / init distributions, ``model(inputs[B,T] int64, attn_mask) -> logits[B,T,V]``
called positionally, ``count_params``, ``tie_weights``; ``state_dict`` round-trips with
the reference (models/transformer.py:86-140).  There is no CPU path: calling the model
with CPU tensors, or without the built HIP library, raises.

Extra (not in the reference): ``model.loss(inputs, targets, attn_mask)`` fuses lm_head
with cross-entropy; ``attn_mask`` may also be an int32 ``doc_start[B,T]`` tensor, which
avoids materialising the O(T^2) boolean mask of engine/engine.py:21-23.
"""

import math
import os
from dataclasses import dataclass

import torch
from torch import nn

from . import functional as Fn
from . import ops


@dataclass
class ModelConfig:
  """Field-for-field the reference's ModelConfig (models/transformer.py:13-23)."""
  vocab_size: int
  seq_len: int
  dim: int
  expand: float
  n_layers: int
  n_heads: int
  mlp: str = 'mlp'
  rmsnorm_eps: float = 1e-6
  tie_embeddings: bool = False


def rope_tables(head_dim, seq_len, theta=500000.0):
  """cos/sin [T, hd/2] fp32, angle[t,i] = t * theta^(-2i/hd) (models/embeddings.py:8-12;
  theta = 500000 at models/transformer.py:99).  Built on the host like the reference."""
  expo = torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim
  ang = torch.outer(torch.arange(seq_len, dtype=torch.float32), 1.0 / (theta ** expo)).float()
  return torch.cos(ang).contiguous(), torch.sin(ang).contiguous()


class HipLinear(nn.Module):
  """Bias-free Linear: fp32 master ``weight[out, in]`` + lazily refreshed bf16 shadows
  (``[out,in]`` for y = x W^T and ``[in,out]`` for dX), the analogue of autocast's per-forward
  weight cast (engine/engine.py:75)."""

  def __init__(self, in_features, out_features, pad_out_to=8):
    super().__init__()
    self.in_features, self.out_features = in_features, out_features
    # the transposed shadow [in, out_pad] is zero-padded along `out` so dX GEMMs see a K that is a multiple of 64
    self.out_pad = -(-out_features // pad_out_to) * pad_out_to
    self.weight = nn.Parameter(torch.empty(out_features, in_features))
    self.sink = None
    self._shadow = None
    self._shadow_key = None

  def _key(self):
    w = self.weight
    return (w.data_ptr(), w._version, w.device)

  def stale_item(self):
    """None when the shadows are current, else the (src, out, out_t) triple that refreshes them (buffers allocated here);
    the caller runs the cast and then calls mark_fresh()."""
    w = self.weight
    if self._shadow is not None and self._key() == self._shadow_key:
      return None
    if self._shadow is None or self._shadow[0].device != w.device:
      self._shadow = (torch.empty((self.out_features, self.in_features), dtype=torch.bfloat16, device=w.device),
                      torch.zeros((self.in_features, self.out_pad), dtype=torch.bfloat16, device=w.device))
    return (w.detach(), self._shadow[0], self._shadow[1])

  def mark_fresh(self):
    self._shadow_key = self._key()

  def shadow(self):
    """(bf16 W [out, in], bf16 W^T [in, out_pad] with zero pad columns), re-cast when the master weight changed."""
    item = self.stale_item()
    if item is not None:
      ops.cast_bf16_t(item[0], out=item[1], out_t=item[2])
      self.mark_fresh()
    return self._shadow

  def invalidate(self):
    self._shadow_key = None

  def forward(self, x):
    lead = x.shape[:-1]
    y = Fn.LinearFn.apply(x.reshape(-1, self.in_features), self.weight, self)
    return y.view(*lead, self.out_features)

  def extra_repr(self):
    return f'in_features={self.in_features}, out_features={self.out_features}, bias=False'


class HipEmbedding(nn.Module):
  def __init__(self, num_embeddings, embedding_dim):
    super().__init__()
    self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
    self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
    self.sink = None

  def forward(self, ids):
    flat = ids.reshape(-1).contiguous()
    out = Fn.EmbedFn.apply(flat, self.weight, self)
    return out.view(*ids.shape, self.embedding_dim)


class RMSNorm(nn.Module):
  """models/components.py:16-28.  Standalone ``forward`` returns the bf16 value the next
  Linear consumes under autocast (bf16(norm(x) * w))."""

  def __init__(self, dim, eps=1e-6):
    super().__init__()
    self.eps = eps
    self.weight = nn.Parameter(torch.ones(dim))
    self.sink = None

  def forward(self, x):
    lead = x.shape
    _, y = Fn.NormFn.apply(x.reshape(-1, lead[-1]).contiguous(), self.weight, self)
    return y.view(lead)


class GLU(nn.Module):
  """SwiGLU MLP with fused fc1 [2h, d] (models/components.py:43-56)."""

  def __init__(self, dim, hidden_dim, multiple_of=256):
    super().__init__()
    hidden_dim = multiple_of * ((hidden_dim + multiple_of - 1) // multiple_of)
    self.hidden_dim = hidden_dim
    self.fc1 = HipLinear(dim, 2 * hidden_dim)
    self.fc2 = HipLinear(hidden_dim, dim)

  def apply_fn(self, x2d):
    return Fn.SwiGLUMLPFn.apply(x2d, self.fc1.weight, self.fc2.weight, self.fc1, self.fc2)

  def forward(self, x):
    lead = x.shape[:-1]
    return self.apply_fn(x.reshape(-1, x.shape[-1])).view(*lead, -1)


class MLP(nn.Module):
  """models/components.py:31-40 (`MLP`: fc2(silu(fc1 x))) and :59-70 (`MLPReluSquared`: fc2(relu(fc1 x)^2)): fc1 [h, d], fc2 [d, h]."""
  kind = 'silu'

  def __init__(self, dim, hidden_dim, multiple_of=256):
    super().__init__()
    hidden_dim = multiple_of * ((hidden_dim + multiple_of - 1) // multiple_of)
    self.hidden_dim = hidden_dim
    self.fc1 = HipLinear(dim, hidden_dim)
    self.fc2 = HipLinear(hidden_dim, dim)

  def apply_fn(self, x2d):
    return Fn.PlainMLPFn.apply(x2d, self.fc1.weight, self.fc2.weight, self.fc1, self.fc2, self.kind)

  def forward(self, x):
    lead = x.shape[:-1]
    return self.apply_fn(x.reshape(-1, x.shape[-1])).view(*lead, -1)


class MLPReluSquared(MLP):
  kind = 'relu_sq'


MLP_CLASSES = {'mlp': MLP, 'glu': GLU, 'mlp_relu_sq': MLPReluSquared}  # models/transformer.py:26


class Attention(nn.Module):
  """models/transformer.py:29-67 — fused QKV projection, RoPE + SDPA in one kernel, output projection."""

  def __init__(self, cfg):
    super().__init__()
    assert cfg.dim % cfg.n_heads == 0
    self.n_heads = cfg.n_heads
    self.head_dim = cfg.dim // cfg.n_heads
    self.w_qkv = HipLinear(cfg.dim, 3 * cfg.dim)
    self.w_out = HipLinear(cfg.dim, cfg.dim)

  def forward(self, x2d, rope, doc_start, B, T):
    qkv = Fn.QKVRopeFn.apply(x2d, self.w_qkv.weight, self.w_qkv, rope[0], rope[1], B, T, self.n_heads)
    a = Fn.AttnFn.apply(qkv, rope[0], rope[1], doc_start, B, T, self.n_heads)
    return self.w_out(a)


class Block(nn.Module):
  def __init__(self, layer_id, cfg):
    super().__init__()
    if cfg.mlp not in MLP_CLASSES:
      raise NotImplementedError(f"mlp class '{cfg.mlp}': expected one of {sorted(MLP_CLASSES)} (models/transformer.py:26)")
    self.attn = Attention(cfg)
    self.attn_norm = RMSNorm(cfg.dim, cfg.rmsnorm_eps)
    self.mlp = MLP_CLASSES[cfg.mlp](dim=cfg.dim, hidden_dim=int(cfg.expand * cfg.dim))
    self.mlp_norm = RMSNorm(cfg.dim, cfg.rmsnorm_eps)
    self.layer_id = layer_id

  def forward(self, x, branch, rope, doc_start, B, T):
    """x fp32 [M,d] residual stream, branch bf16 [M,d] = previous block's MLP output (or None).
    Returns (x_mid, mlp_out): the residual add of the MLP output is fused into the NEXT norm."""
    if branch is None:
      x, n1 = Fn.NormFn.apply(x, self.attn_norm.weight, self.attn_norm)
    else:
      x, n1 = Fn.AddNormFn.apply(x, branch, self.attn_norm.weight, self.attn_norm)
    a = self.attn(n1, rope, doc_start, B, T)
    x, n2 = Fn.AddNormFn.apply(x, a, self.mlp_norm.weight, self.mlp_norm)
    return x, self.mlp.apply_fn(n2)


class Transformer(nn.Module):
  def __init__(self, cfg):
    super().__init__()
    self.cfg = cfg
    self.n_layers = cfg.n_layers
    if cfg.dim % cfg.n_heads != 0:
      raise ValueError('dim must be divisible by n_heads')
    self.head_dim = cfg.dim // cfg.n_heads
    if self.head_dim not in (32, 64, 128):  # 64: the tuned kernels (every shipped config); 32 / 128: csrc/attn_generic.hip
      raise NotImplementedError(f'head_dim {self.head_dim}: the gfx950 attention kernels are built for head dims 32, 64 and 128')
    if cfg.vocab_size % 8 != 0:  # 16-byte rows of the bf16 lm_head shadows and of the logits (every shipped config: 50280)
      raise NotImplementedError(f'vocab_size {cfg.vocab_size}: must be a multiple of 8 on this path - pad the vocabulary (GPT-2\'s 50257 -> 50264; the '
                                f'reference\'s configs ship 50280), the extra rows are ordinary never-targeted tokens')

    self.embed_tokens = HipEmbedding(cfg.vocab_size, cfg.dim)
    self.layers = nn.ModuleList([Block(idx, cfg) for idx in range(cfg.n_layers)])
    self.out_norm = RMSNorm(cfg.dim, cfg.rmsnorm_eps)
    self.lm_head = HipLinear(cfg.dim, cfg.vocab_size, pad_out_to=64)

    # plain attribute like the reference's freqs_cis: not a buffer, not in state_dict
    self._rope_host = rope_tables(self.head_dim, cfg.seq_len)
    self._rope_dev = None

    self.sink = Fn.GradSink()
    self._flat_grad = None
    self._linears = None
    # loss(): token rows per lm_head + cross-entropy chunk (0 = one [M, V] logits buffer)
    self.head_chunk_rows = int(os.environ.get('PLM_HEAD_CHUNK', '0'))
    self.apply(self._init_weights)
    self._scale_residual_branches()
    if cfg.tie_embeddings:
      self.tie_weights()
    self._wire_sink()

  # ---- init (models/transformer.py:116-129) ---------------------------------
  def _init_weights(self, module):
    if isinstance(module, (HipLinear, HipEmbedding)):
      torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)

  def _scale_residual_branches(self):
    for n, p in self.named_parameters():
      if n.endswith('fc2.weight') or n.endswith('w_out.weight'):
        torch.nn.init.normal_(p, mean=0.0, std=0.02 / math.sqrt(2 * self.n_layers))

  def tie_weights(self):
    self.lm_head.weight = self.embed_tokens.weight

  def count_params(self, non_embedding=True):
    n_params = sum(p.numel() for p in self.parameters())
    if non_embedding:
      n_params -= self.embed_tokens.weight.numel()
      if self.lm_head.weight is not self.embed_tokens.weight:
        n_params -= self.lm_head.weight.numel()
    return n_params

  def _wire_sink(self):
    for m in self.modules():
      if isinstance(m, (HipLinear, HipEmbedding, RMSNorm)):
        m.sink = self.sink

  # ---- flat gradient buffer (our engine / DDP path) ----------------------------
  def enable_main_grad(self):
    """Allocate one flat fp32 gradient buffer; every parameter gets a ``main_grad`` view into it and the backward kernels
    write there directly.  Placement: the RMSNorm weights first, then everything else in ``parameters()`` order -
    [norms | embed_tokens | layer 0 ... | lm_head].  ddp.plan_buckets walks the buffer from its end (= the order gradients
    become ready): lm_head goes first, the layers follow, and the norm weights - whose column sums run as one launch when
    backward reaches the embedding (GradSink.flush_norms) - share the LAST bucket with embed_tokens.  FlatAdamW uses the
    same placement (zero-weight-decay group first)."""
    params = list(self.parameters())
    dev = params[0].device
    total = sum(p.numel() for p in params)
    self._flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
    norm_ids = {id(m.weight) for m in self.modules() if isinstance(m, RMSNorm)}
    spans, off = {}, 0
    for p in sorted(params, key=lambda q: id(q) not in norm_ids):  # stable: norms first, otherwise parameters() order
      p.main_grad = self._flat_grad[off:off + p.numel()].view(p.shape)
      spans[id(p)] = (off, p.numel())
      off += p.numel()
    self._grad_spans = [spans[id(p)] for p in params]  # parameters() order, as ddp.GradReducer expects
    self.sink.enabled = True
    return self._flat_grad

  def grad_writers(self):
    """{id(param): number of backward kernels that write its gradient in one backward pass}: 1 per owning module, so 2
    for a weight tied between lm_head and embed_tokens.  ddp.GradReducer launches a bucket when all writers have reported."""
    n = {}
    for m in self.modules():
      if isinstance(m, (HipLinear, HipEmbedding, RMSNorm)):
        n[id(m.weight)] = n.get(id(m.weight), 0) + 1
    return n

  def grad_groups(self):
    """Parameter indices (parameters() order) of every block's Linear weights: their gradients leave ONE grouped dW launch together,
    so ddp.plan_buckets keeps each list inside one bucket when it fits."""
    index = {id(p): i for i, p in enumerate(self.parameters())}
    return [[index[id(m.weight)] for m in layer.modules() if isinstance(m, HipLinear)] for layer in self.layers]

  def attach_grads(self):
    """Expose the flat buffer through ``p.grad`` for optimizers / clipping."""
    self.sink.flush_dw()
    for p in self.parameters():
      p.grad = p.main_grad

  def linear_modules(self):
    """The HipLinear modules, one per distinct weight (a weight tied between lm_head and embed_tokens appears once)."""
    if self._linears is None:
      seen, self._linears = set(), []
      for m in self.modules():
        if isinstance(m, HipLinear) and id(m.weight) not in seen:  # tied weights: one shadow refresh per call site is enough
          seen.add(id(m.weight))
          self._linears.append(m)
    return self._linears

  def invalidate_shadows(self):
    for m in self.modules():
      if isinstance(m, HipLinear):
        m.invalidate()

  def refresh_shadows(self):
    """Re-cast every stale bf16 weight shadow in ONE launch (the per-Linear casts are launch-latency bound: 49 launches of
    1 - 6 MB at the 160M size).  Called at the top of every forward; HipLinear.shadow() stays as the lazy fallback."""
    stale = [(m, it) for m in self.linear_modules() for it in (m.stale_item(),) if it is not None]
    if not stale or not stale[0][1][0].is_cuda:
      return
    ops.cast_bf16_t_multi([it for _, it in stale])
    for m, _ in stale:
      m.mark_fresh()

  # ---- forward ---------------------------------------------------------------------
  def _rope(self, device):
    if self._rope_dev is None or self._rope_dev[0].device != device:
      self._rope_dev = tuple(t.to(device) for t in self._rope_host)
    return self._rope_dev

  _mask_status = None  # device flag of the LAST bool-mask conversion (checked one call later: reading it at once would wait for the GPU)

  @classmethod
  def _doc_start(cls, attn_mask, B, T):
    if attn_mask is None:
      return None
    if isinstance(attn_mask, Fn.DocMask):
      return attn_mask
    if attn_mask.dim() == 2 and attn_mask.dtype in (torch.int32, torch.int64):
      return attn_mask.to(torch.int32).contiguous()
    if attn_mask.dim() == 3 and attn_mask.dtype == torch.bool:
      # block-diagonal causal mask of data_prep_utils.py:7-23 (the reference's calling convention, engine/engine.py:21-23,109): first allowed
      # key of every query row, by one small kernel that also CHECKS every row to be exactly [first, i]
      if attn_mask.shape != (B, T, T):
        raise ValueError(f'attn_mask must be [B,T,T]={B, T, T}, got {tuple(attn_mask.shape)}')
      if not attn_mask.is_cuda:  # host logic / CPU tests: the same definition with torch ops
        return attn_mask.to(torch.uint8).argmax(dim=-1).to(torch.int32).contiguous()
      cls.check_mask_status()
      ds, cls._mask_status = ops.doc_start_from_mask(attn_mask)
      return ds
    raise TypeError('attn_mask must be None, a bool [B,T,T] mask or an int32 doc_start [B,T]')

  @classmethod
  def check_mask_status(cls):
    """Raises if the previous bool mask was not a block-diagonal causal mask (rows exactly True on [doc_start, i]): any other mask cannot
    be expressed as doc_start and would have been mis-read.  Called at the next conversion (and by the tests); the flag has long been written."""
    st, cls._mask_status = cls._mask_status, None
    if st is not None and int(st.item()) != 0:
      raise ValueError('attn_mask: a row of the previous bool mask was not exactly True on [first allowed key, query]: only block-diagonal causal '
                       'masks (data_prep_utils.py:7-23) are supported')

  def _trunk(self, x, attn_mask):
    if x.dim() != 2 or x.dtype != torch.int64:
      raise TypeError('inputs must be an int64 [B, T] tensor')
    if not x.is_cuda:
      raise RuntimeError('plainlm_amd.Transformer runs on MI355X only (inputs are on CPU; there is no CPU fallback)')
    B, T = x.shape
    if T > self.cfg.seq_len:
      raise ValueError(f'sequence length {T} exceeds cfg.seq_len {self.cfg.seq_len}')
    rope = self._rope(x.device)
    doc_start = self._doc_start(attn_mask, B, T)
    if doc_start is not None and not isinstance(doc_start, Fn.DocMask):
      doc_start = Fn.DocMask(doc_start, self.cfg.n_heads)  # + the plan: one small launch per batch, shared by every layer's attention launches
    self.refresh_shadows()
    h = self.embed_tokens(x).view(B * T, self.cfg.dim)
    branch = None
    for layer in self.layers:
      h, branch = layer(h, branch, rope, doc_start, B, T)
    _, y = Fn.AddNormFn.apply(h, branch, self.out_norm.weight, self.out_norm)
    return y, B, T

  @torch.compiler.disable  # engine/engine.py:68-70 may wrap the model in torch.compile: the kernels are hand-written, there is nothing to trace -
  def forward(self, x, attn_mask=None):  # the compiled module runs this forward as it is (same bits), instead of graph-breaking on every ctypes call
    """inputs int64 [B,T], attn_mask -> bf16 logits [B,T,V] (models/transformer.py:108-114)."""
    y, B, T = self._trunk(x, attn_mask)
    return self.lm_head(y).view(B, T, self.cfg.vocab_size)

  @torch.compiler.disable
  def loss(self, x, targets, attn_mask=None):
    """Mean token cross-entropy (fp32 scalar) with lm_head + CE fused (never keeps fp32 logits)."""
    y, B, T = self._trunk(x, attn_mask)
    tg = targets.reshape(-1).contiguous()
    return Fn.HeadLossFn.apply(y, self.lm_head.weight, self.lm_head, tg, self.head_chunk_rows)

"""torch.autograd.Function wrappers over the HIP kernels (forward + hand-written backward).

dtype flow = the reference's bf16-autocast GPU flow: fp32 residual
stream and parameters, bf16 activations out of every Linear, fp32 loss, fp32 weight
gradients.  Weight gradients are either returned to autograd (drop-in mode: works
under the reference's own engine / DDP) or written straight into a flat fp32 gradient
buffer through a ``GradSink`` (our engine; enables bucketed RCCL all-reduce that
overlaps the rest of backward).
"""

import torch

from . import ops


class GradSink:
  """Routes parameter gradients into ``param.main_grad`` views of one flat fp32 buffer.

  The first write to a parameter inside an accumulation window overwrites, later
  writes add (covers gradient accumulation and tied embeddings alike).  ``on_ready``
  is invoked after the kernel producing a parameter's gradient has been enqueued."""

  def __init__(self):
    self.enabled = False
    self.written = set()
    self.on_ready = None  # callable(param) or None
    # callable(param) -> True / False / None, asked when a weight gradient has been queued: True = issue the queued group now (the
    # reducer answers True when the queue holds every outstanding gradient of a bucket and enough bytes for an efficient grouped
    # launch), None = no opinion (the count-based default below decides)
    self.on_queued = None
    # Weight gradients of bias-free Linears have no consumer inside backward, so they are queued and issued several at a
    # time as ONE grouped launch (ops.gemm_tn_grouped): whole-K tiles for the full rounds of the persistent grid, split-K only
    # for the remainder.  The more problems per launch the smaller that remainder (one block = 108 tiles is all remainder:
    # 0.47 ms; three blocks 0.41 ms per block; all twelve = 1296 tiles leave 16 to split: 0.39), but the later the gradients
    # exist: with a gradient consumer attached (DDP buckets, `on_ready`) a group is three transformer blocks (85 MB of
    # gradients, about one 64 MiB bucket and a half), without one it is the whole model (flushed by the embedding's backward).
    self.dw_group_local, self.dw_group_ddp = 48, 12
    # The queue keeps each Linear's dy and saved input alive until its launch (autograd would free them node by node): ~0.8 GB per
    # block at 160M / 32 x 1024 tokens, ~10 GB for the whole model.  Bounded by bytes as well as by count: the group is issued
    # early once it pins more than this (PLM_DW_QUEUE_GB; default a quarter of the device's memory, never reached by the BASELINE
    # configs), so deeper models and larger micro-batches do not run out of memory where a group of four blocks would not.
    self.dw_queue_bytes_max = None
    self.dw_queue_bytes = 0
    self.dw_queue = []
    # RMSNorm weight gradients: the backward kernel leaves per-block partial sums; their column sums (25 launch-bound
    # kernels at the 160M size) are queued and run as ONE launch when backward reaches the embedding (flush_dw)
    self.norm_queue = []

  def begin_window(self):
    self.flush_dw()
    self.written.clear()

  def defer_dw(self, dy, x, p):
    """Queue dW(p) (+)= dy^T x; runs when the group is full or at flush_dw()."""
    if isinstance(dy, torch.Tensor) and isinstance(x, torch.Tensor):
      nbytes = dy.numel() * dy.element_size() + x.numel() * x.element_size()
      if self.dw_queue_bytes_max is None:
        self.resolve_queue_budget(dy.device)
      if self.dw_queue and self.dw_queue_bytes + nbytes > self.dw_queue_bytes_max:
        self._flush_linear_dw()  # BEFORE the append: the queue never pins more than the budget
      self.dw_queue_bytes += nbytes
    self.dw_queue.append((dy, x, p, not self.first_write(p)))
    verdict = self.on_queued(p) if self.on_queued is not None else None
    if verdict is None:
      verdict = len(self.dw_queue) >= (self.dw_group_ddp if self.on_ready is not None else self.dw_group_local)
    if verdict or len(self.dw_queue) >= self.dw_group_local:
      self._flush_linear_dw()

  def resolve_queue_budget(self, device, agreed=None):
    """Byte budget of the dW queue, resolved ONCE (not on the backward path): PLM_DW_QUEUE_GB, else a quarter of the device's
    memory; data-parallel runs pass the minimum over the ranks (`agreed`, ddp.agree_min) - ranks that flush at different points
    would launch their gradient buckets in different orders."""
    if agreed is not None:
      self.dw_queue_bytes_max = int(agreed)
      return self.dw_queue_bytes_max
    import os
    gb = os.environ.get('PLM_DW_QUEUE_GB')
    dev = torch.device(device)
    self.dw_queue_bytes_max = (int(float(gb) * 2 ** 30) if gb else
                               torch.cuda.get_device_properties(dev).total_memory // 4 if dev.type == 'cuda' else 1 << 62)
    return self.dw_queue_bytes_max

  def defer_norm_dw(self, part, p):
    self.norm_queue.append((part, p, not self.first_write(p)))

  def flush_norms(self):
    q, self.norm_queue = self.norm_queue, []
    if not q:
      return
    by_shape = {}
    for part, p, acc in q:
      by_shape.setdefault(tuple(part.shape), []).append((part, p.main_grad, acc))
    for items in by_shape.values():
      ops.colsum_multi(items)
    for _, p, _ in q:
      self.ready(p)

  def flush_dw(self):
    """Everything still queued (end of backward: called when the embedding's gradient is due, and by the optimizer)."""
    self._flush_linear_dw()
    self.flush_norms()

  def _flush_linear_dw(self):
    q, self.dw_queue = self.dw_queue, []
    self.dw_queue_bytes = 0
    if not q:
      return
    if len(q) == 1 or not ops.gemm_tn_grouped([(dy, x, p.main_grad, acc, None) for dy, x, p, acc in q]):
      for dy, x, p, acc in q:
        ops.gemm_tn(dy, x, out=p.main_grad, accumulate=acc)
    for _, _, p, _ in q:
      self.ready(p)

  def active_for(self, p):
    return self.enabled and getattr(p, 'main_grad', None) is not None

  def first_write(self, p):
    first = id(p) not in self.written
    self.written.add(id(p))
    return first

  def ready(self, p):
    if self.on_ready is not None:
      self.on_ready(p)


class LinearFn(torch.autograd.Function):
  """y = x W^T (nn.Linear, bias-free: transformer.py:36-37,97; components.py:50-51)."""

  @staticmethod
  def forward(ctx, x, weight, lin):
    wb, _ = lin.shadow()
    ctx.save_for_backward(x)
    ctx.lin = lin
    return ops.gemm_nt(x, wb)

  @staticmethod
  def backward(ctx, dy):
    (x,) = ctx.saved_tensors
    lin = ctx.lin
    dy = dy.contiguous()
    _, wbt = lin.shadow()
    dx = ops.gemm_nt(dy, wbt[:, :lin.out_features]) if ctx.needs_input_grad[0] else None
    dw = None
    if ctx.needs_input_grad[1]:
      sink, p = lin.sink, lin.weight
      if sink is not None and sink.active_for(p):
        sink.defer_dw(dy, x, p)
      else:
        dw = ops.gemm_tn(dy, x)
    return dx, dw, None


class SwiGLUMLPFn(torch.autograd.Function):
  """y = fc2(silu(gate) * up) with (gate | up) = x W_fc1^T (components.py:50-57): the whole MLP as one autograd node, so that both
  activations live in GEMM epilogues.  Forward: fc1 with the SwiGLU gate in its epilogue (ops.fc1_swiglu), then fc2.  Backward: the
  dX GEMM of fc2 with the SwiGLU backward in ITS epilogue (ops.fc2_dx_swiglu_bwd: d(act) never reaches memory), then the plain
  Linear backward of fc1.  Every intermediate carries the bits of the unfused chain (Linear -> SwiGLU -> Linear)."""

  @staticmethod
  def forward(ctx, x, w1, w2, fc1, fc2):
    w1b, _ = fc1.shadow()
    w2b, _ = fc2.shadow()
    u, act = ops.fc1_swiglu(x, w1b)
    ctx.save_for_backward(x, u, act)
    ctx.fc1, ctx.fc2 = fc1, fc2
    return ops.gemm_nt(act, w2b)

  @staticmethod
  def backward(ctx, dy):
    x, u, act = ctx.saved_tensors
    fc1, fc2 = ctx.fc1, ctx.fc2
    dy = dy.contiguous()
    _, w1t = fc1.shadow()
    _, w2t = fc2.shadow()
    du = ops.fc2_dx_swiglu_bwd(dy, w2t[:, :fc2.out_features], u)
    dw2 = dw1 = None
    if ctx.needs_input_grad[2]:
      sink, p = fc2.sink, fc2.weight
      if sink is not None and sink.active_for(p):
        sink.defer_dw(dy, act, p)
      else:
        dw2 = ops.gemm_tn(dy, act)
    dx = ops.gemm_nt(du, w1t[:, :fc1.out_features]) if ctx.needs_input_grad[0] else None
    if ctx.needs_input_grad[1]:
      sink, p = fc1.sink, fc1.weight
      if sink is not None and sink.active_for(p):
        sink.defer_dw(du, x, p)
      else:
        dw1 = ops.gemm_tn(du, x)
    return dx, dw1, dw2, None, None


class PlainMLPFn(torch.autograd.Function):
  """y = fc2(act(fc1 x)) for the reference's plain MLP classes (components.py:31-40 `MLP`: silu; :59-70 `MLPReluSquared`: relu squared)
  as one autograd node: two NT GEMMs around a stand-alone activation kernel (these classes are off every shipped config's path, so no
  fused epilogue is built for them), backward = dX of fc2 -> activation backward -> the plain Linear backward of fc1."""

  @staticmethod
  def forward(ctx, x, w1, w2, fc1, fc2, kind):
    w1b, _ = fc1.shadow()
    w2b, _ = fc2.shadow()
    u = ops.gemm_nt(x, w1b)
    act = ops.act_fwd(u, kind)
    ctx.save_for_backward(x, u, act)
    ctx.fc1, ctx.fc2, ctx.kind = fc1, fc2, kind
    return ops.gemm_nt(act, w2b)

  @staticmethod
  def backward(ctx, dy):
    x, u, act = ctx.saved_tensors
    fc1, fc2 = ctx.fc1, ctx.fc2
    dy = dy.contiguous()
    _, w1t = fc1.shadow()
    _, w2t = fc2.shadow()
    du = ops.act_bwd(ops.gemm_nt(dy, w2t[:, :fc2.out_features]), u, ctx.kind)
    dw2 = dw1 = None
    if ctx.needs_input_grad[2]:
      sink, p = fc2.sink, fc2.weight
      if sink is not None and sink.active_for(p):
        sink.defer_dw(dy, act, p)
      else:
        dw2 = ops.gemm_tn(dy, act)
    dx = ops.gemm_nt(du, w1t[:, :fc1.out_features]) if ctx.needs_input_grad[0] else None
    if ctx.needs_input_grad[1]:
      sink, p = fc1.sink, fc1.weight
      if sink is not None and sink.active_for(p):
        sink.defer_dw(du, x, p)
      else:
        dw1 = ops.gemm_tn(du, x)
    return dx, dw1, dw2, None, None, None


class QKVRopeFn(torch.autograd.Function):
  """w_qkv projection + RoPE (transformer.py:42-47): returns the projection with q | k ALREADY rotated.
  Contract with AttnFn: AttnFn.backward returns the gradient w.r.t. the UN-rotated projection (the inverse
  rotation is folded into the attention backward epilogues), so this backward is the plain Linear backward."""

  @staticmethod
  def forward(ctx, x, weight, lin, cos, sin, B, T, nh):
    wb, _ = lin.shadow()
    ctx.save_for_backward(x)
    ctx.lin = lin
    return ops.qkv_rope(x, wb, cos, sin, B, T, nh)

  @staticmethod
  def backward(ctx, dy):
    dx, dw, _ = LinearFn.backward(ctx, dy)
    return dx, dw, None, None, None, None, None, None


class EmbedFn(torch.autograd.Function):
  """nn.Embedding (transformer.py:94,110): fp32 row gather; backward = sort-based deterministic segment sum (atomic
  scatter-add only for shapes beyond 65536 tokens / ids)."""

  @staticmethod
  def forward(ctx, ids, weight, emb):
    ctx.save_for_backward(ids)
    ctx.emb = emb
    ctx.wshape = weight.shape
    return ops.embed_fwd(ids, weight)

  @staticmethod
  def backward(ctx, g):
    (ids,) = ctx.saved_tensors
    emb = ctx.emb
    g = g.contiguous()
    sink, p = emb.sink, emb.weight
    if sink is not None:
      sink.flush_dw()  # the embedding is the first op of forward = the last node of backward: nothing is left queued after it
    if sink is not None and sink.active_for(p):
      first = sink.first_write(p)
      if not ops.embed_bwd_sorted(ids, g, p.main_grad, accumulate=not first):  # sort-based, no atomics, writes every row
        if first:
          p.main_grad.zero_()
        ops.embed_bwd(ids, g, p.main_grad)
      sink.ready(p)
      return None, None, None
    dw = torch.empty(ctx.wshape, dtype=torch.float32, device=g.device)
    if not ops.embed_bwd_sorted(ids, g, dw, accumulate=False):
      dw.zero_()
      ops.embed_bwd(ids, g, dw)
    return None, dw, None


def _norm_dw(norm, dy, x, w, rstd, gin, want_bf16):
  """Shared backward of the two norm Functions; returns (dx, dx_bf16, dw_for_autograd)."""
  sink, p = norm.sink, norm.weight
  if sink is not None and sink.active_for(p):
    dx, dxb, part = ops.rmsnorm_bwd(dy, x, w, rstd, gin=gin, want_bf16=want_bf16, defer_dw=True)
    sink.defer_norm_dw(part, p)
    return dx, dxb, None
  return ops.rmsnorm_bwd(dy, x, w, rstd, gin=gin, want_bf16=want_bf16)


class NormFn(torch.autograd.Function):
  """(x, y) = (x, bf16(RMSNorm(x) * w)) for the fp32 x of the first block (the embedding output), which feeds the norm AND the residual
  stream.  x is handed back as a second output (an alias, no copy) so that both of its consumers' gradients arrive HERE: the residual
  path's gradient goes into the norm-backward kernel as its `gin` term - autograd would otherwise sum the two [M, d] fp32 gradients
  with an elementwise add of its own (the one non-plm kernel round 2's step trace still showed).

  Constraint: the first output is a VIEW of an input created inside a custom Function, so autograd refuses any later in-place
  operation on it (`x += ...` on block 0's residual stream raises).  The blocks never do that: every residual add is AddNormFn, which
  writes a new tensor.  tests/test_model_gpu.py::test_normfn_gradient_paths covers the three gradient cases."""

  @staticmethod
  def forward(ctx, x, weight, norm):
    _, y, rstd = ops.rmsnorm_fwd(x, weight, norm.eps)
    ctx.save_for_backward(x, weight, rstd)
    ctx.norm = norm
    ctx.set_materialize_grads(False)
    return x.view_as(x), y

  @staticmethod
  def backward(ctx, g_x, g_y):
    x, w, rstd = ctx.saved_tensors
    if g_y is None:
      return g_x, None, None
    gin = g_x.contiguous() if g_x is not None else None
    dx, _, dw = _norm_dw(ctx.norm, g_y.contiguous(), x, w, rstd, gin, False)
    return dx, dw, None


class AddNormFn(torch.autograd.Function):
  """(x_new, y) = (x + branch, bf16(RMSNorm(x + branch) * w)): the residual add of
  transformer.py:81-82 fused with the following RMSNorm (components.py:22-28)."""

  @staticmethod
  def forward(ctx, x, branch, weight, norm):
    xout, y, rstd = ops.rmsnorm_fwd(x, weight, norm.eps, branch=branch)
    ctx.save_for_backward(xout, weight, rstd)
    ctx.norm = norm
    ctx.set_materialize_grads(False)
    return xout, y

  @staticmethod
  def backward(ctx, g_xout, g_y):
    xout, w, rstd = ctx.saved_tensors
    if g_y is None:  # y unused: pure residual add
      gb = g_xout.to(torch.bfloat16) if g_xout is not None else None
      return g_xout, gb, None, None
    gin = g_xout.contiguous() if g_xout is not None else None
    dx, dxb, dw = _norm_dw(ctx.norm, g_y.contiguous(), xout, w, rstd, gin, True)
    return dx, dxb, dw, None


class SwiGLUFn(torch.autograd.Function):
  """silu(u[:, :h]) * u[:, h:] (components.py:55-56) in the reference's bf16 rounding order."""

  @staticmethod
  def forward(ctx, u):
    ctx.save_for_backward(u)
    return ops.swiglu_fwd(u)

  @staticmethod
  def backward(ctx, dout):
    (u,) = ctx.saved_tensors
    return ops.swiglu_bwd(dout.contiguous(), u)


class DocMask:
  """A document mask as the kernels take it: ``doc_start`` int32 [B,T] (query i sees key j iff doc_start[i] <= j <= i) and its
  plan (ops.attn_doc_plan: doc_end[B,T] + the cost-sorted tile lists), built once per batch and shared by all layers."""
  __slots__ = ('doc_start', 'plan', 'n_heads')

  def __init__(self, doc_start, n_heads, plan=None):
    self.doc_start = doc_start
    self.n_heads = n_heads
    self.plan = ops.attn_doc_plan(doc_start, n_heads) if plan is None else plan


class AttnFn(torch.autograd.Function):
  """RoPE + causal/doc-masked SDPA on the raw w_qkv output (transformer.py:43-65).  ``doc``: None (causal), a DocMask, or a bare
  int32 doc_start [B,T] (its plan is then built per call)."""

  @staticmethod
  def forward(ctx, qkv, cos, sin, doc, B, T, nh):
    # qkv arrives with q, k already rotated (QKVRopeFn); attn_bwd returns the gradient w.r.t. the UN-rotated projection.
    if doc is not None and (not isinstance(doc, DocMask) or doc.n_heads != nh):
      doc = DocMask(doc.doc_start if isinstance(doc, DocMask) else doc, nh)
    ds, plan = (doc.doc_start, doc.plan) if doc is not None else (None, None)
    out, lse = ops.attn_fwd(qkv, B, T, nh, ds, plan)
    ctx.save_for_backward(qkv, out, lse, cos, sin)
    ctx.doc = doc
    ctx.dims = (B, T, nh)
    return out

  @staticmethod
  def backward(ctx, dout):
    qkv, out, lse, cos, sin = ctx.saved_tensors
    B, T, nh = ctx.dims
    ds, plan = (ctx.doc.doc_start, ctx.doc.plan) if ctx.doc is not None else (None, None)
    dqkv = ops.attn_bwd(qkv, out, dout.contiguous(), lse, cos, sin, B, T, nh, ds, plan)
    return dqkv, None, None, None, None, None, None


class HeadLossFn(torch.autograd.Function):
  """lm_head + CrossEntropyLoss (transformer.py:114 + engine.py:111) with the logits buffer
  turned into dlogits in place; returns the mean token loss (fp32 scalar).

  chunk_rows > 0: the [M, V] logits are never materialised.  Forward walks the token rows in chunks:
  logits chunk -> cross-entropy in place (-> dlogits chunk) -> its dX rows and its contribution to dW, all while the
  chunk is hot; only the un-scaled dX [M, d] and a private fp32 dW [V, d] survive forward, and backward multiplies both
  by the upstream gradient (3.3 GB -> chunk_rows/M of it at the 160M bench shape)."""

  @staticmethod
  def forward(ctx, y, weight, lin, targets, chunk_rows=0):
    wb, wbt = lin.shadow()
    M, V = y.shape[0], lin.out_features
    ctx.lin = lin
    ctx.chunked = bool(chunk_rows) and 0 < chunk_rows < M
    if not ctx.chunked:
      # rows padded to a multiple of 64 columns: 128-byte aligned rows, and dlogits feeds the dX GEMM with K % 64 == 0
      buf = torch.empty((M, lin.out_pad), dtype=torch.bfloat16, device=y.device)
      ops.gemm_nt(y, wb, out=buf[:, :V])
      rows = ops.ce_fwd_bwd_(buf, targets, 1.0 / M, V=V)
      ctx.save_for_backward(y, buf)
      return ops.mean(rows)
    buf = torch.empty((chunk_rows, lin.out_pad), dtype=torch.bfloat16, device=y.device)
    need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]  # False under no_grad (evaluation): loss only
    dy = torch.empty_like(y) if need else None
    dw = torch.empty((V, lin.in_features), dtype=torch.float32, device=y.device) if need else None
    rows = []
    for c0 in range(0, M, chunk_rows):
      n = min(chunk_rows, M - c0)
      b, yc = buf[:n], y[c0:c0 + n]
      ops.gemm_nt(yc, wb, out=b[:, :V])
      rows.append(ops.ce_fwd_bwd_(b, targets[c0:c0 + n], 1.0 / M, V=V))
      if need:
        ops.gemm_nt(b, wbt, out=dy[c0:c0 + n])
        ops.gemm_tn(b[:, :V], yc, out=dw, accumulate=c0 > 0)
    if need:
      ctx.save_for_backward(dy, dw)
    return ops.mean(torch.cat(rows))

  @staticmethod
  def backward(ctx, g):
    lin = ctx.lin
    alpha = g.to(torch.float32).contiguous()
    sink, p = lin.sink, lin.weight
    to_sink = sink is not None and sink.active_for(p)
    if ctx.chunked:
      dy_un, dw_un = ctx.saved_tensors
      dy = ops.scale_bf16_(dy_un, alpha) if ctx.needs_input_grad[0] else None  # the saved buffer is ours: scaled in place
      dw = None
      if ctx.needs_input_grad[1]:
        if to_sink:
          ops.axpy_f32_(p.main_grad, dw_un, alpha, accumulate=not sink.first_write(p))
          sink.ready(p)
        else:
          dw = ops.axpy_f32_(torch.empty_like(dw_un), dw_un, alpha, accumulate=False)
      return dy, dw, None, None, None
    y, dbuf = ctx.saved_tensors
    _, wbt = lin.shadow()
    dlogits = dbuf[:, :lin.out_features]
    dy = ops.gemm_nt(dbuf, wbt, alpha=alpha) if ctx.needs_input_grad[0] else None  # K = out_pad, pads are zero on both sides
    dw = None
    if ctx.needs_input_grad[1]:
      if to_sink:
        ops.gemm_tn(dlogits, y, out=p.main_grad, accumulate=not sink.first_write(p), alpha=alpha)
        sink.ready(p)
      else:
        dw = ops.gemm_tn(dlogits, y, alpha=alpha)
    return dy, dw, None, None, None

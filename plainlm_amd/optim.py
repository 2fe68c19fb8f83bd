"""Optimizer tail on the gfx950 kernels (engine/engine.py:126-135, optim/init_optim.py:14-21).

``FlatAdamW`` is a ``torch.optim.AdamW`` subclass, so ``engine.optimizer`` keeps the reference's interface
(``param_groups`` with per-group ``lr`` written by the LR schedule, ``state_dict()`` / ``load_state_dict()`` in
torch's layout: per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), but

  * parameters, gradients and both moments live in flat fp32 buffers (one span per weight-decay group),
  * ``clip_and_step(max_norm)`` = one deterministic ||g||^2 reduction + one fused AdamW launch per group; the
    clip coefficient min(1, max_norm / (||g|| + 1e-6)) is computed on the device and folded into the AdamW
    kernel, so clipping costs no extra pass over the gradients and no host synchronisation.

Arithmetic matches ``torch.optim.AdamW`` (decoupled decay ``p *= 1 - lr*wd``; bias-corrected moments;
``denom = sqrt(v)/sqrt(bc2) + eps``) and ``torch.nn.utils.clip_grad_norm_``.
"""

import torch

from . import ops


class FlatAdamW(torch.optim.AdamW):
  def __init__(self, model, param_groups, lr, betas, eps, weight_decay):
    super().__init__(param_groups, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, fused=False, foreach=False)
    if getattr(model, '_flat_grad', None) is None:
      raise RuntimeError('FlatAdamW needs model.enable_main_grad() first (flat gradient buffer)')
    self.model = model
    dev = model._flat_grad.device
    # Re-lay parameters and gradients group by group so that every group is ONE contiguous span.
    order = [p for g in self.param_groups for p in g['params']]
    if len({id(p) for p in order}) != len(list(model.parameters())):
      raise ValueError('param_groups must cover every model parameter exactly once')
    total = sum(p.numel() for p in order)
    self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
    self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
    self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
    self.flat_g = model._flat_grad
    if self.flat_g.numel() != total:
      raise ValueError('flat gradient buffer does not match the parameter groups')
    self.group_spans = [None] * len(self.param_groups)
    self._views = {}
    off = 0
    spans_by_param = {}
    # Placement: zero-weight-decay groups (the norm weights, 77 KB at 160M) FIRST, so that the flat buffer reads
    # [norms | embed_tokens | layer 0 ... | lm_head]: ddp.plan_buckets walks it from the end (= the order gradients become
    # ready) and the norm weights - complete only when layer 0 has been differentiated - share the LAST bucket with
    # embed_tokens instead of holding up lm_head's.  param_groups keeps the reference's order (decay, no-decay): only the
    # placement changes, optimizer.state_dict() does not.
    placement = sorted(range(len(self.param_groups)), key=lambda i: (self.param_groups[i]['weight_decay'] != 0.0, i))
    for gi in placement:
      g = self.param_groups[gi]
      lo = off
      for p in g['params']:
        n = p.numel()
        self.flat_p[off:off + n].copy_(p.data.reshape(-1))
        p.data = self.flat_p[off:off + n].view(p.shape)
        p.main_grad = self.flat_g[off:off + n].view(p.shape)
        spans_by_param[id(p)] = (off, n)
        self._views[id(p)] = (self.flat_m[off:off + n].view(p.shape), self.flat_v[off:off + n].view(p.shape))
        off += n
      self.group_spans[gi] = (lo, off)
    # the model's span table (used by the gradient reducer) follows parameters() order
    model._grad_spans = [spans_by_param[id(p)] for p in model.parameters()]
    model.invalidate_shadows()
    self._scratch = torch.empty(4096, dtype=torch.float32, device=dev)
    self._step_count = 0
    self.last_grad_norm = None
    self._publish_state()
    # SURVEY section 8f N1, second half: the Linear weights' update also writes their bf16 shadows (W and W^T), so the training step has
    # no stand-alone weight cast (what autocast does per forward, engine/engine.py:75).  Per weight-decay group: the Linear weights
    # of the group (one fused launch) + the rest of its span (embed_tokens / the norm weights: flat kernel).  PLM_ADAMW_SHADOWS=0
    # restores the flat kernel for everything + invalidated shadows.
    import os
    self.emits_shadows = os.environ.get('PLM_ADAMW_SHADOWS', '1') != '0' and hasattr(model, 'linear_modules')
    self._fused = [None] * len(self.param_groups)  # per group: (linear modules, item tensors, ctypes table or None, leftover [(lo, hi)])
    if self.emits_shadows:
      for gi, g in enumerate(self.param_groups):
        ids = {id(p) for p in g['params']}
        lins = [m for m in model.linear_modules() if id(m.weight) in ids]
        if not lins:
          continue
        items, taken = [], []
        for lin in lins:
          w = lin.weight
          lin.stale_item()  # allocates the shadow buffers
          mview, vview = self._views[id(w)]
          items.append((w.data, w.main_grad, mview, vview, lin._shadow[0], lin._shadow[1]))
          taken.append(spans_by_param[id(w)])
        lo, hi = self.group_spans[gi]
        rest, cur = [], lo
        for o, n in sorted(taken):
          if o > cur:
            rest.append((cur, o))
          cur = o + n
        if cur < hi:
          rest.append((cur, hi))
        self._fused[gi] = (lins, items, None, rest)

  def _publish_state(self):
    """torch-layout per-parameter state backed by views of the flat moment buffers."""
    for g in self.param_groups:
      for p in g['params']:
        m, v = self._views[id(p)]
        self.state[p] = {'step': torch.tensor(float(self._step_count)), 'exp_avg': m, 'exp_avg_sq': v}

  @torch.no_grad()
  def clip_and_step(self, max_norm=None):
    if getattr(self.model, 'sink', None) is not None:
      self.model.sink.flush_dw()  # queued weight-gradient GEMMs (functional.GradSink) must have been issued
    self._step_count += 1
    clip = None
    if max_norm:
      sq = ops.sumsq(self.flat_g, self._scratch)
      self.last_grad_norm = torch.sqrt(sq)
      clip = torch.clamp(float(max_norm) / (self.last_grad_norm + 1e-6), max=1.0).reshape(1).contiguous()
    fresh = []
    for gi, (g, (lo, hi)) in enumerate(zip(self.param_groups, self.group_spans)):
      if hi == lo:
        continue
      b1, b2 = g['betas']
      spans = [(lo, hi)]
      if self._fused[gi] is not None:
        lins, items, table, spans = self._fused[gi]
        if any(lin._shadow[0] is not it[4] or lin.weight.data_ptr() != it[0].data_ptr() for lin, it in zip(lins, items)):
          # a shadow buffer or a weight was re-allocated behind our back (model moved, weights re-laid): rebuild the cached table
          items = [(lin.weight.data, lin.weight.main_grad) + self._views[id(lin.weight)] + (lin._shadow[0], lin._shadow[1]) for lin in lins]
          table = None
        table = ops.adamw_cast_multi_(items, float(g['lr']), b1, b2, g['eps'], g['weight_decay'], self._step_count, clip, table)
        self._fused[gi] = (lins, items, table, spans)
        fresh.extend(lins)
      for a, b in spans:
        ops.adamw_(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], float(g['lr']), b1, b2,
                   g['eps'], g['weight_decay'], self._step_count, clip)
    for st in self.state.values():
      st['step'].fill_(float(self._step_count))
    self.model.invalidate_shadows()  # raw-pointer update: torch's version counters did not move
    for lin in fresh:                # ... except where this step has just written the shadows itself
      lin.mark_fresh()

  @torch.no_grad()
  def step(self, closure=None):
    if closure is not None:
      raise NotImplementedError('FlatAdamW.step does not take a closure')
    self.clip_and_step(None)

  def zero_grad(self, set_to_none=True):
    # gradients live in the flat buffer and are overwritten by the first write of the next window
    for g in self.param_groups:
      for p in g['params']:
        p.grad = None

  def load_state_dict(self, state_dict):
    super().load_state_dict(state_dict)
    steps = []
    for g in self.param_groups:
      for p in g['params']:
        st = self.state.get(p)
        if st:
          m, v = self._views[id(p)]
          m.copy_(st['exp_avg'])
          v.copy_(st['exp_avg_sq'])
          steps.append(int(float(st['step'])))
    self._step_count = max(steps) if steps else 0
    self._publish_state()

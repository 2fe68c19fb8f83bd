"""Training engine with the reference's ``TorchEngine`` interface (engine/engine.py:37-177)
on top of the gfx950 kernels: ``HipEngine(model, cfg, device, local_rank, ckpt)``,
``.step(batch) -> 0-d loss``, ``.eval(loader) -> float``, attributes ``.optimizer``,
``.scheduler``, ``.scaler``, ``.model``.

Differences that stay inside the contract:
  * forward uses the fused ``model.loss`` (lm_head + cross-entropy) instead of materialising
    fp32 logits; the returned loss is the same un-divided micro-batch mean;
  * torch DDP is replaced by ``ddp.GradReducer`` (bucketed RCCL all-reduce on a side stream,
    skipped on non-final accumulation micro-steps exactly like ``require_backward_grad_sync``);
  * document masks travel as ``doc_start[B,T]`` (prefix sums of ``docs_lengths``) instead of a
    [B,T,T] boolean tensor;
  * ``eval`` implements the evident intent (sum of losses / number of batches across ranks);
    the reference's version raises AttributeError at engine.py:175.
"""

import math

import numpy as np
import torch
import torch.distributed as dist

from .construct import get_param_groups
from . import ddp


# ---- LR schedules (optim/lr_schedule.py:29-132), host-side scalar math --------------------
# (state_dict() keeps the reference's attribute names: scheduler states travel in checkpoints, checkpoint_utils.py:22-24)
class _Schedule:
  def __init__(self, optimizer):
    self.optimizer = optimizer

  def set_optim_lr(self, lr):
    for group in self.optimizer.param_groups:
      group['lr'] = lr

  def state_dict(self):
    return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

  def load_state_dict(self, state):
    self.__dict__.update(state)

  def step(self):
    self.iter += 1
    self.set_optim_lr(self.get_lr(self.iter))


class WarmupCosine(_Schedule):
  def __init__(self, optimizer, lr_start, lr_max, lr_end, warmup_steps, T):
    super().__init__(optimizer)
    self.lr_start, self.lr_max, self.lr_end = lr_start, lr_max, lr_end
    self.warmup_steps, self.T, self.iter = warmup_steps, T, 0
    self.set_optim_lr(lr_start)

  def get_lr(self, t):
    if t <= self.warmup_steps:
      return self.lr_start + (self.lr_max - self.lr_start) / self.warmup_steps * t
    if t <= self.T:
      prog = (t - self.warmup_steps) / (self.T - self.warmup_steps)
      return self.lr_end + 0.5 * (self.lr_max - self.lr_end) * (1 + math.cos(math.pi * prog))
    return self.lr_end


class WarmupConstant(_Schedule):
  def __init__(self, optimizer, lr_start, lr_max, warmup_steps):
    super().__init__(optimizer)
    self.lr_start, self.lr_max, self.warmup_steps, self.iter = lr_start, lr_max, warmup_steps, 0
    self.set_optim_lr(lr_start)

  def get_lr(self, t):
    if t <= self.warmup_steps:
      return self.lr_start + (self.lr_max - self.lr_start) / self.warmup_steps * t
    return self.lr_max


def _ramp(t, t0, n, y0, y1):
  """the straight line through (t0, y0) and (t0 + n, y1), at t"""
  return y0 + (y1 - y0) / n * (t - t0)


class WSD(_Schedule):
  """Trapezoid (optim/lr_schedule.py:57-82): linear warm-up, a plateau at lr_max up to `cooldown_start_step`, then a straight line that
  reaches lr_end `cooldown_steps` later (and keeps its slope beyond, like the reference's)."""

  def __init__(self, optimizer, lr_start, lr_max, lr_end, warmup_steps, cooldown_start_step, cooldown_steps):
    super().__init__(optimizer)
    self.lr_start, self.lr_max, self.lr_end = lr_start, lr_max, lr_end
    self.warmup_steps, self.cooldown_start_step, self.cooldown_steps, self.iter = warmup_steps, cooldown_start_step, cooldown_steps, 0
    self.set_optim_lr(lr_start)

  def get_lr(self, t):
    if t <= self.warmup_steps:
      return _ramp(t, 0, self.warmup_steps, self.lr_start, self.lr_max)
    if t <= self.cooldown_start_step:
      return self.lr_max
    return _ramp(t, self.cooldown_start_step, self.cooldown_steps, self.lr_max, self.lr_end)


class LinearCooldown(_Schedule):
  """The cool-down leg alone, for a run resumed at `cooldown_start_step` (optim/lr_schedule.py:108-132).  Like the reference's it leaves the
  optimizer's learning rate alone until its first step, and takes ONLY `iter` from a saved state (the other fields come from the new config)."""

  def __init__(self, optimizer, lr_max, lr_end, cooldown_start_step, cooldown_steps):
    super().__init__(optimizer)
    self.lr_max, self.lr_end, self.cooldown_start_step, self.cooldown_steps, self.iter = lr_max, lr_end, cooldown_start_step, cooldown_steps, 0

  def get_lr(self, t):
    if t <= self.cooldown_start_step:
      return self.lr_max
    return _ramp(t, self.cooldown_start_step, self.cooldown_steps, self.lr_max, self.lr_end)

  def load_state_dict(self, state):
    self.iter = state.get('iter', 0)


def _steps(value, budget):
  return value if isinstance(value, int) else int(value * budget)


def initialize_scheduler(optimizer, cfg):
  """optim/init_optim.py:73-137: the reference's four schedulers."""
  name = getattr(cfg, 'scheduler', None)
  if name is None:
    return None
  warmup = _steps(cfg.warmup_steps, cfg.steps_budget) if getattr(cfg, 'warmup_steps', None) is not None else 0
  lr_end = None
  if getattr(cfg, 'lr_end', None) is not None or getattr(cfg, 'lr_end_pct', None) is not None:
    lr_end = cfg.lr_end if cfg.lr_end is not None else cfg.lr_end_pct * cfg.lr
  if name == 'warmup_cosine':
    return WarmupCosine(optimizer, cfg.lr_start, cfg.lr, lr_end, warmup, cfg.steps_budget)
  if name == 'warmup_constant':
    return WarmupConstant(optimizer, cfg.lr_start, cfg.lr, warmup)
  cooldown = _steps(cfg.cooldown_steps, cfg.steps_budget) if getattr(cfg, 'cooldown_steps', None) is not None else None
  if name == 'wsd':
    return WSD(optimizer, cfg.lr_start, cfg.lr, lr_end, warmup, cfg.steps_budget - cooldown, cooldown)
  if name == 'linear_cooldown':
    return LinearCooldown(optimizer, cfg.lr, lr_end, cfg.resume_step, cooldown)
  raise NotImplementedError(f'Not implemented scheduler: {name}.')


def intialize_optimizer(param_groups, cfg, model=None):
  """(sic) optim/init_optim.py:7-21 — AdamW is the only optimizer on the shipped configs' path.
  ``fused_optim: True`` (the shipped configs) selects FlatAdamW: clip + AdamW on flat buffers with our kernels;
  ``False`` keeps torch.optim.AdamW on the per-parameter views."""
  if cfg.optim != 'adamw':
    raise NotImplementedError(f'Not implemented optim: {cfg.optim}.')
  if bool(getattr(cfg, 'fused_optim', True)) and model is not None:
    from .optim import FlatAdamW
    return FlatAdamW(model, param_groups, lr=cfg.lr, betas=[cfg.beta1, cfg.beta2], eps=getattr(cfg, 'eps', 1e-8),
                     weight_decay=cfg.weight_decay)
  return torch.optim.AdamW(param_groups, lr=cfg.lr, betas=[cfg.beta1, cfg.beta2], weight_decay=cfg.weight_decay,
                           eps=getattr(cfg, 'eps', 1e-8))


def doc_start_from_lengths(docs_lengths, seq_len):
  """Host prefix sums: for every token the index of its document's first token, int32 [B, T].
  Equivalent to the mask of data_prep_utils.py:7-23 cropped at engine.py:23 (lengths sum to T+1)."""
  out = np.empty((len(docs_lengths), seq_len), dtype=np.int32)
  for r, lens in enumerate(docs_lengths):
    lens = np.asarray([int(v) for v in lens], dtype=np.int64)
    if int(lens.sum()) != seq_len + 1:
      raise ValueError('Sum of doc_boundaries does not match max_seq_length.')
    starts = np.cumsum(lens) - lens                      # first token of every document
    out[r] = np.repeat(starts, lens)[:seq_len]           # one vectorised expansion per row (was a Python list of T ints per row)
  return torch.from_numpy(out)


class _Stager:
  """Host -> device staging for `_move_to_device`: the token block travels ONCE ([B, T+1] int64, sliced into
  inputs / targets on the device) through a small ring of persistent pinned buffers, asynchronously on the compute
  stream.  The reference pins two fresh tensors per micro-step (engine.py:29-30)."""

  def __init__(self, depth=4):
    self.bufs, self.depth, self.k = {}, depth, 0

  def to_device(self, t, device):
    if t.is_cuda:
      return t
    key = (tuple(t.shape), t.dtype)
    ring = self.bufs.setdefault(key, [])
    if len(ring) < self.depth:
      ring.append((torch.empty(t.shape, dtype=t.dtype).pin_memory(), torch.cuda.Event()))
    buf, ev = ring[self.k % len(ring)]
    self.k += 1
    ev.synchronize()  # the copy that last used this pinned buffer has finished (it was issued `depth` transfers ago)
    # a plain memcpy: Tensor.copy_ on the host enters an OpenMP region, and libgomp's whole team (one thread per visible CPU)
    # then spins for milliseconds after it - every micro-step.  On a host whose cgroup quota is smaller than its CPU count that
    # burns the quota and the process is throttled for tens of ms at a time (measured: 12 busy cores, 34-37 vs 32.4 ms per step).
    np.copyto(buf.numpy(), t.numpy())
    out = buf.to(device, non_blocking=True)
    ev.record()
    return out


def _move_to_device(batch, seq_len, device, intra_doc_masking, stager=None):
  """engine/engine.py:13-34 with doc_start instead of the [B,T,T] mask."""
  ids = batch['input_ids']
  doc_start = doc_start_from_lengths(batch['docs_lengths'], seq_len) if intra_doc_masking else None
  if stager is None:
    stager = _Stager(depth=1)
  ids = stager.to_device(ids[:, :seq_len + 1].contiguous(), device)
  inputs = ids[:, :seq_len].contiguous()
  targets = ids[:, 1:seq_len + 1].contiguous()
  if doc_start is not None:
    doc_start = stager.to_device(doc_start, device)
  return inputs, targets, doc_start


class HipEngine(torch.nn.Module):
  def __init__(self, model, cfg, device, local_rank=None, ckpt=None, comm_backend=None, bucket_cap_mb=64):
    super().__init__()
    self.micro_steps = 0
    self.accumulated_samples = 0
    self.seq_len = cfg.seq_len
    self.accumulation_steps = cfg.grad_accumulation_steps
    self.grad_clip = cfg.grad_clip
    self.dtype = cfg.dtype
    self.intra_doc_masking = getattr(cfg, 'intra_doc_masking', False)
    self.device = device
    self._stager = _Stager()
    # 'Train loss is nan' (engine.py:116-117).  The reference reads the loss on the host before it enqueues backward: a wait for
    # the whole forward with an empty launch queue behind it, every micro-step - measured +5.5 ms on a 34 ms step
    # (tools/engine_bench.py --nan-check-lag 0).  Default nan_check_lag = 1:
    # the flag of micro-step k (one pinned byte with its own event) is read AFTER k's backward has been enqueued - at the window's
    # last micro-step, before clip + AdamW, or at the submission of micro-step k + 1 - by which time the forward has long
    # finished.  The error is raised from the same step() call when k ends a window (always, with accumulation 1) and from the
    # next call otherwise; an update computed from a NaN loss is never applied, no micro-step goes unchecked (eval() and
    # check_losses() drain).  nan_check_lag = 0 restores the reference's order exactly; larger values defer further inside a window.
    self.nan_check_lag = int(getattr(cfg, 'nan_check_lag', 1))
    # cfg.gc_freeze (default True since round 6; False opts out): after the first optimizer step, collect once and move every object alive at
    # that point (model, optimizer, autograd function classes, everything `import torch` created) out of the cyclic garbage collector's
    # generations (gc.freeze: the collector stays on).  Python's full collections otherwise walk all of them every few dozen steps: 60-90 ms
    # stalls of a host loop that does 5-7 ms of work per step - tools/step_times.py: one step in ~30 at 90 ms instead of 9.5.
    # A process-wide setting, but a harmless one: frozen objects are the ones a training process keeps for its whole life anyway.
    self._gc_freeze_pending = bool(getattr(cfg, 'gc_freeze', True))
    self._unchecked, self._flag_pool = [], []
    if self.dtype != 'bfloat16':
      raise NotImplementedError(f"dtype '{self.dtype}': the gfx950 kernels implement the bfloat16 flow only")
    if 'cuda' not in str(device):
      raise RuntimeError('HipEngine needs an MI355X device (got %r); there is no CPU path' % (device,))

    if getattr(cfg, 'resume', False):
      model.load_state_dict(ckpt['state_dict'])
      self.micro_steps = ckpt['step'] * cfg.grad_accumulation_steps

    self.model = model.to(device)
    flat = self.model.enable_main_grad()
    self.params = list(self.model.parameters())

    # optimizer first: FlatAdamW re-lays parameters / gradients into one span per weight-decay group
    self.scaler = torch.amp.GradScaler(enabled=False)  # bf16 needs no loss scaling; kept for checkpoint layout
    param_groups = get_param_groups(model, cfg.weight_decay)
    self.optimizer = intialize_optimizer(param_groups, cfg, self.model)
    self.scheduler = initialize_scheduler(self.optimizer, cfg)

    self.reducer = None
    if dist.is_initialized() and dist.get_world_size() > 1:
      # default: the uncapped root communicator alone, ncclAllReduce, no CU reserve - the data plane bench.py times first (ddp.COMM_CUS).
      # PLM_COMM_CUS=<n> adds a child capped at n workgroups for the buckets reduced while backward runs (the tail bucket - embed_tokens,
      # ready when backward has ended - then goes through the root unless PLM_COMM_TAIL=0) and the same GEMM-side reserve
      comm, comm_tail, reserve = ddp.pick_comms(ddp.make_comm_set(device, comm_backend))
      self.reducer = ddp.GradReducer(flat, self.params, self.model._grad_spans, comm, bucket_cap_mb=bucket_cap_mb,
                                      writers=self.model.grad_writers(), comm_tail=comm_tail, reserve_cus=reserve,
                                      groups=self.model.grad_groups())
      self.reducer.broadcast_params([p.data for p in self.params])  # DDP ctor's _sync_module_states
      self.model.invalidate_shadows()
      self.model.sink.on_ready = self.reducer.param_ready
      self.model.sink.on_queued = self.reducer.param_queued
      # every rank flushes its dW queue at the same points (ranks that differ would launch their buckets in different orders)
      self.model.sink.resolve_queue_budget(device, agreed=ddp.agree_min(self.model.sink.resolve_queue_budget(device)))

    if getattr(cfg, 'resume', False):
      self.optimizer.load_state_dict(ckpt['optimizer'])
      if self.scheduler is not None:
        self.scheduler.load_state_dict(ckpt['scheduler'])
      self.scaler.load_state_dict(ckpt['scaler'])

  def step(self, batch):
    self.model.train()
    self.micro_steps += 1
    self.accumulated_samples += 1
    inputs, targets, doc_start = _move_to_device(batch, self.seq_len, self.device, self.intra_doc_masking, self._stager)

    last = self.accumulated_samples == self.accumulation_steps
    if self.accumulated_samples == 1:
      self.model.sink.begin_window()
    if self.reducer is not None:
      self.reducer.begin(sync=last)

    try:
      loss = self.model.loss(inputs, targets, doc_start)
      loss_val = loss.detach()
      self._submit_nan_flag(loss_val)
      self.check_losses(keep=self.nan_check_lag)
      (loss / self.accumulation_steps).backward()
    except BaseException:
      if self.reducer is not None:  # 'Train loss is nan', out of memory: the launch hook and the CU reserve must not outlive the step
        self.reducer.abort()
      raise
    if self.reducer is not None:
      self.reducer.finish()

    if last:
      self.accumulated_samples = 0
      self.check_losses()  # no optimizer update from a window that contains a NaN loss
      if hasattr(self.optimizer, 'clip_and_step'):
        self.optimizer.clip_and_step(self.grad_clip or None)  # fused global-norm clip + AdamW on the flat buffers
      else:
        self.model.attach_grads()
        if self.grad_clip:
          torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
        self.optimizer.step()
      self.optimizer.zero_grad(set_to_none=True)
      if self.scheduler:
        self.scheduler.step()
      if self._gc_freeze_pending:
        import gc
        gc.collect()
        gc.freeze()
        self._gc_freeze_pending = False
    return loss_val

  def _submit_nan_flag(self, loss_val):
    """isnan(loss) -> pinned host byte, copied asynchronously; the event marks the copy.  Reading the flag later waits
    for that event only (a `.item()` on the device tensor would wait for everything submitted since)."""
    if len(self._flag_pool) > 0:
      host, ev = self._flag_pool.pop()
    else:
      host, ev = torch.empty((), dtype=torch.bool).pin_memory(), torch.cuda.Event()
    host.copy_(torch.isnan(loss_val), non_blocking=True)
    ev.record()
    self._unchecked.append((host, ev))

  def check_losses(self, keep=0):
    """Raise the reference's ValueError for every submitted micro-step but the newest `keep`."""
    while len(self._unchecked) > keep:
      host, ev = self._unchecked.pop(0)
      ev.synchronize()
      bad = bool(host)
      self._flag_pool.append((host, ev))
      if bad:
        self._unchecked.clear()
        raise ValueError('Train loss is nan')

  @torch.no_grad()
  def eval(self, dataloader):
    self.check_losses()
    self.model.eval()
    total_loss, num_batches = 0.0, 0
    for batch in dataloader:
      inputs, targets, doc_start = _move_to_device(batch, self.seq_len, self.device, self.intra_doc_masking, self._stager)
      loss = self.model.loss(inputs, targets, doc_start)
      if torch.isnan(loss):
        raise ValueError('Validation loss is nan')
      total_loss += loss.item()
      num_batches += 1
    if dist.is_initialized():
      t = torch.tensor([total_loss, float(num_batches)], device=self.device, dtype=torch.float64)
      dist.all_reduce(t, op=dist.ReduceOp.SUM)
      total_loss, num_batches = t[0].item(), t[1].item()
    return total_loss / num_batches


TorchEngine = HipEngine  # the name train.py imports (train.py:44)

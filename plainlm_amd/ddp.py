"""Bucketed data-parallel gradient all-reduce over a flat fp32 gradient buffer.

Replaces torch DDP's reducer (engine/engine.py:64-65,104-105):
gradients are produced by our backward kernels directly into one flat buffer, so a bucket
is a contiguous span.  When the last parameter of a bucket has been written, an event is
recorded on the compute stream, the communication stream waits for it and runs one
all-reduce-mean of the span; ``finish()`` joins the streams before clip/optimizer.

Backends:
  'rccl'  — RCCL called directly through the C ABI (plm_comm_*) on our side HIP stream.
  'torch' — torch.distributed (nccl == RCCL on ROCm, gloo on CPU for the world_size>1 tests).
"""

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib


# CUs kept free for RCCL's kernels while a gradient bucket's collective is running, and the matching cap on RCCL channels (one
# workgroup each; 0 = no reserve and RCCL's own channel count).  The persistent GEMMs launch one workgroup per CU; a workgroup that
# finds its CU taken by a collective would run after the others (a second round), so the GEMM grids shrink by this many while - and
# only while - an all-reduce is expected to be in flight (GradReducer: a window of estimated GPU time behind every bucket launch);
# the forward pass, the lm_head backward and every GEMM behind a finished collective keep all 256 CUs.  Sizing: 649 MB of fp32
# gradients per ~20 ms of backward is 32 GB/s of algorithm bandwidth = 57 GB/s of bus bandwidth at 8 ranks; one RCCL channel moves
# 15-25 GB/s over an xGMI link.  The SIZE of the reserve hardly matters to the GEMMs (4 vs 16 CUs: 0.1-0.2 ms per step,
# profiles/r04_ddp_whatif.txt).
# DEFAULT: NO cap and NO reserve (round 6).  None of this machinery has ever run with more than one RCCL rank (no multi-GPU box has been
# available to this build), so the engine's first contact is the most travelled path of the library - the uncapped root communicator,
# ncclAllReduce, every CU left to the GEMMs - exactly what bench.py times first.  A capped child communicator + the GEMM-side reserve are
# opt-in: PLM_COMM_CUS=<workgroups> (with PLM_COMM_TAIL / PLM_COMM_ALGO) - the variables through which the winner of bench.py's autotune
# table (`comm.alternatives` of a --gpus N line: {algo, cap, tail}) is handed to HipEngine: cap -> PLM_COMM_CUS, algo -> PLM_COMM_ALGO,
# tail -> PLM_COMM_TAIL (1 / 0).
COMM_CUS = int(os.environ.get('PLM_COMM_CUS', '0'))
# Prior rates (flop/s) by kernel family for the reducer's clock of enqueued GPU time.  They only fix the RATIOS between families:
# the clock is rescaled every step by measured / estimated time of the previous step (HIP events around begin() ... finish()), so a
# slower or faster box, or the 420M shapes, do not drift the windows away from the collectives they follow.
_FAMILY_RATE = {'gemm_nt': 1.15e15, 'gemm_nt_fused': 0.85e15, 'gemm_tn': 1.15e15, 'attn_fwd': 0.55e15, 'attn_bwd': 0.36e15}
# How the reserve windows get their lengths (PLM_COMM_WINDOWS):
#   'frozen' (default) - a bucket's window is its collective's measured duration (HIP events on the side stream, EMA over the first
#            FREEZE_AFTER communicating steps), then the ranks agree on the maximum over ranks once (control plane) and the table - and
#            the clock scale - never change again: from that step on every rank takes the same GEMM plans in every step.
#   'model'  - bytes / PLM_COMM_MODEL_GBPS (implied when that variable is set): no measurement at all; the only mode in which a
#            data-parallel run is bit-reproducible RUN TO RUN (the plans - stream-K / split-K partitions - depend on the windows).
FREEZE_AFTER = 3
# The caps are per communicator (ncclConfig_t.maxCTAs through plm_comm_split); NCCL_MAX_NCHANNELS is never set by this package
# (round 4 exported it process-wide as belt and braces, which also throttled the exposed tail bucket).  The communicator
# used for buckets that are reduced when backward has ended (embed_tokens: nothing left to overlap with) is the UNCAPPED root
# communicator (PLM_COMM_TAIL=0 sends them through the capped one).


class RcclComm:
  """Direct RCCL communicator (one per process = one per GPU).

  Creation is collective.  The unique id is exchanged over the already-initialised control-plane group; rank 0 ALWAYS takes
  part in that broadcast (it ships ``None`` when ncclGetUniqueId failed), so a failure before ncclCommInitRank raises on
  every rank together and make_comm's fallback stays in step.  A failure INSIDE ncclCommInitRank on a subset of the ranks is
  not recoverable (the others block in it) - that is RCCL's contract, not ours."""

  def __init__(self, rank, world_size, device_index, store_group=None, max_ctas=None, _handle=None):
    lib = _lib.load()
    self.lib, self.rank, self.world_size, self.device_index = lib, rank, world_size, device_index
    self.max_ctas = (max(COMM_CUS, 0) if world_size > 1 else 0) if max_ctas is None else int(max_ctas)
    self.backend = 'rccl-direct'
    if _handle is not None:  # split()
      self.handle = _handle
      return
    uid, err = (C.c_uint8 * 128)(), None
    if rank == 0:
      try:
        _lib.check(lib.plm_comm_unique_id(C.cast(uid, C.c_void_p)), 'plm_comm_unique_id')
      except RuntimeError as e:
        err = e
    if world_size > 1:
      # ship the 128-byte id over the control-plane group (gloo or nccl); None = rank 0 has no id
      obj = [bytes(uid) if err is None else None] if rank == 0 else [None]
      dist.broadcast_object_list(obj, src=0, group=store_group)
      if obj[0] is None:
        raise err if err is not None else RuntimeError('rank 0 could not create an RCCL unique id')
      uid = (C.c_uint8 * 128).from_buffer_copy(obj[0])
    elif err is not None:
      raise err
    handle = C.c_void_p()
    _lib.check(lib.plm_comm_init_capped(C.byref(handle), C.cast(uid, C.c_void_p), rank, world_size, device_index, self.max_ctas),
               'plm_comm_init_capped')
    self.handle = handle

  def split(self, max_ctas=0):
    """A second communicator over the same ranks with its own CTA cap (0 = RCCL's default); collective."""
    child = C.c_void_p()
    _lib.check(self.lib.plm_comm_split(self.handle, C.byref(child), int(max_ctas)), 'plm_comm_split')
    return RcclComm(self.rank, self.world_size, self.device_index, max_ctas=max_ctas, _handle=child)

  def allreduce_avg_(self, span, stream, algo=None):
    # algo 'rsag': the mean as reduce-scatter + all-gather in place (one-hop collectives over all xGMI links) instead of RCCL's own
    # all-reduce (PLM_COMM_ALGO is the default when the caller does not say; bench.py's autotune measures both)
    algo = algo or os.environ.get('PLM_COMM_ALGO') or 'allreduce'
    fn, name = ((self.lib.plm_comm_rsag_avg_f32, 'plm_comm_rsag_avg_f32') if algo == 'rsag' else
                (self.lib.plm_comm_allreduce_avg_f32, 'plm_comm_allreduce_avg_f32'))
    _lib.check(fn(self.handle, C.c_void_p(span.data_ptr()), span.numel(), C.c_void_p(stream.cuda_stream)), name)

  def broadcast_(self, span, root, stream):
    _lib.check(self.lib.plm_comm_broadcast_f32(self.handle, C.c_void_p(span.data_ptr()), span.numel(), root,
                                               C.c_void_p(stream.cuda_stream)), 'plm_comm_broadcast_f32')

  def close(self):
    if self.handle:
      self.lib.plm_comm_destroy(self.handle)
      self.handle = None


class TorchDistComm:
  """torch.distributed backend (gloo on CPU for tests; nccl==RCCL on GPU)."""

  def __init__(self, group=None):
    self.group = group
    self.rank = dist.get_rank(group)
    self.world_size = dist.get_world_size(group)
    self.backend = 'torch-' + dist.get_backend(group)

  def allreduce_avg_(self, span, stream=None, algo=None):  # algo: RCCL-direct only
    if dist.get_backend(self.group) == 'gloo':  # gloo has no AVG
      dist.all_reduce(span, op=dist.ReduceOp.SUM, group=self.group)
      span.div_(self.world_size)
    else:
      dist.all_reduce(span, op=dist.ReduceOp.AVG, group=self.group)

  def broadcast_(self, span, root, stream=None):
    dist.broadcast(span, src=root, group=self.group)

  def close(self):
    pass


def plan_buckets(spans, cap_bytes, groups=None):
  """spans: [(offset, numel)] per parameter (any order, together tiling the flat buffer without gaps).
  Returns buckets as (lo, hi, [param idx]): contiguous regions of the flat buffer, built from its END backwards
  (the flat layout follows parameters() order, so the end holds the gradients that become ready first), each at
  most cap_bytes unless a single parameter is larger (it then forms its own bucket).
  groups: optional lists of parameter indices that belong together (the Linear weights of one transformer block: their
  gradients come out of ONE grouped dW launch).  A group that fits a bucket is never split across two: the walk closes the
  current bucket in front of a group that would not fit into the rest of it."""
  order = sorted(range(len(spans)), key=lambda i: spans[i][0], reverse=True)
  group_of, group_bytes = {}, {}
  for gi, g in enumerate(groups or []):
    for i in g:
      group_of[i] = gi
    group_bytes[gi] = sum(spans[i][1] for i in g) * 4
  seen_groups = set()
  buckets, cur, cur_bytes = [], [], 0
  for idx in order:
    nbytes = spans[idx][1] * 4
    gi = group_of.get(idx)
    if gi is not None and gi not in seen_groups:  # first (= highest) member of its group
      seen_groups.add(gi)
      if cur and group_bytes[gi] <= cap_bytes < cur_bytes + group_bytes[gi]:
        buckets.append(cur)
        cur, cur_bytes = [], 0
    if cur and cur_bytes + nbytes > cap_bytes:
      buckets.append(cur)
      cur, cur_bytes = [], 0
    cur.append(idx)
    cur_bytes += nbytes
  if cur:
    buckets.append(cur)
  # a trailing bucket of small tensors (< 1 MiB: the norm weights, which FlatAdamW lays in front of embed_tokens) joins its
  # predecessor instead of costing one more latency-bound collective at the very end of backward
  if len(buckets) >= 2:
    last, prev = (sum(spans[i][1] for i in b) * 4 for b in (buckets[-1], buckets[-2]))
    if last < (1 << 20) and last * 100 < prev:
      buckets[-2].extend(buckets.pop())
  out = []
  for b in buckets:
    lo = min(spans[i][0] for i in b)
    hi = max(spans[i][0] + spans[i][1] for i in b)
    assert hi - lo == sum(spans[i][1] for i in b), 'bucket must be a contiguous span'
    out.append((lo, hi, sorted(b)))
  return out


class GradReducer:
  """Owns the bucket plan and the side stream.  Usage per optimizer step:

      reducer.begin(sync=is_last_micro_step)     # before backward
      ... backward: sink.on_ready -> reducer.param_ready(p)
      reducer.finish()                           # after backward, before clip/optimizer
  """

  def __init__(self, flat_grad, params, spans, comm, bucket_cap_mb=64, overlap=True, force=False, reserve_cus=None,
               writers=None, comm_tail=None, groups=None, algo=None, ctl_group=None):
    """groups: see plan_buckets (Transformer.grad_groups()).
    writers: optional {id(param): n} = how many backward kernels write that parameter's gradient per backward pass
    (default 1).  A weight shared by lm_head and embed_tokens (tie_embeddings, models/transformer.py:131-132) has two: the
    head's dW first, the embedding scatter last.  A bucket is launched only when every writer of every member has
    reported; launching after the first would let RCCL reduce the span in place while the second kernel still adds to it.
    algo: 'allreduce' | 'rsag' (None: PLM_COMM_ALGO, else all-reduce).  ctl_group: control-plane process group for the one-off
    agreement on the window table (None: the default group)."""
    self.flat = flat_grad
    self.comm = comm
    # Buckets that become ready when backward has ended (nothing left to hide them behind) go through comm_tail when one is
    # given (see COMM_CUS): the bucket of params[0] - embed_tokens, whose gradient is the last kernel of backward - and
    # whatever finish() still has to launch.
    self.comm_tail = comm_tail
    self.algo = algo
    self.ctl_group = ctl_group
    self.buckets = plan_buckets(spans, int(bucket_cap_mb * (1 << 20)), groups)
    self.numel = [n for _, n in spans]
    self.index_of = {id(p): i for i, p in enumerate(params)}
    self.writers = [int((writers or {}).get(id(p), 1)) for p in params]
    if min(self.writers, default=1) < 1:
      raise ValueError('GradReducer: every parameter needs at least one gradient writer')
    self.bucket_of = {}
    for b, (_, _, idxs) in enumerate(self.buckets):
      for i in idxs:
        self.bucket_of[i] = b
    self.tail_bucket = self.bucket_of[0] if params else -1
    self.on_gpu = flat_grad.is_cuda
    self.overlap = overlap and self.on_gpu
    self.stream = torch.cuda.Stream(device=flat_grad.device) if self.on_gpu else None
    self.force = force  # run the collectives even with a single rank (tests the RCCL path on one GPU)
    self.sync = False
    self.pending = None
    self.launched = []
    # CUs left to the collectives while one is expected to be running (see COMM_CUS).  The host enqueues kernels milliseconds ahead
    # of the GPU, so "is a collective running when THIS GEMM starts" is answered on a clock of estimated GPU time: every MFMA launch
    # advances it by flops / its family's rate x a learned scale (ops.LAUNCH_HOOK), a bucket launch opens a window [now or the end of
    # the previous window, + the bucket's duration], and a launch inside a window shrinks its grid.  Window lengths: see
    # PLM_COMM_WINDOWS above.
    if reserve_cus is None:
      reserve_cus = COMM_CUS if (self.on_gpu and comm.world_size > 1) else 0
    self.reserve_cus = int(reserve_cus) if self.on_gpu else 0
    self._reserved = False
    model_gbps = os.environ.get('PLM_COMM_MODEL_GBPS')
    self.model_gbps = float(model_gbps) if model_gbps else None
    self.window_mode = 'model' if self.model_gbps else os.environ.get('PLM_COMM_WINDOWS', 'frozen')
    if self.window_mode not in ('model', 'frozen'):
      raise ValueError(f"PLM_COMM_WINDOWS={self.window_mode!r}: expected 'frozen' or 'model'")
    self._bucket_bytes = [(hi - lo) * 4 for lo, hi, _ in self.buckets]
    self.reset_windows()
    # the dW queue (functional.GradSink) is flushed at bucket boundaries once it holds this many bytes of gradients
    self.dw_group_bytes = int(float(os.environ.get('PLM_DW_GROUP_MB', '80')) * 1e6)
    self._queued, self._queued_per_bucket, self._queued_bytes = set(), {}, 0
    self._exposed = None  # (event before, event after) the join of the last communicating step
    self.n_mfma_launches = self.n_reserved_launches = 0

  def reset_windows(self):
    """Forget what was learned about the collectives' durations and the clock (a different communicator / algorithm / cap)."""
    gbps = self.model_gbps or 60.0
    self.bucket_secs = [nb / (gbps * 1e9) for nb in self._bucket_bytes]
    # Measurements of the communicating steps still in flight, oldest first: {'buckets': {b: (start, end)}, 'span': (event at begin(),
    # event in front of the join, estimated clock there)}.  The host runs a step or two ahead of the GPU, so a step's events are read
    # when a LATER begin() finds them complete - never waited for, except once, at the freeze.
    self._hist = []
    self._cur = None
    self._event_pool = []
    self.rate_scale = 1.0      # measured / estimated GPU time from begin() to the join: rescales _FAMILY_RATE
    self.sync_steps = 0        # communicating steps begun since the last reset
    self.learned_steps = 0     # ... whose measurements have been folded in
    self.frozen = self.window_mode == 'model'
    self.clock, self.window_end = 0.0, -1.0

  def _event(self):
    return self._event_pool.pop() if self._event_pool else torch.cuda.Event(enable_timing=True)

  def configure(self, comm=None, comm_tail=False, reserve_cus=None, algo=None):
    """Switch the data plane between two steps (bench.py's autotune): any of the communicator used while backward runs, the one
    for the tail buckets (None = the same), the CU reserve and the algorithm; what had been learned about durations is dropped."""
    if self.sync:
      raise RuntimeError('GradReducer.configure: inside a step (between begin() and finish())')
    if comm is not None:
      self.comm = comm
    if comm_tail is not False:
      self.comm_tail = comm_tail
    if reserve_cus is not None:
      self.reserve_cus = int(reserve_cus) if self.on_gpu else 0
    if algo is not None:
      self.algo = algo
    self.reset_windows()

  def _fold(self, rec):
    for b, (t0, t1) in rec['buckets'].items():
      # clamped to bytes / 400 GB/s ... bytes / 10 GB/s: a collective measured while a rank was late (or a 1-rank local copy) must
      # not switch the reserve off for good or hold it for the whole of backward
      secs = min(max(t0.elapsed_time(t1) * 1e-3, self._bucket_bytes[b] / 400e9), self._bucket_bytes[b] / 10e9)
      self.bucket_secs[b] = secs if self.learned_steps == 0 else 0.5 * self.bucket_secs[b] + 0.5 * secs
      self._event_pool += [t0, t1]
    if rec['span'] is not None:
      t0, t1, est, scale = rec['span']
      if est > 0:
        new = scale * (t0.elapsed_time(t1) * 1e-3) / est  # est was computed with `scale`
        self.rate_scale = min(max(new if self.learned_steps == 0 else 0.5 * self.rate_scale + 0.5 * new, 0.25), 4.0)
      self._event_pool += [t0, t1]
    self.learned_steps += 1

  def _learn(self):
    """Top of a communicating step: fold the measurements of earlier steps that have finished on the GPU into the window table and
    the clock scale.  'frozen' mode: at the top of step FREEZE_AFTER + 1 (the same step on every rank - the agreement is a collective)
    the host waits ONCE for the steps it has enqueued, folds them, the ranks agree on the elementwise maximum, and nothing changes
    any more; 'model' never measures."""
    if self.frozen:
      return
    freeze_now = self.sync_steps >= FREEZE_AFTER
    while self._hist and self._hist[0] is not self._cur:
      rec = self._hist[0]
      ends = [t1 for _, t1 in rec['buckets'].values()] + ([rec['span'][1]] if rec['span'] is not None else [])
      if not all(e.query() for e in ends):
        if not freeze_now:
          break
        for e in ends:
          e.synchronize()
      self._hist.pop(0)
      self._fold(rec)
    if freeze_now:
      vals = agree_max_floats(self.bucket_secs + [self.rate_scale], self.ctl_group)
      self.bucket_secs, self.rate_scale = vals[:-1], vals[-1]
      self.frozen = True
      self._hist = []

  def begin(self, sync):
    if self.on_gpu and (self._reserved or self.sync):
      self.abort()  # a step that raised between begin() and finish() left the hook / the reserve behind
    self.sync = bool(sync) and (self.comm.world_size > 1 or self.force)
    self.pending = [sum(self.writers[i] for i in idxs) for (_, _, idxs) in self.buckets]
    self.reports = [0] * len(self.writers)
    self.launched = []
    self._queued, self._queued_per_bucket, self._queued_bytes = set(), {}, 0
    self.clock, self.window_end = 0.0, -1.0
    if self.on_gpu and self.sync:
      self._learn()
      self.sync_steps += 1
      self._cur = None
      if not self.frozen:
        self._cur = {'buckets': {}, 'span': None, 't0': None}
        if len(self._hist) < 8:  # a GPU that never catches up (it does): stop recording rather than grow
          self._hist.append(self._cur)
      if self.reserve_cus:
        from . import ops
        ops.LAUNCH_HOOK = self._on_launch
        if self._cur is not None:
          self._cur['t0'] = self._event()
          self._cur['t0'].record()

  def abort(self):
    """Drop the launch hook and the CU reserve (a step raised between begin() and finish(): 'Train loss is nan', out of memory);
    engine.step calls this from its exception path, begin() as a safety net.  Collectives already enqueued are joined."""
    from . import ops
    if self.on_gpu:
      if ops.LAUNCH_HOOK == self._on_launch:
        ops.LAUNCH_HOOK = None
      if self._reserved:
        ops.set_cu_reserve(0)
        self._reserved = False
      torch.cuda.current_stream().wait_stream(self.stream)
    if self._cur is not None:  # a half-recorded step is not a measurement
      self._hist = [r for r in self._hist if r is not self._cur]
      self._cur = None
    self.sync = False

  def _on_launch(self, family, flops):
    """ops.LAUNCH_HOOK: called right before an MFMA kernel is enqueued."""
    want = self.clock < self.window_end
    if want != self._reserved:
      from . import ops
      ops.set_cu_reserve(self.reserve_cus if want else 0)
      self._reserved = want
    self.n_mfma_launches += 1
    self.n_reserved_launches += int(want)
    self.clock += self.rate_scale * flops / _FAMILY_RATE.get(family, 1.0e15)

  def stats(self):
    """What the windows are doing (bench.py prints it; ADVICE r04: nothing used to show whether the reserve was ever on)."""
    exposed = None
    if self._exposed is not None and self._exposed[1].query():
      exposed = round(self._exposed[0].elapsed_time(self._exposed[1]), 3)
    return {'window_mode': self.window_mode, 'frozen': self.frozen, 'bucket_ms': [round(1e3 * v, 3) for v in self.bucket_secs],
            'clock_scale': round(self.rate_scale, 3), 'exposed_comm_ms': exposed,
            'reserved_launch_frac': round(self.n_reserved_launches / max(1, self.n_mfma_launches), 3)}

  def param_queued(self, p):
    """functional.GradSink.on_queued: the weight gradient of p has been queued for a grouped dW launch.  True = launch the group
    now: the queue holds every outstanding gradient of p's bucket (the bucket can go out as soon as the group has run) and at
    least dw_group_bytes of gradients (small groups leave most of their tiles to the split-K remainder).  None on
    micro-steps that do not communicate (the sink's count-based default decides)."""
    if not self.sync:
      return None
    i = self.index_of.get(id(p))
    if i is None or i in self._queued:
      return None
    b = self.bucket_of[i]
    self._queued.add(i)
    self._queued_per_bucket[b] = self._queued_per_bucket.get(b, 0) + 1
    self._queued_bytes += self.numel[i] * 4
    return self._queued_per_bucket[b] >= self.pending[b] and self._queued_bytes >= self.dw_group_bytes

  def _launch(self, b, tail=False):
    lo, hi, _ = self.buckets[b]
    span = self.flat[lo:hi]
    comm = self.comm_tail if (self.comm_tail is not None and (tail or b == self.tail_bucket)) else self.comm
    if self.on_gpu:
      ev = torch.cuda.Event()
      ev.record(torch.cuda.current_stream())
      self.stream.wait_event(ev)
      with torch.cuda.stream(self.stream):
        if self._cur is not None:
          t0, t1 = self._event(), self._event()
          t0.record(self.stream)
        comm.allreduce_avg_(span, self.stream, self.algo)
        if self._cur is not None:
          t1.record(self.stream)
          self._cur['buckets'][b] = (t0, t1)
      # GEMMs enqueued inside this window may run beside the collective (see __init__)
      self.window_end = max(self.window_end, self.clock) + self.bucket_secs[b]
    else:
      comm.allreduce_avg_(span, None, self.algo)
    self.launched.append(b)

  def param_ready(self, p):
    if not self.sync:
      return
    i = self.index_of.get(id(p))
    if i is None:
      return
    b = self.bucket_of[i]
    if i in self._queued:
      self._queued.discard(i)
      self._queued_per_bucket[b] -= 1
      self._queued_bytes -= self.numel[i] * 4
    self.reports[i] += 1
    if self.reports[i] > self.writers[i]:
      raise RuntimeError(f'GradReducer: parameter {i} reported more gradient writes than declared ({self.writers[i]}); '
                         'a shared weight needs writers={id(p): n}')
    self.pending[b] -= 1
    if self.pending[b] == 0 and self.overlap:
      self._launch(b)

  def finish(self):
    """Reduce whatever was not launched during backward and make the compute stream wait."""
    if not self.sync:
      return
    for b in range(len(self.buckets)):
      if b not in self.launched:
        self._launch(b, tail=True)
    if self.on_gpu:
      from . import ops
      cur = torch.cuda.current_stream()
      if ops.LAUNCH_HOOK == self._on_launch:
        ops.LAUNCH_HOOK = None
        if self._cur is not None and self._cur.get('t0') is not None:
          t1 = self._event()
          t1.record(cur)  # measured vs estimated GPU time of this step's kernels, begin() ... in front of the join
          self._cur['span'] = (self._cur['t0'], t1, self.clock, self.rate_scale)
      # the join, bracketed by events: what the compute stream waits here is the EXPOSED part of the step's communication
      e0, e1 = self._exposed or (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
      e0.record(cur)
      cur.wait_stream(self.stream)
      e1.record(cur)
      self._exposed = (e0, e1)
      if self._reserved:  # everything enqueued after the join runs with no collective in flight
        ops.set_cu_reserve(0)
        self._reserved = False
    self._cur = None
    self.sync = False

  def broadcast_params(self, flat_params_or_list):
    """C1: rank-0 parameters to every rank once at wrap time."""
    if self.comm.world_size == 1 and not self.force:
      return
    tensors = flat_params_or_list if isinstance(flat_params_or_list, (list, tuple)) else [flat_params_or_list]
    for t in tensors:
      if self.on_gpu:
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
          self.comm.broadcast_(t.view(-1), 0, self.stream)
        torch.cuda.current_stream().wait_stream(self.stream)
      else:
        self.comm.broadcast_(t.view(-1), 0, None)


def make_comm_set(device, backend=None, group=None, caps=None):
  """The data-plane communicators of a run: {0: uncapped root, cap: split of the root limited to `cap` workgroups, ...}.
  On GPU the default is direct RCCL through the C ABI: the root is a plain ncclCommInitRank (the most travelled path of the
  library), every capped communicator is an ncclCommSplit of it with ncclConfig_t.maxCTAs.  Collective; every decision that can
  differ between ranks (a failed creation) is agreed on the control plane, so the ranks always end up with the same set.
  backend 'torch' (or a failed direct set-up): {0: TorchDistComm} - torch.distributed has no per-communicator cap, the GEMM-side
  reserve is then the only knob."""
  caps = ([COMM_CUS] if COMM_CUS > 0 else []) if caps is None else list(caps)  # default: the root alone (see COMM_CUS)
  world = dist.get_world_size(group) if dist.is_initialized() else 1
  rank = dist.get_rank(group) if dist.is_initialized() else 0
  backend = backend or os.environ.get('PLM_COMM', 'rccl' if torch.device(device).type == 'cuda' else 'torch')
  if backend != 'rccl':
    if not dist.is_initialized():
      raise RuntimeError("backend 'torch' needs torch.distributed to be initialised")
    return {0: TorchDistComm(group)}
  idx = torch.device(device).index
  idx = torch.cuda.current_device() if idx is None else idx
  root, err = None, None
  try:
    root = RcclComm(rank, world, idx, store_group=group, max_ctas=0)
  except Exception as e:  # noqa: BLE001 - reported below, on every rank
    err = e
  if not all_ranks_ok(err is None, group):
    if root is not None:
      root.close()
    if world == 1:
      raise err
    # if the direct communicator fails on ANY rank every rank falls back to torch.distributed's nccl (= RCCL) backend, so that a
    # box whose RCCL set-up differs from the build machine still trains
    print(f'[plainlm_amd.ddp] rank {rank}: direct RCCL communicator unavailable ({err}); using torch.distributed nccl', flush=True)
    return {0: TorchDistComm(dist.new_group(backend='nccl'))}
  return add_capped_comms({0: root}, caps, group)


def add_capped_comms(comms, caps, group=None):
  """Add one ncclCommSplit child of the root per cap (ncclConfig_t.maxCTAs) to a make_comm_set; collective, a cap whose split fails on
  any rank is left out on every rank.  bench.py calls it AFTER its first timed region: a run whose RCCL cannot split must not lose the
  measurement on the root communicator."""
  root = comms[0]
  if not isinstance(root, RcclComm):
    return comms
  for cap in sorted({int(c) for c in caps if int(c) > 0} - set(comms)):
    child, err = None, None
    try:
      child = root.split(max_ctas=cap)
    except RuntimeError as e:
      err = e
    if all_ranks_ok(err is None, group):
      comms[cap] = child
    else:
      if child is not None:
        child.close()
      print(f'[plainlm_amd.ddp] rank {root.rank}: no communicator capped at {cap} workgroups ({err}); that cap is not available', flush=True)
  return comms


def pick_comms(comms, cap=None, tail=None):
  """(communicator for the buckets reduced while backward runs, communicator for the tail buckets or None, CU reserve) out of a
  make_comm_set: the cap asked for (default COMM_CUS) when the set has it, else the uncapped root with the same GEMM-side reserve;
  tail (default: PLM_COMM_TAIL != 0) sends the buckets that are reduced after backward through the uncapped root."""
  cap = COMM_CUS if cap is None else int(cap)
  tail = (os.environ.get('PLM_COMM_TAIL', '1') != '0') if tail is None else bool(tail)
  comm = comms.get(cap, comms[0])
  comm_tail = comms[0] if (tail and comm is not comms[0]) else None
  return comm, comm_tail, max(cap, 0)


def make_comm(device, backend=None, group=None):
  """Engine entry point: (comm, comm_tail) with the default cap and tail policy."""
  comm, comm_tail, _ = pick_comms(make_comm_set(device, backend, group))
  return comm, comm_tail


def agree_min(value, group=None):
  """Minimum of an integer over the ranks (control plane; the value itself without a process group)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return int(value)
  dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
  t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
  return int(t.item())


def all_ranks_ok(ok, group=None):
  """True iff `ok` is true on every rank (control-plane all-reduce on CPU tensors; works on gloo and nccl groups)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return bool(ok)
  dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
  flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
  dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
  return bool(flag.item())


def agree_max_floats(values, group=None):
  """Elementwise maximum of a list of floats over the ranks (control plane; the values themselves without a process group)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return [float(v) for v in values]
  dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
  t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  return [float(v) for v in t.tolist()]


def agree_winner(local_ms, group=None):
  """Autotune consensus: every rank passes its own timing of each alternative (same order on all ranks); the time of an alternative is
  the SLOWEST rank's (that is the step time of a data-parallel job), the winner the smallest of those, ties to the lowest index.
  Returns (index, agreed timings) - identical on every rank whatever the local timings were."""
  agreed = agree_max_floats(local_ms, group)
  best = min(range(len(agreed)), key=lambda i: (agreed[i], i))
  return best, agreed


def apply_alternative(reducer, comms, alt):
  """Put a GradReducer on one data-plane alternative {'algo', 'cap', 'tail'} of a make_comm_set (between two steps)."""
  comm, comm_tail, reserve = pick_comms(comms, cap=alt['cap'], tail=alt['tail'])
  reducer.configure(comm=comm, comm_tail=comm_tail, reserve_cus=reserve, algo=alt['algo'])


def data_plane_alternatives(comms, first=None):
  """The alternatives bench.py's autotune measures on a communicator set: {all-reduce, reduce-scatter + all-gather} (direct RCCL only) x
  {no reserve, every cap the set has a child communicator for, PLM_COMM_CUS if set} x {tail buckets on the capped communicator, on the
  uncapped root} (only where the set has that cap); `first` (the data plane already timed) is kept / put in front."""
  direct = isinstance(comms[0], RcclComm)
  caps = sorted({0, 8, 16} | {int(c) for c in comms} | ({COMM_CUS} if COMM_CUS > 0 else set()))  # 8 / 16 without a child: root + GEMM-side reserve
  alts = []
  for algo in (('allreduce', 'rsag') if direct else ('allreduce',)):
    for cap in caps:
      for tail in ((False, True) if (cap and cap in comms) else (False,)):
        alts.append({'algo': algo, 'cap': cap, 'tail': tail})
  if first is not None and first not in alts:
    alts.insert(0, first)
  return alts


def autotune(reducer, comms, run_steps, first=None, n_try=4, caps=(8, 16), group=None):
  """Measure every data-plane alternative for a few steps and agree on the winner.  Collective: every rank calls it at the same point
  with the same arguments.  run_steps(n) runs n whole steps (begin ... finish through `reducer`), waits for them, and returns the
  seconds they took on THIS rank.  Per alternative: FREEZE_AFTER + 1 untimed steps (the reserve windows are learned and frozen - one host
  wait and one agreement - before the clock starts), then n_try timed ones.  Returns (alternatives, agreed ms per step = the slowest
  rank's, index of the winner); the reducer is left on the LAST alternative - the caller applies the one it wants."""
  add_capped_comms(comms, caps, group)
  alts = data_plane_alternatives(comms, first)
  local_ms = []
  for alt in alts:
    apply_alternative(reducer, comms, alt)
    run_steps(FREEZE_AFTER + 1)
    local_ms.append(1e3 * run_steps(n_try) / n_try)
  win, agreed = agree_winner(local_ms, group)
  return alts, agreed, win


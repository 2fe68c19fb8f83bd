"""bench.py — throughput of the plainLM hot path (fwd+bwd of the 160M decoder at seq 1024, bf16)
on N MI355X GPUs of one node, one process per GPU.

  python bench.py                                (N = 1, 50 timed steps after 10 warm-up steps)
  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --config 160m --seq-len 2048  (the reference's yaml files as shipped: tr_160M_x8gpu.yaml has seq_len 2048)
  python bench.py --doc-mask --micro-batch 8    (config_doc_mask.yaml's document-masked batches)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...      (no launcher: the parent, which never touches a GPU, starts the N rank processes itself)

A "step" = one fwd+bwd of one micro-batch of 32 x 1024 synthetic tokens per GPU through the fused
lm_head+cross-entropy loss, including the per-step fp32->bf16 weight casts (what autocast does
per forward) and, for N > 1, the bucketed RCCL gradient all-reduce overlapped with backward.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

After the timed region rank 0 also (untimed) measures: the same step with per-launch HIP events
for the roofline object (the dominant MFMA family, plus the HBM-bound kernel families under
``roofline.hbm``), a full training step (clip + AdamW) for reference, and — at N = 1 —
the CPU oracle on the host cores as ``cpu_baseline`` (a bounded sample of about 10 s).
"""

import argparse
import gc
import json
import os
import sys
import time
from collections import namedtuple

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this stack needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails under the legacy mode); the variable is read when HIP
# initialises, which nothing above has done (importing torch does not)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_GBPS = 8000.0     # HBM3E peak (MI355X_MICROARCH.md: ~8 TB/s)

CONFIGS = {
  # BASELINE.json configs[1]/[2]: 12-layer d=768 (160M) seq=1024 bf16, batch 32 per GPU
  '160m': dict(vocab_size=50280, seq_len=1024, d_model=768, n_layers=12, n_heads=12, micro_batch=32),
  # configs[3]: tr_420M_x8gpu.yaml
  '420m': dict(vocab_size=50280, seq_len=2048, d_model=1024, n_layers=24, n_heads=16, micro_batch=8),
}


def flops_per_token(c, hidden, pairs_per_token=None):
  """fwd = L(2d*3d + 2d*d + 2d*2h + 2h*d) + L*2d(T+1) + 2dV ; fwd+bwd = 3x.  The attention term is
  4 d x (visible (query, key) pairs per token): (T+1)/2 for the causal mask; with document masks the pairs the mask leaves
  (``pairs_per_token``, counted from doc_start) - flops the kernels skip are not flops the step did."""
  d, L, T, V = c['d_model'], c['n_layers'], c['seq_len'], c['vocab_size']
  pairs = (T + 1) / 2.0 if pairs_per_token is None else pairs_per_token
  fwd = L * (2 * d * 3 * d + 2 * d * d + 2 * d * 2 * hidden + 2 * hidden * d) + L * 4 * d * pairs + 2 * d * V
  return 3 * fwd


def build_model(c, device, seed=100):
  import plainlm_amd as P
  keys = dict(model='transformer', vocab_size=c['vocab_size'], seq_len=c['seq_len'], d_model=c['d_model'], expand='8/3',
              n_layers=c['n_layers'], n_heads=c['n_heads'], mlp_class='glu', tie_embeddings=False)
  cfg = namedtuple('Config', keys.keys())(**keys)
  torch.manual_seed(seed)
  from plainlm_amd.transformer import ModelConfig, Transformer
  from fractions import Fraction
  model = Transformer(ModelConfig(vocab_size=cfg.vocab_size, seq_len=cfg.seq_len, dim=cfg.d_model,
                                  expand=float(Fraction(cfg.expand)), n_layers=cfg.n_layers, n_heads=cfg.n_heads, mlp='glu'))
  return model.to(device)


def physical_cores():
  """Distinct (package, core) pairs of /proc/cpuinfo; falls back to os.cpu_count()."""
  try:
    seen, phys = set(), None
    for line in open('/proc/cpuinfo'):
      if line.startswith('physical id'):
        phys = line.split(':')[1].strip()
      elif line.startswith('core id'):
        seen.add((phys, line.split(':')[1].strip()))
    return len(seen) or (os.cpu_count() or 1)
  except OSError:
    return os.cpu_count() or 1


def cpu_baseline(c, budget_s=10.0):
  """Oracle (oracle/cpu_ref.py, fp32 eager) fwd+bwd on the host cores: B=1 sequences of seq_len tokens, a BOUNDED sample (about
  ``budget_s`` seconds: the GPU part of a default run is well under a second, the baseline must not be the run).
  More threads is not faster for this eager fp32 graph (oversubscription on big hosts): one untimed warm-up iteration, then one
  iteration each at 32 threads (the fastest count on every box of this pool so far) and at one thread per physical core (SURVEY.md
  section 8d's definition), then more iterations at the faster of the two while the budget lasts.  ``value`` / ``cores`` are the faster
  count's median, ``all_physical_cores`` the rate at one thread per physical core."""
  from oracle import cpu_ref as O
  ocfg = O.OracleConfig(vocab_size=c['vocab_size'], seq_len=c['seq_len'], dim=c['d_model'], n_layers=c['n_layers'],
                        n_heads=c['n_heads'])
  params = O.init_params(ocfg, seed=0)
  rng = np.random.default_rng(1234)
  tok = torch.from_numpy(rng.integers(0, c['vocab_size'], size=(1, c['seq_len'] + 1)))
  ids, tgt = tok[:, :-1], tok[:, 1:]
  ncpu, nphys = os.cpu_count() or 1, physical_cores()
  t_all = time.time()

  def one():
    t0 = time.time()
    O.loss_and_grads(params, ocfg, ids, tgt)
    return time.time() - t0

  default_threads = torch.get_num_threads()
  n_fast = min(32, ncpu)
  torch.set_num_threads(n_fast)
  one()  # warm-up (allocations, thread pool)
  best_n, best_t = n_fast, one()
  t_phys = best_t
  if nphys != n_fast:
    torch.set_num_threads(nphys)
    t_phys = one()
    if t_phys < best_t:
      best_n, best_t = nphys, t_phys
  torch.set_num_threads(best_n)
  times = [best_t]
  while len(times) < 5 and (time.time() - t_all) + best_t < budget_s:
    times.append(one())
  torch.set_num_threads(default_threads)
  med = float(np.median(times))
  return {'value': round(c['seq_len'] / med, 1), 'unit': 'tokens/s', 'cores': best_n, 'kind': 'port',
          'all_physical_cores': {'value': round(c['seq_len'] / t_phys, 1), 'cores': nphys},
          'seconds': round(time.time() - t_all, 1),
          'sample': f'oracle/cpu_ref.py fp32 eager fwd+bwd, batch 1 x {c["seq_len"]} tokens, median of {len(times)} iterations '
                    f'at the faster of {n_fast} / {nphys} torch threads (= {best_n}) on a host with {nphys} physical cores / '
                    f'{ncpu} logical CPUs, bounded to ~{budget_s:.0f} s; all_physical_cores = one iteration with {nphys} threads'}


def csrc_sha():
  """sha256 over the kernel sources the library is built from (names + contents, sorted): ties a PMC profile to a tree."""
  import hashlib
  d = os.path.join(ROOT, 'plainlm_amd', 'csrc')
  h = hashlib.sha256()
  for f in sorted(os.listdir(d)):
    if f.endswith(('.hip', '.h', '.cpp')) or f == 'Makefile':
      h.update(f.encode())
      h.update(open(os.path.join(d, f), 'rb').read())
  return h.hexdigest()[:16]


def _pmc_profile():
  """The committed PMC traffic profile to use: the newest profiles/r*_pmc_gemm_traffic.json taken on THIS tree's csrc (by hash); when
  none matches, the newest one (pmc_traffic then reports why it is not used)."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_gemm_traffic.json')), reverse=True)
  sha = csrc_sha()
  for f in files:
    try:
      if json.load(open(f)).get('csrc_sha') == sha:
        return f
    except Exception:  # noqa: BLE001 - an unreadable profile is simply not the one
      pass
  return files[0] if files else os.path.join(ROOT, 'profiles', 'none_pmc_gemm_traffic.json')




def pmc_traffic(family, config, tokens, n_layers):
  """HBM-side bytes per launch of the dominant kernel family, from the committed rocprofv3 PMC passes
  (profiles/r<NN>_pmc_gemm_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB -> bytes, one row per GEMM shape of the
  160M / 32768-token step; tools/prof_traffic.py + tools/pmc_traffic_report.py).  bench.py cannot run the profiler on
  itself, so the figure is the profile's, averaged over the launches of one step - and ONLY when the profile was taken on
  this very tree (the profile records the sha of plainlm_amd/csrc): otherwise traffic stays null and the reason is stated."""
  if family not in ('gemm_nt', 'gemm_nt_fused', 'gemm_tn') or config != '160m' or tokens != 32768:
    return {}
  PMC_PROFILE = _pmc_profile()
  if not os.path.exists(PMC_PROFILE):
    return {'traffic_note': 'no PMC profile committed for this workload'}
  prof = json.load(open(PMC_PROFILE))
  if prof.get('csrc_sha') != csrc_sha():
    return {'traffic_note': f'profiles/{os.path.basename(PMC_PROFILE)} was taken on csrc {prof.get("csrc_sha")}, this tree is '
                            f'{csrc_sha()}: not reported'}
  try:
    rows = {r['gemm']: r for r in prof['rows']}

    def row(name):  # exact name, or the row of the same launch with its fused epilogue spelled out ('nt fc1 fwd + swiglu (epilogue)')
      hits = [r for g, r in rows.items() if g == name or g.startswith(name + ' + ') or g.startswith(name + ' (')]
      main = [r for r in hits if 'reduce' not in r['gemm']]
      if not main:
        raise KeyError(name)
      parts = [r for r in hits if 'reduce' in r['gemm']]  # split-K / stream-K reduce kernels belong to their GEMM's launch
      return {k: main[0][k] + sum(q.get(k, 0.0) for q in parts) for k in ('traffic_bytes', 'algorithmic_bytes')}

    if family == 'gemm_nt':  # plain launches per layer: out fwd, fc2 fwd, dX qkv, dX out (the out-fwd shape), dX fc1; plus lm_head fwd / dX
      plan = [('nt out fwd', 2 * n_layers), ('nt fc2 fwd', n_layers), ('nt dX qkv', n_layers), ('nt dX fc1', n_layers),
              ('nt head fwd', 1), ('nt dX head', 1)]
    elif family == 'gemm_nt_fused':  # launches that carry an elementwise pass in their epilogue: qkv + RoPE, fc1 + SwiGLU, dX fc2 + SwiGLU bwd
      plan = [('nt qkv fwd', n_layers), ('nt fc1 fwd', n_layers), ('nt dX fc2', n_layers)]
    else:  # the dW GEMMs of the blocks as grouped launches of however many blocks the profile measured, plus lm_head
      import re
      nb = next(int(m.group(1)) for m in (re.match(r'tn dW (\d+) blocks', g) for g in rows) if m)
      plan = [(f'tn dW {nb} blocks', n_layers / float(nb)), ('tn dW head', 1)]
    launches = sum(c for _, c in plan)
    tot = lambda key: sum(row(g)[key] * c for g, c in plan)
    return {'traffic': round(tot('traffic_bytes') / launches), 'traffic_unit': 'bytes/launch (L2-miss side: FETCH_SIZE*2 + WRITE_SIZE, Infinity-Cache hits included)',
            'algorithmic_bytes': round(tot('algorithmic_bytes') / launches), 'traffic_source': f'profiles/{os.path.basename(PMC_PROFILE)}',
            'csrc_sha': prof['csrc_sha']}
  except Exception as e:  # a profile this code cannot read must never take the benchmark down
    return {'traffic_note': f'profiles/{os.path.basename(PMC_PROFILE)} could not be used: {type(e).__name__}: {e}'}


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher: start the N rank processes (one per GPU) as children with the
  torch.distributed.run environment and hand back the worst exit code.  The parent never initialises HIP (importing torch
  does not), and no process that has is ever re-exec'ed."""
  import socket
  import subprocess
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PLM_BENCH_CHILD='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
  rc = 0
  try:
    while procs:
      for p in list(procs):
        code = p.poll()
        if code is None:
          continue
        procs.remove(p)
        if code != 0:  # one rank died: the others would wait in a collective forever
          rc = rc or code
          for q in procs:
            q.terminate()
      time.sleep(0.2)
  finally:
    for q in procs:
      q.kill()
  return rc


def claim_stdout():
  """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio when a communicator is created (it lands AFTER
  Python's own output when stdout is a pipe: the C buffer is flushed at exit), and any other library may print as well - so this process's
  fd 1 is pointed at stderr for its whole life, and the JSON line is written to the descriptor returned here."""
  sys.stdout.flush()
  keep = os.dup(1)
  os.dup2(2, 1)
  return keep


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)   # 1.5 s of timed region at the headline shape (round 5's 20 steps were 0.6 s of a 30 s run)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--config', default='160m', choices=sorted(CONFIGS))
  ap.add_argument('--micro-batch', type=int, default=0)
  ap.add_argument('--seq-len', type=int, default=0, help='sequence length (default: the config\'s; config/config.yaml:9 ships the 160M model with 2048)')
  ap.add_argument('--no-extras', action='store_true', help='skip the untimed roofline / full-step / cpu legs')
  ap.add_argument('--comm', default=None, choices=[None, 'rccl', 'torch'])
  ap.add_argument('--bucket-mb', type=float, default=64)
  ap.add_argument('--no-autotune', action='store_true', help='data-parallel runs: time the default data plane only')
  ap.add_argument('--single-device', action='store_true',
                  help='plumbing check on a 1-GPU box: every rank uses cuda:0 and the gradient exchange runs over gloo (RCCL refuses '
                       'two ranks on one device); the printed rate is NOT a throughput measurement and says so')
  ap.add_argument('--doc-mask', action='store_true',
                  help='BASELINE configs[4]: document-boundary attention masks (random documents, mean length ~256)')
  a = ap.parse_args()

  if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    raise SystemExit(spawn_ranks(a.gpus))
  json_fd = claim_stdout()
  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if world != a.gpus:
    raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}')
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU fallback for the product path')
  if a.single_device:
    local_rank, a.comm = 0, 'torch'
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend='gloo')  # control plane only (barrier, timing max, RCCL id exchange)

  from plainlm_amd import ddp, ops
  if os.environ.get('PLM_CU_RESERVE'):  # single-GPU what-if: the persistent GEMM grids with CUs set aside as in a multi-GPU run
    ops.set_cu_reserve(int(os.environ['PLM_CU_RESERVE']))
  c = dict(CONFIGS[a.config])
  if a.micro_batch:
    c['micro_batch'] = a.micro_batch
  if a.seq_len:
    c['seq_len'] = a.seq_len
  B, T, V = c['micro_batch'], c['seq_len'], c['vocab_size']
  model = build_model(c, device)
  hidden = model.layers[0].mlp.hidden_dim
  flat = model.enable_main_grad()
  params = list(model.parameters())

  force_reducer = world == 1 and os.environ.get('PLM_FORCE_REDUCER')  # one-GPU what-if: the whole DDP data plane with a 1-rank communicator
  ddp_on = world > 1 or bool(force_reducer)
  st = {'reducer': None}  # the gradient reducer of the step being timed (None: single GPU, no data plane)

  # synthetic tokens: rank r takes rows r, r+W, ... (DistributedSampler(shuffle=False) order)
  n_pool = 4
  rng = np.random.default_rng(1234)
  tok = torch.from_numpy(rng.integers(0, V, size=(n_pool * B * world, T + 1)))[rank::world]
  def doc_starts(nrows):
    """per row: document lengths ~ geometric(1/256) truncated so they sum to T+1 (data_prep_utils.py:52-77), as doc_start[B,T]"""
    from plainlm_amd.engine import doc_start_from_lengths
    docs = []
    for _ in range(nrows):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
    return doc_start_from_lengths(docs, T).to(device)

  pool = [(tok[i * B:(i + 1) * B, :T].contiguous().to(device), tok[i * B:(i + 1) * B, 1:].contiguous().to(device),
           doc_starts(B) if a.doc_mask else None) for i in range(n_pool)]
  pairs_per_token = None  # visible (query, key) pairs per token, mean over the pool's batches: what the masked attention kernels compute
  if a.doc_mask:
    pos = torch.arange(T, device=device, dtype=torch.int64)[None, :]
    pairs_per_token = float(sum((pos - ds.to(torch.int64) + 1).sum().item() for _, _, ds in pool)) / (n_pool * B * T)

  def fwd_bwd(i, recast=True):
    ids, tgt, dstart = pool[i % n_pool]
    reducer = st['reducer']
    model.sink.begin_window()
    if reducer is not None:
      reducer.begin(sync=True)
    if recast:
      model.invalidate_shadows()  # a real step changes the weights: redo the bf16 casts every step like autocast
    loss = model.loss(ids, tgt, dstart)
    loss.backward()
    if reducer is not None:
      reducer.finish()
    return loss

  def barrier():
    if world > 1:
      dist.barrier()

  def timed(warmup, steps):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; returns (seconds = max over
    ranks, per-rank seconds, last loss)."""
    loss = None
    for i in range(warmup):
      loss = fwd_bwd(i)
    torch.cuda.synchronize()
    # Python's cyclic collector: a full collection walks every object the process holds (import torch alone creates ~10^6) and takes
    # 60-90 ms of host time - tools/step_times.py shows one such stall every few dozen steps, i.e. +3 ms per step on a 20-step region when it
    # lands inside.  Collect once here and move the survivors to the permanent generation (gc.freeze: the collector stays ON, later
    # collections only walk what the steps themselves allocate); plainlm_amd.HipEngine does the same after its first optimizer step
    # (cfg.gc_freeze, default True).
    gc.collect()
    gc.freeze()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
      loss = fwd_bwd(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    local = time.perf_counter() - t0
    per_rank = [local]
    if world > 1:
      tt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
      dist.all_gather(tt, torch.tensor([local], dtype=torch.float64))
      per_rank = [float(t.item()) for t in tt]
    return max(per_rank), per_rank, float(loss.item())

  # ---- N = 1 reference inside the same run (data-parallel runs only): rank 0 alone, before any communicator exists, the others
  # wait at the barrier - the scaling efficiency of this line does not depend on a second run on another box / another day
  n1_ms = None
  if ddp_on:  # (also in the --single-device plumbing check: every rank-asymmetric stretch of this file runs in the 2-rank GPU test)
    if rank == 0:
      for i in range(a.warmup):
        fwd_bwd(i)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(a.steps):
        fwd_bwd(i)
      torch.cuda.synchronize()
      n1_ms = 1e3 * (time.perf_counter() - t0) / a.steps
    barrier()

  comms, alt_default = None, None
  if ddp_on:
    # The contract's timed region runs on the SIMPLEST data plane - the uncapped root communicator (a plain ncclCommInitRank), ncclAllReduce,
    # no CU reserve: the path least likely to fail in a first multi-GPU contact, so that whatever happens afterwards (ncclCommSplit, capped
    # collectives, the other algorithm) a measurement exists.  The capped communicators are created and every alternative - including the
    # caps 8 / 16 (and PLM_COMM_CUS if set) with the tail bucket on either communicator - is measured afterwards, under a watchdog.
    # PLM_BENCH_FIRST_ALT='algo,cap,tail' (e.g. 'allreduce,16,1'; tools/ddp_whatif.sh) times another data plane first - a what-if knob, the
    # driver's run never sets it.
    comms = ddp.make_comm_set(device, a.comm, caps=[])
    alt_default = {'algo': os.environ.get('PLM_COMM_ALGO') or 'allreduce', 'cap': 0, 'tail': False}
    first_alt = os.environ.get('PLM_BENCH_FIRST_ALT')
    if first_alt:
      f_algo, f_cap, f_tail = (first_alt.split(',') + ['0', '0'])[:3]
      alt_default = {'algo': f_algo, 'cap': int(f_cap), 'tail': f_tail.strip().lower() in ('1', 'true', 'tail')}
      ddp.add_capped_comms(comms, [alt_default['cap']])
    c0, c0_tail, c0_reserve = ddp.pick_comms(comms, cap=alt_default['cap'], tail=alt_default['tail'])
    reducer = ddp.GradReducer(flat, params, model._grad_spans, c0, bucket_cap_mb=a.bucket_mb, force=bool(force_reducer),
                              reserve_cus=c0_reserve, writers=model.grad_writers(), comm_tail=c0_tail, groups=model.grad_groups(),
                              algo=alt_default['algo'])
    reducer.broadcast_params([p.data for p in params])
    model.sink.on_ready = reducer.param_ready
    model.sink.on_queued = reducer.param_queued
    model.sink.resolve_queue_budget(device, agreed=ddp.agree_min(model.sink.resolve_queue_budget(device)))
    st['reducer'] = reducer

  def apply_alt(alt):
    ddp.apply_alternative(st['reducer'], comms, alt)

  # ---- the timed region of the contract (for a data-parallel run: on the DEFAULT data plane)
  # (data-parallel: the one-off window freeze - a host wait on earlier steps' events + a control-plane all-reduce at the top of step
  # FREEZE_AFTER + 1 - must land in the warm-up, whatever --warmup says)
  warm = max(a.warmup, ddp.FREEZE_AFTER + 1) if ddp_on else a.warmup
  elapsed, per_rank, last_loss = timed(warm, a.steps)
  fpt = flops_per_token(c, hidden, pairs_per_token)

  def line(elapsed, last_loss):
    ms_per_step = 1e3 * elapsed / a.steps
    value = B * T * world / (elapsed / a.steps)
    o = {
      'metric': f'tokens/sec fwd+bwd ({a.config.upper()}, seq={T}, bf16)', 'value': round(value, 1), 'unit': 'tokens/s',
      'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
      'config': {'workload': f'plainLM {a.config} decoder ({c["n_layers"]}L d={c["d_model"]} nh={c["n_heads"]} h={hidden} V={V}) '
                             f'fwd+bwd incl. per-step weight casts' + (' + bucketed RCCL grad all-reduce' if world > 1 else '')
                             + (', document-boundary masks (mean doc length 256)' if a.doc_mask else ''),
                 'global_batch': B * world, 'seq_len': T, 'parallelism': f'dp{world}'},
      'tokens_per_sec_per_gpu': round(value / world, 1),
      'mfu_bf16': round(value / world * fpt / (PEAK_BF16_TFLOPS * 1e12), 4),
      'flops_per_token': fpt, 'loss': round(last_loss, 4),
    }
    if pairs_per_token is not None:  # what the same rate would be called if the skipped (masked-out) pairs were counted as work
      o['mfu_bf16_causal_counted'] = round(value / world * flops_per_token(c, hidden) / (PEAK_BF16_TFLOPS * 1e12), 4)
      o['attn_pairs_per_token'] = round(pairs_per_token, 1)
    if a.single_device:
      o['data'] = 'synthetic; PLUMBING CHECK ONLY: all ranks share cuda:0, gradients over gloo - not a throughput measurement'
    return o

  def comm_info(alt, per_rank, extra=None, stats=None):
    reducer = st['reducer']
    info = {'backend': reducer.comm.backend, 'ranks': reducer.comm.world_size, 'buckets': len(reducer.buckets),
            'bucket_cap_mb': a.bucket_mb, 'selected': alt, 'cu_reserve': reducer.reserve_cus,
            'reserve_policy': 'only inside a window of estimated GPU time behind each bucket launch; window = the bucket\'s measured collective, '
                              'frozen + agreed over the ranks after %d steps (PLM_COMM_WINDOWS)' % ddp.FREEZE_AFTER
                              + (f'; here modelled at {reducer.model_gbps} GB/s' if reducer.model_gbps else ''),
            'max_ctas_overlapped_buckets': getattr(reducer.comm, 'max_ctas', None),
            'tail_communicator': 'uncapped root' if reducer.comm_tail is not None else 'none (tail bucket on the same communicator)',
            'nccl_max_nchannels_env': os.environ.get('NCCL_MAX_NCHANNELS'),
            'rank_ms_per_step': {'min': round(1e3 * min(per_rank) / a.steps, 3), 'max': round(1e3 * max(per_rank) / a.steps, 3)},
            'n1_ms_per_step': round(n1_ms, 3) if n1_ms is not None else None,
            'n1_note': 'rank 0 alone (no data plane, same process, before any communicator existed), same warm-up / steps'}
    info.update(stats or reducer.stats())  # of the timed run: bucket_ms, exposed_comm_ms (the join wait of its last step), clock scale, share of MFMA launches that ran reserved
    if extra:
      info.update(extra)
    return info

  out = line(elapsed, last_loss)
  run_stats = st['reducer'].stats() if ddp_on else None  # before any re-configuration resets what the reducer has learned
  if ddp_on:
    # (n1_ms lives on rank 0 only - the rank that prints; no collective here: the other ranks never measured it)
    out['comm'] = comm_info(alt_default, per_rank, stats=run_stats)

  def bail(line_out, deadline_t):
    """Leave from a failure path of an untimed leg: rank 0 prints `line_out` (a complete line of a finished timed region) and exits 0; the other
    ranks exit 3 (the failure stays visible to the launcher) - but only after rank 0 has had its chance: a launcher (torch.distributed.run,
    spawn_ranks) kills every rank as soon as one exits non-zero, and rank 0 may learn of the failure only from its own watchdog, at
    `deadline_t` (time.monotonic() of the armed watchdog; None: the watchdogs themselves are firing, all ranks within moments)."""
    if rank == 0:
      for attempt in range(5):
        try:
          text = json.dumps(line_out)
          break
        except RuntimeError:  # a watchdog thread serialising the dict while the main thread adds a key
          time.sleep(0.02)
      os.write(json_fd, (text + '\n').encode())
      os._exit(0)
    grace = float(os.environ.get('PLM_BENCH_EXIT_GRACE', '5'))
    time.sleep(max(0.0, (deadline_t or 0.0) - time.monotonic()) + grace)
    os._exit(3)

  # ---- data-parallel autotune (untimed): capped communicators (ncclCommSplit, maxCTAs 8 / 16), then every alternative of {all-reduce,
  # reduce-scatter + all-gather} x {no reserve, 8, 16 CUs} x {tail bucket on the capped communicator, on the uncapped root} for a few steps
  # each; the ranks agree on the winner (the alternative whose SLOWEST rank is fastest); if it is not the first one it is then timed over the
  # full `steps`, and the better of the two full timed regions is the line's value (both are reported).  A watchdog prints the first
  # region's line if any of this hangs, an exception does the same.
  if ddp_on and not a.no_autotune and os.environ.get('PLM_BENCH_AUTOTUNE', '1') != '0':
    import threading
    deadline = float(os.environ.get('PLM_BENCH_AUTOTUNE_TIMEOUT', '300'))
    first_out = out  # the first timed region's line: what every failure path below prints (never a half-built later dict)
    def give_up():
      first_out['comm']['autotune'] = f'did not finish within {deadline:.0f} s: this line is the timed run on the simplest data plane'
      first_out['autotune_ok'] = False
      bail(first_out, None)  # rank 0 prints a valid line; the others report the failure through their exit code
    dog = threading.Timer(deadline, give_up)
    dog.daemon = True
    dog_t = time.monotonic() + deadline
    dog.start()
    try:
      n_try = int(os.environ.get('PLM_BENCH_AUTOTUNE_STEPS', '4'))

      def run_steps(n):
        t0 = time.perf_counter()
        for i in range(n):
          fwd_bwd(i)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

      alts, agreed, win = ddp.autotune(st['reducer'], comms, run_steps, first=alt_default, n_try=n_try, caps=sorted({8, 16, ddp.COMM_CUS} - {0}))
      table = [dict(alt, ms_per_step=round(ms, 3)) for alt, ms in zip(alts, agreed)]
      runs = [dict(alt_default, ms_per_step=out['ms_per_step'], role='first timed region (simplest data plane)')]
      chosen = alt_default
      if alts[win] != alt_default and agreed[win] < 0.995 * agreed[alts.index(alt_default)]:
        apply_alt(alts[win])
        e2, pr2, l2 = timed(max(2, ddp.FREEZE_AFTER + 1), a.steps)
        runs.append(dict(alts[win], ms_per_step=round(1e3 * e2 / a.steps, 3), role='autotune winner'))
        if e2 < elapsed:
          elapsed, per_rank, last_loss, chosen = e2, pr2, l2, alts[win]
          out = line(elapsed, last_loss)
          run_stats = st['reducer'].stats()
      if chosen == alt_default:
        apply_alt(alt_default)  # the untimed legs below run on the selected data plane
      out['comm'] = comm_info(chosen, per_rank, {'alternatives': table, 'timed_runs': runs,
                                                 'autotune': f'{len(alts)} alternatives x {n_try} steps (max over ranks, agreed); value = the better of the full timed runs'},
                              stats=run_stats)
    except Exception as e:  # noqa: BLE001 - the first timed region's line must survive whatever the alternatives do
      import traceback
      traceback.print_exc()
      first_out['comm']['autotune'] = f'failed on rank {rank} ({type(e).__name__}: {e}): this line is the timed run on the simplest data plane'
      first_out['autotune_ok'] = False
      bail(first_out, dog_t)  # the other ranks may be inside a collective: no clean shutdown to wait for
    dog.cancel()
  ms_per_step = out['ms_per_step']
  reducer = st['reducer']

  # ---- untimed legs.  The contract's measurement exists at this point: whatever happens below (an exception, or - data-parallel runs - a
  # collective of a leg that never returns) rank 0 still prints the line, with what the legs have added so far and an `extras_error` note.
  extras_dog, extras_t = None, None
  def extras_failed(why, deadline_t):
    out['extras_error'] = why
    bail(out, deadline_t)
  if not a.no_extras and world > 1:
    import threading
    extras_deadline = float(os.environ.get('PLM_BENCH_EXTRAS_TIMEOUT', '300'))
    extras_dog = threading.Timer(extras_deadline, lambda: extras_failed(f'the untimed legs did not finish within {extras_deadline:.0f} s', None))
    extras_dog.daemon = True
    extras_t = time.monotonic() + extras_deadline
    extras_dog.start()
  try:
    if not a.no_extras:
      # ---- roofline leg: identical steps with HIP events around every MFMA kernel launch (untimed) ----
      ops.PROFILE = []
      n_prof = 3
      for i in range(n_prof):
        fwd_bwd(i)
      torch.cuda.synchronize()
      fam = {}
      for name, fl, s, e in ops.PROFILE:
        acc = fam.setdefault(name, [0.0, 0.0, 0])
        acc[0] += fl
        acc[1] += s.elapsed_time(e)
        acc[2] += 1
      ops.PROFILE = None
      fams = {k: {'TFLOP/s': round(v[0] / v[1] / 1e9, 1), 'ms_per_step': round(v[1] / n_prof, 3), 'launches_per_step': v[2] // n_prof,
                  'avg_launch_ms': round(v[1] / v[2], 4)} for k, v in fam.items() if not k.startswith('hbm:')}
      # the HBM-bound kernels (north_star: achieved HBM GB/s on the norm / activation kernels): algorithmic bytes of SURVEY.md section 8d
      # (12 M d / 16 M d for the norms with their fused adds, 4 M V for cross-entropy in place, 8 B per parameter for the weight casts) over
      # HIP-event time of the same untimed replay
      hbm = {k[4:]: {'GB/s': round(v[0] / v[1] / 1e6, 1), 'frac_of_8TBps': round(v[0] / v[1] / 1e6 / PEAK_HBM_GBPS, 4),
                     'ms_per_step': round(v[1] / n_prof, 3), 'launches_per_step': v[2] // n_prof, 'avg_launch_ms': round(v[1] / v[2], 4),
                     'algorithmic_MB_per_launch': round(v[0] / v[2] / 1e6, 1)} for k, v in fam.items() if k.startswith('hbm:')}
      dom = max(fams, key=lambda k: fams[k]['ms_per_step'])
      out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': fams[dom]['TFLOP/s'], 'peak': PEAK_BF16_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': round(fams[dom]['TFLOP/s'] / PEAK_BF16_TFLOPS, 4), 'traffic': None,
                         'families': fams, 'hbm': hbm,
                         'families_note': 'gemm_nt = plain NT GEMM launches; gemm_nt_fused = NT launches whose epilogue also does RoPE / SwiGLU / '
                                          'SwiGLU backward (TFLOP/s counts the GEMM flops only, the time includes the fused pass)'}
      out['roofline'].update(pmc_traffic(dom, a.config, B * T, c['n_layers']))

      # ---- full training step (clip + AdamW) for reference, same data (untimed leg) ----
      if rank == 0 or world > 1:
        # the engine's optimizer tail (plainlm_amd/optim.py): ||g|| reduction + clip folded into one fused AdamW launch per
        # weight-decay group, on flat buffers.  NOTE: it re-lays parameters/gradients, so it runs after the legs above.
        import plainlm_amd as P
        from plainlm_amd.optim import FlatAdamW
        opt = FlatAdamW(model, P.get_param_groups(model, 0.1), lr=1e-4, betas=[0.9, 0.95], eps=1e-8, weight_decay=0.1)
        if reducer is not None:  # FlatAdamW has re-laid the gradient spans: a new bucket plan on the same (selected) data plane
          reducer = ddp.GradReducer(flat, params, model._grad_spans, reducer.comm, bucket_cap_mb=a.bucket_mb, force=reducer.force,
                                    reserve_cus=reducer.reserve_cus, writers=model.grad_writers(), comm_tail=reducer.comm_tail,
                                    groups=model.grad_groups(), algo=reducer.algo)
          model.sink.on_ready = reducer.param_ready
          model.sink.on_queued = reducer.param_queued
          st['reducer'] = reducer

        def full(i):
          fwd_bwd(i, recast=not opt.emits_shadows)  # FlatAdamW writes the bf16 shadows of the weights it has just updated
          opt.clip_and_step(1.0)

        for i in range(2):
          full(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nfull = max(3, a.steps // 4)
        for i in range(nfull):
          full(i)
        torch.cuda.synchronize()
        full_ms = 1e3 * (time.perf_counter() - t1) / nfull
        out['full_step'] = {'ms_per_step': round(full_ms, 3), 'tokens_per_sec_per_gpu': round(B * T / full_ms * 1e3, 1),
                            'over_fwd_bwd_ms': round(full_ms - ms_per_step, 3),
                            'note': 'fwd+bwd + global-norm clip + AdamW (plainlm_amd FlatAdamW kernels; rank-local clock, untimed leg)'
                                    + ('; the AdamW launch also emits the bf16 weight shadows, so this leg has no stand-alone weight cast'
                                       if opt.emits_shadows else '')}

      if world == 1:
        out['cpu_baseline'] = cpu_baseline(c)
  except Exception as e:  # noqa: BLE001 - the timed region's line must survive the untimed legs
    import traceback
    traceback.print_exc()
    extras_failed(f'untimed leg failed on rank {rank} ({type(e).__name__}: {e})', extras_t)
  if extras_dog is not None:
    extras_dog.cancel()

  if rank == 0:
    os.write(json_fd, (json.dumps(out) + '\n').encode())
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()

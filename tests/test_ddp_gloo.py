"""world_size-2 tests of the gradient reducer on CPU (gloo), gradients supplied by the oracle.

Invariants: (i) W ranks on W shards == 1 rank with grad accumulation W on the
concatenated shards; (ii) all ranks hold identical reduced gradients / broadcast parameters;
(iii) bucketed reduction == one flat all-reduce; non-final accumulation micro-steps do not communicate."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref as O
from plainlm_amd import ddp

CFG = dict(vocab_size=64, seq_len=16, dim=64, n_layers=1, n_heads=1)


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, out_dir, tied=False):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(1)
  ocfg = O.OracleConfig(**CFG, tie_embeddings=tied)
  names = O.param_names(ocfg)
  shapes = O.param_shapes(ocfg)
  # rank-dependent init, then broadcast from rank 0 (DDP ctor semantics)
  params = O.init_params(ocfg, seed=rank)
  spans, off = [], 0
  for n in names:
    k = int(np.prod(shapes[n]))
    spans.append((off, k))
    off += k
  flat_p = torch.cat([params[n].reshape(-1) for n in names])
  flat_g = torch.zeros(off)
  plist = [flat_p[o:o + k] for o, k in spans]
  comm, comm_tail = ddp.make_comm('cpu', 'torch')
  assert comm_tail is None  # torch.distributed has no per-communicator cap: one communicator for everything
  # tied embeddings: the shared weight (parameter 0) is written twice per backward - lm_head's dW first, the embedding
  # scatter last (functional.HeadLossFn / EmbedFn) - and its bucket may only go out after the second write
  writers = {id(plist[0]): 2} if tied else None
  red = ddp.GradReducer(flat_g, plist, spans, comm, bucket_cap_mb=0.01, writers=writers)  # ~2.6k floats per bucket: many buckets
  assert len(red.buckets) > 3
  red.broadcast_params(flat_p)
  params = {n: flat_p[o:o + k].view(shapes[n]) for n, (o, k) in zip(names, spans)}

  rng = np.random.default_rng(5)
  tok = torch.from_numpy(rng.integers(0, CFG['vocab_size'], size=(2 * world, 2, CFG['seq_len'] + 1)))
  accum = 2
  calls = {'n': 0}
  orig = comm.allreduce_avg_

  def counting(span, stream=None, algo=None):
    calls['n'] += 1
    return orig(span, stream)

  comm.allreduce_avg_ = counting
  for micro in range(accum):
    batch = tok[micro * world + rank]  # rows r, r+W, ... of each micro-step
    _, g = O.loss_and_grads(params, ocfg, batch[:, :-1], batch[:, 1:], scale=1.0 / accum)
    red.begin(sync=(micro == accum - 1))
    if tied:  # the head's share of the shared weight's gradient comes first in backward
      o, k = spans[0]
      half = 0.5 * g[names[0]].reshape(-1)
      if micro == 0:
        flat_g[o:o + k] = half
      else:
        flat_g[o:o + k] += half
      red.param_ready(plist[0])
      assert calls['n'] == 0  # one of two writers has reported: the bucket must not be reduced yet
    # gradients become ready last-parameter-first, like backward
    for i in range(len(names) - 1, -1, -1):
      o, k = spans[i]
      if tied and i == 0:
        flat_g[o:o + k] += half
      elif micro == 0:
        flat_g[o:o + k] = g[names[i]].reshape(-1)
      else:
        flat_g[o:o + k] += g[names[i]].reshape(-1)
      red.param_ready(plist[i])
    if micro < accum - 1:
      assert calls['n'] == 0  # no communication on non-final accumulation micro-steps
    red.finish()
  assert calls['n'] == len(red.buckets)
  if tied:  # a third report for a parameter with two declared writers is a wiring bug and must be loud
    red.begin(sync=True)
    red.param_ready(plist[0]); red.param_ready(plist[0])
    with pytest.raises(RuntimeError, match='more gradient writes'):
      red.param_ready(plist[0])
    red.sync = False
  # collective agreement used by make_comm's RCCL -> torch fallback: one failing rank makes every rank fall back
  assert ddp.all_ranks_ok(True) is True
  assert ddp.all_ranks_ok(rank != 1) is False
  # autotune consensus (bench.py, world > 1): the ranks time the alternatives differently - rank r finds alternative r fastest - and
  # must still pick the SAME winner: the alternative whose slowest rank is fastest (ties: lowest index)
  local = [5.0, 5.0, 5.0, 4.0 + 0.01 * rank, 9.0]
  local[rank % 3] = 1.0
  win, agreed = ddp.agree_winner(local)
  assert win == 3 and agreed[3] == pytest.approx(4.0 + 0.01 * (world - 1)) and agreed[4] == 9.0, (win, agreed)
  win, agreed = ddp.agree_winner([2.0 + rank, 2.0 + (world - 1 - rank), 7.0])
  assert win == 0 and agreed[:2] == [2.0 + world - 1] * 2  # a tie goes to the lowest index on every rank
  # the one-off agreement on the reserve windows ('frozen' mode): the maximum over the ranks, elementwise
  assert ddp.agree_max_floats([0.001 * (rank + 1), 1.0 - 0.1 * rank]) == pytest.approx([0.001 * world, 1.0])
  # bench.py's autotune switches the data plane between two steps (GradReducer.configure): same buckets, same result
  if not tied:
    before = flat_g.clone()
    red.configure(comm=comm, comm_tail=None, reserve_cus=0, algo='rsag')  # torch.distributed data plane: the algorithm is RCCL-direct only
    with pytest.raises(RuntimeError, match='inside a step'):
      red.begin(sync=True)
      red.configure(algo='allreduce')
    red.sync = False
    flat_g.copy_(before * (rank + 1))  # rank-dependent gradients: the mean over ranks is before * (world + 1) / 2
    n0 = calls['n']
    red.begin(sync=True)
    for i in range(len(names) - 1, -1, -1):
      red.param_ready(plist[i])
    red.finish()
    assert calls['n'] - n0 == len(red.buckets)
    assert torch.allclose(flat_g, before * (world + 1) / 2, rtol=1e-6, atol=1e-9)
    flat_g.copy_(before)
    # ddp.autotune - the loop bench.py --gpus N runs after its first timed region: every alternative gets FREEZE_AFTER + 1 untimed and n_try
    # timed steps through the reducer (real collectives here), and the ranks leave with the SAME winner although each prefers another one
    calls_rs, fake = {'n': 0}, {0: (5.0, 1.0, 0.5, 0.2), 1: (3.0, 3.5, 3.2, 3.4), 2: (2.0, 4.0, 3.9, 3.0)}  # alternative -> ms per step seen by rank 0, 1, 2, 3

    def run_steps(n):
      for _ in range(n):
        red.begin(sync=True)
        for i in range(len(names) - 1, -1, -1):
          red.param_ready(plist[i])
        red.finish()
      k = calls_rs['n']
      calls_rs['n'] += 1
      return n * 1e-3 * fake[k // 2][rank]  # calls 2a and 2a + 1 belong to alternative a (warm-up, timed)

    comms = {0: comm}
    first = {'algo': 'allreduce', 'cap': 0, 'tail': False}
    n0 = calls['n']
    alts, agreed, win = ddp.autotune(red, comms, run_steps, first=first, n_try=2)
    assert alts == [first, {'algo': 'allreduce', 'cap': 8, 'tail': False}, {'algo': 'allreduce', 'cap': 16, 'tail': False}]  # no caps / algorithms without direct RCCL
    want = [max(fake[a_][:world]) for a_ in range(3)]
    assert agreed == pytest.approx(want) and win == want.index(min(want)) == 1, (agreed, win)
    assert calls_rs['n'] == 6 and calls['n'] - n0 == 3 * (ddp.FREEZE_AFTER + 1 + 2) * len(red.buckets)
    ddp.apply_alternative(red, comms, alts[win])
    assert red.comm is comm and red.comm_tail is None and red.algo == 'allreduce'
  if not tied:
    # A capped communicator whose ncclCommSplit fails on ONE rank must be dropped on EVERY rank (ddp.add_capped_comms agrees on the control
    # plane): ranks that disagreed about which communicators exist would enter different collectives.  The direct-RCCL root is faked (no GPU
    # here): its split() fails for cap 8 on rank 1 only; every rank must end with {0, 16}, the children created for the lost cap closed, and
    # the alternatives of the autotune table must then carry cap 8 without its tail variant (no child: root + GEMM-side reserve).
    class FakeRccl(ddp.RcclComm):
      closed = []

      def __init__(self, r, cap=0):  # no library, no handle
        self.rank, self.world_size, self.max_ctas, self.backend = r, world, cap, 'rccl-direct'

      def split(self, max_ctas=0):
        if max_ctas == 8 and self.rank == 1:
          raise RuntimeError('simulated ncclCommSplit failure')
        return FakeRccl(self.rank, max_ctas)

      def close(self):
        FakeRccl.closed.append(self.max_ctas)

    fake_set = ddp.add_capped_comms({0: FakeRccl(rank)}, (8, 16))
    assert sorted(fake_set) == [0, 16] and fake_set[16].max_ctas == 16, sorted(fake_set)
    assert FakeRccl.closed == ([] if rank == 1 else [8])  # the ranks whose split had succeeded gave their child back
    alts = ddp.data_plane_alternatives(fake_set)
    assert {'algo': 'allreduce', 'cap': 8, 'tail': False} in alts and {'algo': 'allreduce', 'cap': 8, 'tail': True} not in alts
    assert {'algo': 'rsag', 'cap': 16, 'tail': True} in alts
    assert ddp.pick_comms(fake_set) == (fake_set[0], None, 0)  # the engine's default first contact: root, no tail communicator, no reserve
    assert ddp.pick_comms(fake_set, cap=16) == (fake_set[16], fake_set[0], 16) and ddp.pick_comms(fake_set, cap=8) == (fake_set[0], None, 8)
  torch.save({'params': flat_p.clone(), 'grads': flat_g.clone()}, os.path.join(out_dir, f'r{rank}.pt'))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('tied', [False, True])
def test_two_rank_bucketed_allreduce_equals_accumulation(tmp_path, tied):
  _bucketed_allreduce_equals_accumulation(tmp_path, 2, tied)


@pytest.mark.timeout(600)
def test_four_rank_bucketed_allreduce_equals_accumulation(tmp_path):
  """The same contract at world_size 4 (the row split r::W, the mean over W x accum micro-batches, the agreement helpers)."""
  _bucketed_allreduce_equals_accumulation(tmp_path, 4, False)


def _bucketed_allreduce_equals_accumulation(tmp_path, world, tied):
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), tied), nprocs=world, join=True)
  r = [torch.load(tmp_path / f'r{i}.pt') for i in range(world)]
  # (ii) identical params and reduced grads on every rank
  for i in range(1, world):
    assert torch.equal(r[0]['params'], r[i]['params'])
    assert torch.equal(r[0]['grads'], r[i]['grads'])
  # (i)+(iii) single process, accumulation over all W*accum micro-batches, one flat mean
  ocfg = O.OracleConfig(**CFG, tie_embeddings=tied)
  names = O.param_names(ocfg)
  params = O.init_params(ocfg, seed=0)  # rank 0's init was broadcast
  assert torch.equal(r[0]['params'], torch.cat([params[n].reshape(-1) for n in names]))
  rng = np.random.default_rng(5)
  tok = torch.from_numpy(rng.integers(0, CFG['vocab_size'], size=(2 * world, 2, CFG['seq_len'] + 1)))
  tot = None
  for i in range(2 * world):
    _, g = O.loss_and_grads(params, ocfg, tok[i][:, :-1], tok[i][:, 1:], scale=1.0 / (2 * world))
    flat = torch.cat([g[n].reshape(-1) for n in names])
    tot = flat if tot is None else tot + flat
  err = (r[0]['grads'] - tot).abs().max().item()
  assert err <= 1e-6 * tot.abs().max().item() + 1e-9, err

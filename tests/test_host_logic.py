"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side
mirror of the reference's module API has the right names/shapes/init/param groups, the product
path refuses to run without a GPU, schedules and doc_start match the oracle."""

import json
import os
from collections import namedtuple

import numpy as np
import pytest
import torch

import plainlm_amd as P
from plainlm_amd import _lib, ddp, engine
from oracle import cpu_ref as O


def test_library_exports_every_header_symbol():
  if not os.path.exists(_lib.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  lib = _lib.load()
  names = _lib.header_functions()
  assert len(names) >= 25
  assert set(names) == set(_lib.SIGNATURES)
  for n in names:
    assert hasattr(lib, n), n
  assert lib.plm_version() >= 100


def _cfg(**over):
  EC = dict(model='transformer', vocab_size=512, seq_len=128, d_model=128, expand='8/3', n_layers=2, n_heads=2,
            mlp_class='glu', tie_embeddings=False)
  EC.update(over)
  return namedtuple('Config', EC.keys())(**EC)


def test_construct_model_names_shapes_init():
  torch.manual_seed(0)
  model, mcfg = P.construct_model(_cfg())
  ocfg = O.OracleConfig(vocab_size=512, seq_len=128, dim=128, n_layers=2, n_heads=2)
  names = [n for n, _ in model.named_parameters()]
  assert names == O.param_names(ocfg)
  assert {n: tuple(p.shape) for n, p in model.named_parameters()} == O.param_shapes(ocfg)
  assert list(model.state_dict()) == names  # no extra buffers (RoPE table is a plain attribute like the reference's)
  assert mcfg.expand == 8 / 3 and model.layers[0].mlp.hidden_dim == 512
  for n, p in model.named_parameters():
    assert p.dtype == torch.float32
    if 'norm' in n:
      assert torch.equal(p.detach(), torch.ones_like(p))
    else:
      want = 0.02 / np.sqrt(4) if (n.endswith('fc2.weight') or n.endswith('w_out.weight')) else 0.02
      assert abs(p.std().item() - want) < 0.05 * want, n
  assert model.count_params(False) == sum(p.numel() for p in model.parameters())
  assert model.count_params(True) == model.count_params(False) - 2 * 512 * 128


def test_golden_init_statistics(golden_dir):
  z = np.load(os.path.join(golden_dir, 'model.npz'))
  torch.manual_seed(1)
  m = P.Transformer(P.ModelConfig(vocab_size=256, seq_len=64, dim=128, expand=8 / 3, n_layers=2, n_heads=2, mlp='glu'))
  for i, (n, p) in enumerate(m.named_parameters()):
    ref = float(z['init_std'][i])
    got = p.std().item() if p.numel() > 1 and 'norm' not in n else 0.0
    assert abs(got - ref) <= 0.06 * ref + 1e-12, (n, got, ref)
  m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w:')})  # reference checkpoint layout


def test_tied_embeddings_and_param_groups():
  model, _ = P.construct_model(_cfg(tie_embeddings=True))
  assert model.lm_head.weight is model.embed_tokens.weight
  assert model.count_params(True) == model.count_params(False) - 512 * 128
  model, _ = P.construct_model(_cfg())
  groups = P.get_param_groups(model, 0.1)
  assert groups[0]['weight_decay'] == 0.1 and groups[1]['weight_decay'] == 0.0
  no_decay = {id(p) for p in groups[1]['params']}
  for n, p in model.named_parameters():
    assert (id(p) in no_decay) == ('norm' in n), n
  assert len(groups[0]['params']) == 10 and len(groups[1]['params']) == 5


def test_unsupported_configs_fail_loudly():
  with pytest.raises(NotImplementedError):
    P.construct_model(_cfg(model='pythia-160m'))
  with pytest.raises(NotImplementedError):
    P.construct_model(_cfg(mlp_class='moe'))
  for kind in ('mlp', 'mlp_relu_sq'):  # the reference's other two MLP classes are built (models/transformer.py:26): one fc1 projection of h rows
    m, _ = P.construct_model(_cfg(mlp_class=kind))
    assert tuple(m.layers[0].mlp.fc1.weight.shape) == (512, 128) and tuple(m.layers[0].mlp.fc2.weight.shape) == (128, 512)
  with pytest.raises(NotImplementedError):
    P.construct_model(_cfg(n_heads=8))  # head_dim 16: below the 32 / 64 / 128 the attention kernels are built for
  assert P.construct_model(_cfg(n_heads=4))[0].head_dim == 32 and P.construct_model(_cfg(n_heads=1))[0].head_dim == 128
  with pytest.raises(ValueError):
    P.construct_model(_cfg(d_model=100, n_heads=3))
  with pytest.raises(NotImplementedError, match='multiple of 8'):
    P.construct_model(_cfg(vocab_size=50257))  # GPT-2's vocabulary unpadded: refused at construction, with the remedy in the message


def test_no_cpu_fallback():
  model, _ = P.construct_model(_cfg())
  ids = torch.zeros(1, 128, dtype=torch.int64)
  with pytest.raises(RuntimeError, match='MI355X'):
    model(ids, None)
  with pytest.raises(RuntimeError):
    model.loss(ids, ids)
  from plainlm_amd import ops
  with pytest.raises(RuntimeError, match='no CPU path'):
    ops.rmsnorm_fwd(torch.zeros(4, 128), torch.ones(128), 1e-6)
  ecfg = namedtuple('C', ['seq_len', 'grad_accumulation_steps', 'grad_clip', 'dtype'])(128, 1, 1.0, 'bfloat16')
  with pytest.raises(RuntimeError, match='no CPU path'):
    P.TorchEngine(model, ecfg, 'cpu', None, None)


def test_doc_start_and_mask_conversion():
  docs = [[10, 20, 35], [1, 31, 32, 1]]
  ds = engine.doc_start_from_lengths(docs, 64)
  assert torch.equal(ds, O.doc_start_from_lengths(docs, 64))
  got = P.Transformer._doc_start(O.mask_from_doc_start(ds), 2, 64)
  assert torch.equal(got, ds)
  assert P.Transformer._doc_start(None, 2, 64) is None
  with pytest.raises(ValueError):
    engine.doc_start_from_lengths([[3, 3]], 64)
  with pytest.raises(TypeError):
    P.Transformer._doc_start(torch.zeros(2, 64, 64), 2, 64)


def test_rope_tables_match_oracle():
  from plainlm_amd.transformer import rope_tables
  c, s = rope_tables(64, 2048)
  oc, os_ = O.rope_table(64, 2048)
  assert torch.equal(c, oc) and torch.equal(s, os_)


def test_lr_schedule_matches_oracle():
  p = torch.nn.Parameter(torch.zeros(1))
  opt = torch.optim.SGD([p], lr=1.0)
  sch = engine.WarmupCosine(opt, 0.0, 3e-3, 1e-5, 2, 8)
  assert opt.param_groups[0]['lr'] == 0.0
  for t in range(1, 12):
    sch.step()
    assert opt.param_groups[0]['lr'] == O.warmup_cosine_lr(t, 0.0, 3e-3, 1e-5, 2, 8)
  st = sch.state_dict()
  assert 'optimizer' not in st and st['iter'] == 11
  cfg = namedtuple('C', ['scheduler', 'warmup_steps', 'steps_budget', 'lr_end', 'lr_end_pct', 'lr_start', 'lr'])(
    'warmup_cosine', 0.1, 4800, 1e-5, None, 0.0, 3e-3)
  s2 = engine.initialize_scheduler(opt, cfg)
  assert s2.warmup_steps == 480 and s2.T == 4800


def test_bucket_plan():
  spans, off = [], 0
  for n in [1000, 10, 300, 300, 10, 5000]:
    spans.append((off, n))
    off += n
  b = ddp.plan_buckets(spans, cap_bytes=4 * 700)
  # a permuted span table (FlatAdamW groups parameters by weight decay) still yields contiguous buckets
  perm = [spans[i] for i in (2, 0, 5, 1, 4, 3)]
  for lo, hi, idxs in ddp.plan_buckets(perm, cap_bytes=4 * 700):
    assert hi - lo == sum(perm[i][1] for i in idxs)
  # last parameter first; oversize parameters alone; every bucket contiguous; full cover, no overlap
  assert b[0][2] == [5] and b[-1][2] == [0]
  covered = sorted((lo, hi) for lo, hi, _ in b)
  assert covered[0][0] == 0 and covered[-1][1] == off
  for (l0, h0), (l1, h1) in zip(covered, covered[1:]):
    assert h0 == l1
  for lo, hi, idxs in b:
    assert (hi - lo) * 4 <= 4 * 700 or len(idxs) == 1


def test_bench_metric_definition(monkeypatch):
  """bench.py's work model is SURVEY §8(d)'s: 797 976 576 FLOP per token for the 160M config at T = 1024 (causal-counted
  attention, fwd+bwd = 3x fwd), 2 460 745 728 for 420M at T = 2048; the committed PMC profile feeds roofline.traffic."""
  import bench
  c = bench.CONFIGS['160m']
  assert bench.flops_per_token(c, 2048) == 797_976_576
  c = bench.CONFIGS['420m']
  assert bench.flops_per_token(c, 2816) == 2_460_745_728
  # roofline.traffic comes from the committed PMC profile ONLY when that profile was taken on this very tree (it records
  # the sha of plainlm_amd/csrc); otherwise bench.py says why it reports null
  tr = bench.pmc_traffic('gemm_nt', '160m', 32768, 12)
  if 'traffic' in tr:
    assert tr['traffic'] > tr['algorithmic_bytes'] > 0 and tr['csrc_sha'] == bench.csrc_sha()
    assert os.path.exists(os.path.join(os.path.dirname(bench.__file__), tr['traffic_source']))
  else:
    assert 'traffic_note' in tr
  assert bench.pmc_traffic('gemm_nt', '420m', 16384, 24) == {}
  assert len(bench.csrc_sha()) == 16 and bench.physical_cores() >= 1
  # the launch plan bench.py sums over must match the rows of the COMMITTED profile whatever this tree's hash is (a plan / row-name
  # mismatch once took the whole benchmark down on the GPU box, where the hashes did match)
  prof = bench._pmc_profile()  # the newest committed profile (the one of this tree's csrc when there is one)
  if os.path.exists(prof):
    monkeypatch.setattr(bench, 'csrc_sha', lambda: json.load(open(prof))['csrc_sha'])
    for fam in ('gemm_nt', 'gemm_nt_fused', 'gemm_tn'):
      tr = bench.pmc_traffic(fam, '160m', 32768, 12)
      assert tr.get('traffic', 0) > tr.get('algorithmic_bytes', 1) > 0, (fam, tr)


def test_bucket_plan_160m_engine_layout():
  """FlatAdamW's placement at the 160M size: [25 norm weights | embed_tokens | 12 x (w_qkv, w_out, fc1, fc2) | lm_head].
  Walking it from the end (the order gradients become ready): lm_head alone goes first, the layers follow in <= 64 MiB
  buckets, and the LAST collective carries embed_tokens together with the 77 KB of norm weights (complete only when
  layer 0 has been differentiated) - one tail collective, not two; every bucket is one contiguous span."""
  d, h, V, L = 768, 2048, 50280, 12
  sizes = [d] * (2 * L + 1) + [V * d] + [3 * d * d, d * d, 2 * h * d, d * h] * L + [V * d]
  spans, off = [], 0
  for n in sizes:
    spans.append((off, n))
    off += n
  assert off == 162_183_936  # SURVEY A12 [probed]
  b = ddp.plan_buckets(spans, cap_bytes=64 << 20)
  n_norm = 2 * L + 1
  assert b[0][2] == [len(sizes) - 1]                               # lm_head: first to be reduced, alone
  assert b[-1][2] == list(range(n_norm + 1))                       # norms + embed_tokens: the tail
  assert all((hi - lo) * 4 <= (64 << 20) for lo, hi, _ in b[1:-1])
  assert sum(hi - lo for lo, hi, _ in b) == off and len(b) <= 12


def test_gradsink_groups_weight_gradient_gemms(monkeypatch):
  """GradSink's queueing policy (host logic only; the launches are faked): without a gradient consumer the dW GEMMs of the whole
  model go out as ONE grouped launch (48 problems at the 160M depth) when backward reaches the embedding; with one (DDP buckets)
  a group is three transformer blocks = 12 problems; accumulate flags follow first-write-in-window; every parameter is reported
  ready exactly once per flush, after its launch."""
  from plainlm_amd import functional as Fn
  calls, ready = [], []
  monkeypatch.setattr(Fn.ops, 'gemm_tn_grouped', lambda probs: calls.append([(id(o), acc) for _, _, o, acc, _ in probs]) or True)
  monkeypatch.setattr(Fn.ops, 'gemm_tn', lambda *a, **k: calls.append(('single', k.get('accumulate'))))
  monkeypatch.setattr(Fn.ops, 'colsum_multi', lambda items: None)

  class FakeParam:
    def __init__(self):
      self.main_grad = object()

  for consumer, group in ((None, 48), (lambda p: ready.append(p), 12)):
    sink = Fn.GradSink()
    sink.on_ready = consumer
    calls.clear()
    ready.clear()
    params = [FakeParam() for _ in range(48)]
    sink.begin_window()
    for i, p in enumerate(params):  # 12 blocks x 4 projections, as backward reaches them
      sink.defer_dw('dy', 'x', p)
      assert len(calls) == (i + 1) // group
    sink.flush_dw()
    assert [len(c) for c in calls] == [group] * (48 // group)
    assert all(acc is False for c in calls for _, acc in c)          # first write in the window overwrites
    if consumer is not None:
      assert ready == params                                        # one report per parameter, in launch order
    calls.clear()
    sink.defer_dw('dy', 'x', params[0])                             # second micro-step of the window: accumulate
    sink.flush_dw()
    assert calls == [('single', True)]
  # the queue is bounded by the bytes it keeps alive as well: 5 pairs of 2 x 400 bytes against a 3000-byte budget
  import torch
  sink = Fn.GradSink()
  sink.dw_queue_bytes_max = 3000
  calls.clear()
  params = [FakeParam() for _ in range(5)]
  sink.begin_window()
  for p in params:
    sink.defer_dw(torch.zeros(100), torch.zeros(100), p)
  assert [len(c) for c in calls] == [3] and sink.dw_queue_bytes == 1600  # issued BEFORE the 4th pair would have crossed the budget; two pairs left
  sink.flush_dw()
  assert len(calls[1]) == 2 and sink.dw_queue_bytes == 0


def test_buckets_keep_blocks_whole_and_dw_groups_end_at_bucket_boundaries():
  """Round 4: with the blocks' Linear weights declared as groups, no block straddles two buckets (its four
  gradients leave one grouped dW launch together), and the reducer tells the dW queue to launch exactly at bucket boundaries once
  the queue holds PLM_DW_GROUP_MB of gradients: 160M, 64 MiB buckets = two blocks each, groups of four blocks = two buckets."""
  import torch
  from plainlm_amd import ddp
  d, h, V, L = 768, 2048, 50280, 12
  sizes = [d] * (2 * L + 1) + [V * d] + [3 * d * d, d * d, 2 * h * d, d * h] * L + [V * d]
  spans, off = [], 0
  for n in sizes:
    spans.append((off, n))
    off += n
  first = 2 * L + 2
  groups = [list(range(first + 4 * l, first + 4 * l + 4)) for l in range(L)]
  b = ddp.plan_buckets(spans, cap_bytes=64 << 20, groups=groups)
  assert b[0][2] == [len(sizes) - 1] and b[-1][2] == list(range(2 * L + 2))
  for _, _, idxs in b[1:-1]:
    assert len(idxs) == 8 and idxs[0] == groups[(idxs[0] - first) // 4][0]  # two whole blocks
  assert sum(hi - lo for lo, hi, _ in b) == off
  # without the groups the same cap cuts through blocks
  assert any(len(idxs) % 4 for _, _, idxs in ddp.plan_buckets(spans, cap_bytes=64 << 20)[1:-1])

  class FakeComm:
    world_size, backend = 2, 'fake'
    def allreduce_avg_(self, span, stream=None):
      pass
  flat = torch.zeros(8)  # CPU: nothing is launched on a device; the spans only need to be consistent
  params = [object() for _ in sizes]
  red = ddp.GradReducer(flat, params, spans, FakeComm(), bucket_cap_mb=64, groups=groups)
  assert red.dw_group_bytes == 80_000_000
  red.begin(sync=True)
  launched = []
  red._launch = lambda bkt, tail=False: launched.append(bkt)
  red.overlap = True  # (CPU reducers launch everything in finish(); here the launch order is what is being checked)
  red.param_ready(params[-1])  # lm_head
  assert launched == [0]
  verdicts = []
  for l in range(L - 1, -1, -1):  # backward: block by block from the top, fc2 / fc1 / w_out / w_qkv inside a block
    for j in (3, 2, 1, 0):
      i = first + 4 * l + j
      v = red.param_queued(params[i])
      verdicts.append(bool(v))
      if v:  # the sink launches the group and reports every member
        for q in sorted(red._queued, reverse=True):
          red.param_ready(params[q])
  # a launch after every FOURTH block (113 MB >= 80 MB, at a bucket boundary; two blocks = 57 MB is a boundary but too small)
  assert [k for k, v in enumerate(verdicts) if v] == [15, 31, 47]
  assert launched == [0, 1, 2, 3, 4, 5, 6]
  red.begin(sync=False)
  assert red.param_queued(params[first]) is None  # micro-steps that do not communicate: the sink's own count decides


@pytest.mark.timeout(600)
def test_no_register_spills_in_the_persistent_gemm_kernels():
  """Round 4's conditional alpha multiply pushed the stream-K (HYB) instantiation of gemm_nt_big_kernel from 234 VGPRs to 256 + 59
  spills (+30 % on every hybrid launch of a data-parallel run) and nobody looked: the kernel was still correct.  Compile
  csrc/gemm_big.hip with the library's flags (hipcc cross-compiles without a GPU) and require: no VGPR / SGPR spill and no scratch in
  any persistent GEMM kernel, and no draining `s_waitcnt vmcnt(0)` inside the K loops of the plain NT kernels."""
  import ast
  import re
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'isa_scan.py'), 'gemm_big.hip'], capture_output=True, text=True, timeout=540)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = r.stdout.splitlines()
  kernels = {}
  for i, l in enumerate(lines):
    if l.startswith('void gemm_') or l.startswith('gemm_'):
      meta = ast.literal_eval(lines[i + 1].strip())  # the dict isa_scan prints
      kernels[l.strip()] = (meta, lines[i + 2] if i + 2 < len(lines) else '')
  big = {k: v for k, v in kernels.items() if 'gemm_nt_big_kernel' in k or 'gemm_tn_big_kernel' in k}
  assert len(big) >= 12, sorted(kernels)
  for k, (meta, span) in big.items():
    assert meta['vspill'] == 0 and meta['scratch'] == 0 and meta['vgpr'] <= 256, (k, meta)
    if 'gemm_tn_big_kernel<true>' in k:
      # the grouped dW kernel keeps its 48-problem tables (3.2 KB of kernel arguments) in SGPRs that hipcc parks in VGPR lanes: known and
      # outside the K loop's critical path as long as the lane moves inside the MFMA span stay a handful (11 today)
      m = re.search(r'v_readlane/v_writelane: (\d+)', span)
      assert m and int(m.group(1)) <= 16, (k, span)
    else:
      assert meta['sspill'] == 0, (k, meta)
    if 'gemm_nt_big_kernel' in k and re.search(r', (true|false), false, false, false>$', k):  # plain AND stream-K (HYB) instantiations - no GLU / GLUB / RoPE epilogue: the K loop never drains the DMA ring
      m = re.search(r'unconditional vmcnt\(0\): (\d+)', span)
      assert m and int(m.group(1)) <= 1, (k, span)


@pytest.mark.timeout(900)
def test_register_budget_of_the_attention_and_128x128_gemm_kernels():
  """The same guard for the other MFMA files: no spill / scratch in any attention backward kernel, document-mask kernel or 128 x 128 GEMM; the causal
  forward kernel is known to park 3 VGPRs and 14 SGPRs at its loop transitions (rounds 3-5: measured, outside the tile loops) - pinned at that, so
  a fourth one is noticed."""
  import ast
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  seen = 0
  for f in ('attn_causal.hip', 'attn_doc.hip', 'attn_generic.hip', 'gemm.hip'):
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'isa_scan.py'), f], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    for i, l in enumerate(lines):
      if i + 1 < len(lines) and lines[i + 1].strip().startswith("{'vgpr'"):
        meta = ast.literal_eval(lines[i + 1].strip())
        name = l.strip()
        if 'attn' not in name and 'gemm' not in name:
          continue
        seen += 1
        if 'attn_fwd_causal_kernel' in name:
          assert meta['vspill'] <= 3 and meta['sspill'] <= 14 and meta['scratch'] <= 16, (name, meta)
        else:
          assert meta['vspill'] == 0 and meta['sspill'] == 0 and meta['scratch'] == 0, (name, meta)
        if 'attn_fwd_doc_kernel' in name or 'attn_bwd_dq_doc_kernel' in name:
          # three workgroups per CU (768 resident workgroups = the whole grid of config_doc_mask.yaml's micro-batch): 512 / 3 registers per lane
          assert meta['vgpr'] <= 168, (name, meta)
  assert seen >= 22, seen



def test_every_entry_point_refuses_null_pointers_without_a_gpu():
  """Error behaviour of the boundary (include/plainlm_hip.h): argument checks come before any HIP call, so they can be exercised here - every
  int-returning entry point that takes a pointer answers a call with null pointers with a negative PLM_E_* code and a message that names the
  entry point (plm_last_error_string), instead of launching or crashing.  (plm_comm_destroy(NULL) is a documented no-op.)"""
  import ctypes as C
  from plainlm_amd import _lib
  lib = _lib.load()
  checked = 0
  for name, (restype, argtypes) in sorted(_lib.SIGNATURES.items()):
    is_ptr = [a is _lib._P or (isinstance(a, type) and issubclass(a, C._Pointer)) for a in argtypes]
    if restype is not _lib._I or not any(is_ptr):
      continue
    args = [None if p else (1.0 if a is _lib._F else 8) for a, p in zip(argtypes, is_ptr)]
    rc = getattr(lib, name)(*args)
    if name == 'plm_comm_destroy':
      assert rc == 0
      continue
    msg = (lib.plm_last_error_string() or b'').decode()
    assert rc < 0, (name, rc)
    stem = name.replace('_ex', '').replace('_ws', '').replace('_capped', '')  # variants report under their family's name
    assert stem in msg or name in msg, (name, msg)
    checked += 1
  assert checked >= 40, checked


def test_shape_checks_of_the_boundary_without_a_gpu():
  """The documented limits of a few entry points, asked with plausible (never dereferenced) pointers: the refusal comes from the argument
  check, names the limit, and nothing is launched."""
  import ctypes as C
  from plainlm_amd import _lib
  lib = _lib.load()
  p = lambda: C.c_void_p(0x100000)
  def refused(name, *args):
    rc = getattr(lib, name)(*args)
    assert rc < 0, (name, rc)
    return (lib.plm_last_error_string() or b'').decode()
  assert 'd % 4 == 0' in refused('plm_rmsnorm_fwd', p(), p(), p(), p(), p(), p(), 16, 6, 1e-6, None)
  assert 'd <= 2048' in refused('plm_rmsnorm_fwd', p(), p(), p(), p(), p(), p(), 16, 4096, 1e-6, None)
  assert 'head_dim 48 unsupported' in refused('plm_attn_fwd', p(), None, None, p(), p(), 2, 64, 2, 48, None)
  assert 'multiple of 4' in refused('plm_attn_fwd', p(), None, None, p(), p(), 2, 63, 2, 64, None)
  assert 'needs its plan' in refused('plm_attn_fwd', p(), p(), None, p(), p(), 2, 64, 2, 64, None)  # a document mask without a plan
  assert 'aligned' in refused('plm_attn_fwd', C.c_void_p(0x100002), None, None, p(), p(), 2, 64, 2, 64, None)
  assert 'M <= 65536' in refused('plm_embed_bwd_sorted', p(), p(), p(), 70000, 64, 1000, 0, p(), 1 << 20, None)
  assert lib.plm_embed_bwd_workspace_bytes(70000, 1000) == 0 and lib.plm_embed_bwd_workspace_bytes(65536, 1000) > 0
  assert lib.plm_attn_doc_plan_bytes(0, 64) == 0 and lib.plm_attn_doc_plan_bytes(8, 1024) > 4 * 8 * 1024


def test_committed_pmc_traffic_profile_is_of_this_tree():
  """`roofline.traffic` of the bench line comes from committed rocprofv3 PMC passes and is reported only when they were taken on THIS tree's
  kernel sources (bench.csrc_sha).  A kernel change without a fresh `bash tools/profile_round.sh rNN` (and its pmc.json copied to
  profiles/rNN_pmc_gemm_traffic.json) would silently turn the driver's `traffic` into null: fail here instead."""
  import json
  import bench
  prof = bench._pmc_profile()
  assert os.path.exists(prof), prof
  assert json.load(open(prof)).get('csrc_sha') == bench.csrc_sha(), (
      f'{os.path.basename(prof)} was taken on another tree: re-run tools/profile_round.sh on a GPU box and commit its pmc.json / pmc.txt')
  got = bench.pmc_traffic('gemm_nt', '160m', 32768, 12)
  assert got['traffic'] > got['algorithmic_bytes'] > 1e8 and got['csrc_sha'] == bench.csrc_sha()


def test_bucket_plan_properties_on_random_layouts():
  """plan_buckets on 300 random flat layouts (parameter sizes over five orders of magnitude, permuted span tables, random block groups of
  adjacent parameters, random caps): the buckets tile the flat buffer exactly once, each is one contiguous span, walks from the END of the buffer
  (bucket k lies above bucket k + 1), holds at most `cap` bytes unless it is a single parameter - or unless the small trailing bucket was merged into
  it - and a block group that fits a bucket is never split."""
  rng = np.random.default_rng(2024)
  for case in range(300):
    n = int(rng.integers(1, 40))
    sizes = [int(10 ** rng.uniform(0, 5.5)) for _ in range(n)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    spans = [(int(offs[i]), sizes[i]) for i in range(n)]
    perm = list(rng.permutation(n))
    table = [spans[i] for i in perm]           # the span table in another order than the buffer
    pos = {int(p): k for k, p in enumerate(perm)}  # buffer index -> table index
    groups, i = [], 0
    while i < n:
      g = int(rng.integers(1, 6))
      if rng.random() < 0.5 and i + g <= n:
        groups.append([pos[j] for j in range(i, i + g)])
      i += g
    cap = int(4 * 10 ** rng.uniform(1, 6))
    out = ddp.plan_buckets(table, cap, groups=groups or None)
    total = int(offs[-1])
    cover = sorted((lo, hi) for lo, hi, _ in out)
    assert cover[0][0] == 0 and cover[-1][1] == total and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), (case, cover)
    assert [lo for lo, _, _ in out] == sorted((lo for lo, _, _ in out), reverse=True), case  # from the end of the buffer backwards
    seen = sorted(i for _, _, idxs in out for i in idxs)
    assert seen == list(range(n)), case
    bucket_of = {i: k for k, (_, _, idxs) in enumerate(out) for i in idxs}
    for k, (lo, hi, idxs) in enumerate(out):
      assert hi - lo == sum(table[i][1] for i in idxs), case
      merged_tail = k == len(out) - 1 and len(out) >= 1  # the last bucket may have absorbed a < 1 MiB trailing one
      assert (hi - lo) * 4 <= cap or len(idxs) == 1 or merged_tail, (case, k, (hi - lo) * 4, cap)
    for g in groups:
      if sum(table[i][1] for i in g) * 4 <= cap and len({bucket_of[i] for i in g}) > 1:
        # allowed only through the tail merge (the group then still sits in ONE bucket) - anything else is a split group
        raise AssertionError((case, 'group split', g, [bucket_of[i] for i in g]))

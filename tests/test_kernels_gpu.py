"""Per-kernel parity tests on a real MI355X, through the C ABI, against fp32 references
(the CPU oracle for the operators it defines; plain fp32 torch matmul for the GEMMs).

Tolerances (stated per test) are relative to the reference tensor's max magnitude:
  bf16 outputs  : 1.6e-2 (two bf16 roundings: 2^-8 each on inputs/intermediates + output)
  fp32 outputs  : 2e-3 for bf16-input MFMA sums, 1e-5 for pure fp32 kernels
"""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402


@pytest.fixture(scope='module')
def ops():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from plainlm_amd import ops as _ops
  return _ops


@pytest.fixture
def plm_env(monkeypatch, ops):
  """Set one of the library's environment switches for one test: the library caches them (no getenv on a launch path), so it is
  told to read them again now and once more when the test's environment has been restored."""
  def set_(name, value):
    monkeypatch.setenv(name, value)
    ops.reload_env()
  yield set_
  monkeypatch.undo()
  ops.reload_env()


def relerr(a, ref):
  a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
  return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def close(a, ref, tol, what=''):
  e = relerr(a, ref)
  assert e <= tol, f'{what}: rel-to-max error {e:.3e} > {tol:.1e}'
  return e


def bf(x):
  return x.to(torch.bfloat16)


# --------------------------------------------------------------------------------------
# hardware probes: the layout facts the MFMA kernels are built on
# --------------------------------------------------------------------------------------
def test_probe_ds_read_tr16(ops):
  got = ops.probe_ds_read_tr16().cpu().numpy()
  # lane l supplied address 8*l bytes over a linear uint16 ramp; within each 16-lane group the lane
  # must receive column (l&15) of the [4 rows][16 cols] block: value = group*64 + j*16 + (l&15)
  l = np.arange(64)[:, None]
  j = np.arange(4)[None, :]
  want = (l >> 4) * 64 + j * 16 + (l & 15)
  assert (got == want).all(), f'ds_read_b64_tr_b16 semantics differ:\n{got[:20]}'


def test_probe_mfma32_layout(ops):
  g = torch.Generator().manual_seed(0)
  A = torch.randn(32, 16, generator=g).bfloat16().float()
  B = torch.randn(16, 32, generator=g).bfloat16().float()  # asymmetric: catches transposes
  got = ops.probe_mfma32(A.cuda(), B.cuda())
  close(got, A @ B, 1e-5, 'mfma 32x32x16 layout')


# --------------------------------------------------------------------------------------
# casts / embedding
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [(8, 8), (64, 64), (136, 72), (2304, 768), (50280, 768)])
def test_cast_transpose(ops, shape):
  x = torch.randn(shape, device='cuda')
  y, yt = ops.cast_bf16_t(x)
  assert torch.equal(y, x.bfloat16())
  assert torch.equal(yt, x.bfloat16().t().contiguous())


@pytest.mark.parametrize('n', [1, 7, 8, 1000, 1 << 20])
def test_cast_flat(ops, n):
  x = torch.randn(n, device='cuda')
  assert torch.equal(ops.cast_bf16(x), x.bfloat16())


def test_embedding_fwd_bwd_repeated_ids(ops, golden_dir):
  z = np.load(f'{golden_dir}/ops.npz')
  W = torch.from_numpy(z['emb_w']).cuda()
  ids = torch.from_numpy(z['emb_ids']).cuda().reshape(-1)
  out = ops.embed_fwd(ids, W)
  assert torch.equal(out.cpu(), torch.from_numpy(z['emb_out']).reshape(-1, W.shape[1]))
  dW = torch.zeros_like(W)
  ops.embed_bwd(ids, torch.from_numpy(z['emb_dout']).cuda().reshape(-1, W.shape[1]).contiguous(), dW)
  close(dW, torch.from_numpy(z['emb_dw']), 1e-6, 'embedding bwd (golden)')


@pytest.mark.parametrize('M,V,d', [(32768, 50280, 768), (4096, 256, 128), (1000, 65535, 64), (65536, 512, 256), (7, 3, 8),
                                   (65537, 300, 64), (98304, 50280, 768), (200000, 1000, 32)])  # last three: more than one slice of 65536 tokens
def test_embedding_bwd_sorted(ops, M, V, d):
  """Sort-based embedding backward: equals index_add (fp64 reference), ignores out-of-range ids, writes zeros into unused
  rows when overwriting, adds onto dW when accumulating, and is bit-reproducible (no atomics) - heavy duplicates included."""
  g = torch.Generator().manual_seed(M + V)
  ids = torch.randint(0, V, (M,), generator=g)
  ids[: M // 4] = ids[0]                       # one very frequent token
  if M > 5:
    ids[5] = -1
    ids[M - 1] = V + 3                         # ignored
  dout = torch.randn(M, d, generator=g)
  ok = (ids >= 0) & (ids < V)
  ref = torch.zeros(V, d, dtype=torch.float64).index_add_(0, ids[ok], dout[ok].double())
  dW = torch.full((V, d), 7.0, device='cuda')  # stale contents must be overwritten
  assert ops.embed_bwd_sorted(ids.cuda(), dout.cuda(), dW, accumulate=False)
  err = (dW.double().cpu() - ref).abs().max().item()
  assert err <= 1e-6 * max(1.0, ref.abs().max().item()) * max(1, M // 4) ** 0.5, err
  dW2 = torch.full((V, d), 7.0, device='cuda')
  ops.embed_bwd_sorted(ids.cuda(), dout.cuda(), dW2, accumulate=False)
  assert torch.equal(dW, dW2)                  # deterministic
  acc = torch.ones(V, d, device='cuda')
  ops.embed_bwd_sorted(ids.cuda(), dout.cuda(), acc, accumulate=True)
  err = (acc.double().cpu() - (ref + 1)).abs().max().item()
  assert err <= 1e-6 * max(1.0, ref.abs().max().item()) * max(1, M // 4) ** 0.5, err


def test_embedding_bwd_sorted_unsupported_shapes(ops):
  ids = torch.zeros(8, dtype=torch.int64, device='cuda')
  assert ops.embed_bwd_sorted(ids, torch.zeros(8, 8, device='cuda'), torch.zeros(70000, 8, device='cuda'), accumulate=False) is False


# --------------------------------------------------------------------------------------
# RMSNorm
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,d', [(48, 128), (1000, 768), (37, 1024), (64, 2048), (5, 64)])
@pytest.mark.parametrize('with_branch', [False, True])
def test_rmsnorm_fwd_bwd(ops, M, d, with_branch):
  g = torch.Generator().manual_seed(M * d)
  x = torch.randn(M, d, generator=g)
  w = 1 + 0.1 * torch.randn(d, generator=g)
  br = bf(0.5 * torch.randn(M, d, generator=g)) if with_branch else None
  dy = bf(torch.randn(M, d, generator=g))
  gin = torch.randn(M, d, generator=g)
  r = (x + br.float()) if with_branch else x
  rr = r.clone().requires_grad_(True)
  ww = w.clone().requires_grad_(True)
  yref = O.rmsnorm(rr, ww)
  yref.backward(dy.float())
  xout, y, rstd = ops.rmsnorm_fwd(x.cuda(), w.cuda(), 1e-6, branch=None if br is None else br.cuda(), write_xout=True)
  close(xout, r, 1e-7, 'xout')
  close(y.float(), yref, 6e-3, 'rmsnorm y (bf16 out)')
  close(rstd, torch.rsqrt(r.pow(2).mean(-1) + 1e-6), 1e-5, 'rstd')
  dx, dxb, dw = ops.rmsnorm_bwd(dy.cuda(), xout, w.cuda(), rstd, gin=gin.cuda(), want_bf16=True)
  close(dx, rr.grad + gin, 2e-5, 'rmsnorm dx')
  assert torch.equal(dxb, dx.bfloat16())
  close(dw, ww.grad, 2e-5, 'rmsnorm dw')
  # accumulate into an existing dw
  acc = torch.ones(d, device='cuda')
  ops.rmsnorm_bwd(dy.cuda(), xout, w.cuda(), rstd, dw_out=acc, dw_accumulate=True)
  close(acc, ww.grad + 1, 2e-5, 'rmsnorm dw accumulate')


def test_rmsnorm_golden(ops, golden_dir):
  z = {k: torch.from_numpy(v) for k, v in np.load(f'{golden_dir}/ops.npz').items() if k.startswith('rms_')}
  _, y, rstd = ops.rmsnorm_fwd(z['rms_x'].cuda(), z['rms_w'].cuda(), 1e-6)
  close(y.float(), z['rms_y'], 6e-3, 'rmsnorm y vs reference')
  dx, _, dw = ops.rmsnorm_bwd(bf(z['rms_dy']).cuda(), z['rms_x'].cuda(), z['rms_w'].cuda(), rstd)
  close(dx, z['rms_dx'], 8e-3, 'rmsnorm dx vs reference (dy rounded to bf16)')
  close(dw, z['rms_dw'], 8e-3, 'rmsnorm dw vs reference')


# --------------------------------------------------------------------------------------
# SwiGLU
# --------------------------------------------------------------------------------------
def test_colsum_multi(ops):
  """One launch for the column sums of a list of [rows, d] partial buffers (the RMSNorm weight gradients of a backward
  pass) vs fp64 sums; accumulate adds into the existing output."""
  g = torch.Generator().manual_seed(9)
  rows, d = 1024, 768
  parts = [torch.randn(rows, d, generator=g).cuda() for _ in range(27)]
  outs = [torch.full((d,), float(i), device='cuda') for i in range(27)]
  ops.colsum_multi([(p, o, i % 2 == 1) for i, (p, o) in enumerate(zip(parts, outs))])
  for i, (p, o) in enumerate(zip(parts, outs)):
    ref = p.double().sum(0) + (float(i) if i % 2 == 1 else 0.0)
    close(o, ref.float(), 1e-5, f'colsum_multi item {i}')
  small = torch.randn(5, 64, generator=g).cuda()
  o1 = torch.empty(64, device='cuda')
  ops.colsum_multi([(small, o1, False)])
  close(o1, small.double().sum(0).float(), 1e-6, 'colsum_multi small')


@pytest.mark.parametrize('M,h', [(48, 512), (333, 2048), (16, 2816)])
def test_swiglu_fwd_bwd(ops, M, h):
  g = torch.Generator().manual_seed(h)
  u = bf(2 * torch.randn(M, 2 * h, generator=g))
  dout = bf(torch.randn(M, h, generator=g))
  uu = u.float().requires_grad_(True)
  ref = O.swiglu(uu, h)
  ref.backward(dout.float())
  out = ops.swiglu_fwd(u.cuda())
  close(out.float(), ref, 1.6e-2, 'swiglu fwd')
  du = ops.swiglu_bwd(dout.cuda(), u.cuda())
  close(du.float(), uu.grad, 1.6e-2, 'swiglu bwd')
  # bit-exact against the reference's bf16 autograd chain evaluated with torch bf16 ops
  ub = u.cuda().requires_grad_(True)
  x, z = ub.split(h, dim=1)
  yb = torch.nn.functional.silu(x) * z
  yb.backward(dout.cuda())
  assert (out.float() - yb.float()).abs().max() <= 2 ** -7 * yb.float().abs().max()
  assert (du.float() - ub.grad.float()).abs().max() <= 2 ** -6 * ub.grad.float().abs().max()


# --------------------------------------------------------------------------------------
# GEMMs
# --------------------------------------------------------------------------------------
NT_SHAPES = [(128, 128, 64), (256, 384, 128), (200, 136, 72), (1000, 2304, 768), (64, 50280, 768), (512, 768, 50280),
             (1, 8, 8), (4096, 768, 2048), (300, 768, 50304), (130, 136, 192)]
# BASELINE configs[3] (config/tr_420M_x8gpu.yaml:20-24,34: d=1024, h=2816, B=8 x T=2048 -> M=16384) at FULL size: the
# launch plans bench.py --config 420m runs - w_qkv / w_out / fc1 / fc2 forward and their dX duals
NT_SHAPES_420M = [(16384, 3072, 1024), (16384, 1024, 1024), (16384, 5632, 1024), (16384, 1024, 2816),
                  (16384, 1024, 3072), (16384, 1024, 5632), (16384, 2816, 1024)]


def test_gemm_families_vs_cpu_matmul(ops):
  """One shape per kernel family against an fp32 matmul computed on the HOST (the other GEMM tests take their reference
  from the same device's fp32 matmul): persistent NT (the three tile shapes of the automatic policy), the hybrid
  whole-K + stream-K NT launch, 128x128 NT, persistent TN (whole-K and split-K items), the grouped TN launch."""
  g = torch.Generator().manual_seed(41)
  def pair(m, k, n, tn=False):
    A = bf(torch.randn((k, m) if tn else (m, k), generator=g))
    B = bf(torch.randn((k, n) if tn else (n, k), generator=g))
    ref = (A.float().t() @ B.float()) if tn else (A.float() @ B.float().t())
    return A, B, ref
  for M, N, K in ((2048, 2304, 768), (2048, 768, 768), (4096, 4096, 256), (200, 136, 72)):
    A, B, ref = pair(M, K, N)
    close(ops.gemm_nt(A.cuda(), B.cuda()).float().cpu(), ref, 6e-3, f'gemm_nt {M}x{N}x{K} vs host')
  # hybrid plan: 640 tiles of 256x256 on 256 CUs (2.5 rounds), and neither 192- nor 128-column tiles pack to >= 0.9
  A, B, ref = pair(32768, 8192, 1280)
  assert _lib_ws(32768, 1280, 8192) > 0
  close(ops.gemm_nt(A.cuda(), B.cuda()).float().cpu(), ref, 6e-3, 'gemm_nt hybrid vs host')
  assert _lib_ws(32768, 768, 8192) == 0  # lm_head dX on the whole chip: 4 x 192 columns = two exact rounds, no hybrid
  for M, N, K in ((2304, 768, 4096), (768, 768, 8192), (264, 136, 200)):
    A, B, ref = pair(M, K, N, tn=True)
    close(ops.gemm_tn(A.cuda(), B.cuda()).cpu(), ref, 2e-5 * math.sqrt(K) + 1e-6, f'gemm_tn {M}x{N}x{K} vs host')
  Kc = 4096
  probs = [pair(m, Kc, n, tn=True) for m, n in ((2304, 768), (768, 768), (4096, 768), (768, 2048))]
  outs = [torch.zeros(r.shape, device='cuda') for _, _, r in probs]
  assert ops.gemm_tn_grouped([(a.cuda(), b.cuda(), o, False, None) for (a, b, _), o in zip(probs, outs)])
  for (_, _, r), o in zip(probs, outs):
    close(o.cpu(), r, 2e-5 * math.sqrt(Kc), 'gemm_tn_grouped vs host')


def _lib_ws(M, N, K):
  from plainlm_amd import _lib
  return _lib.load().plm_gemm_nt_workspace_bytes(M, N, K)


def host_rows(out, A, B, tol, what, tn=False, n_rows=48):
  """A slice of the result against the HOST's fp32 arithmetic (the full-size references of these tests come from the same device's fp32
  matmul - an independent implementation, but not the CPU): 48 output rows spread over the whole matrix, first and last included -
  rows of A B^T (NT), columns of A as rows of A^T B (TN) - recomputed on the CPU in fp64."""
  M = out.shape[0]
  rows = torch.unique(torch.cat([torch.tensor([0, M - 1]), torch.linspace(0, M - 1, n_rows).long(),
                                 torch.randint(0, M, (8,), generator=torch.Generator().manual_seed(M))]))
  Ah, Bh = A.cpu().double(), B.cpu().double()
  ref = (Ah[:, rows].t() @ Bh) if tn else (Ah[rows] @ Bh.t())
  got = out[rows.to(out.device)].double().cpu()
  e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
  assert e <= tol, f'{what} vs host fp64 on {len(rows)} rows: rel-to-max error {e:.3e} > {tol:.1e}'


@pytest.mark.parametrize('M,N,K', NT_SHAPES + NT_SHAPES_420M)
def test_gemm_nt(ops, M, N, K):
  g = torch.Generator().manual_seed(M + N + K)
  A = bf(torch.randn(M, K, generator=g)).cuda()
  B = bf(torch.randn(N, K, generator=g)).cuda()
  ref = A.float() @ B.float().t()
  out32 = ops.gemm_nt(A, B, out_dtype=torch.float32)
  close(out32, ref, 2e-5 * math.sqrt(K), f'gemm_nt fp32 {M}x{N}x{K}')
  host_rows(out32, A, B, 2e-5 * math.sqrt(K), f'gemm_nt fp32 {M}x{N}x{K}')
  out16 = ops.gemm_nt(A, B)
  close(out16.float(), ref, 6e-3, f'gemm_nt bf16 {M}x{N}x{K}')
  # accumulate + device alpha
  alpha = torch.tensor(0.5, device='cuda')
  acc = torch.ones(M, N, device='cuda')
  ops.gemm_nt(A, B, out=acc, accumulate=True, alpha=alpha)
  close(acc, 1 + 0.5 * ref, 2e-5 * math.sqrt(K), 'gemm_nt accumulate/alpha')


@pytest.mark.parametrize('variant', [2, 3, 4, 5, 6, 7])  # 128x128 LDS-DMA, the plain 256x256 ring, the four tile shapes of the automatic policy's deep ring (256x256, 256x192, 256x128, 128x192)
@pytest.mark.parametrize('M,N,K', [(512, 512, 64), (1000, 392, 192), (2048, 768, 768), (8192, 2304, 128), (300, 136, 64),
                                   (33000, 768, 64), (32768, 768, 2304)])
def test_gemm_nt_variants(ops, M, N, K, variant):
  """Every NT kernel variant incl. the persistent big-tile ones: tails in M and N, one and many K-tiles,
  more tiles than CUs (several tiles per persistent workgroup, prefetch across tile boundaries)."""
  g = torch.Generator().manual_seed(M + N + K + variant)
  A = bf(torch.randn(M, K, generator=g)).cuda()
  B = bf(torch.randn(N, K, generator=g)).cuda()
  ref = A.float() @ B.float().t()
  out = ops.gemm_nt(A, B, variant=variant)
  close(out.float(), ref, 6e-3, f'gemm_nt variant {variant} {M}x{N}x{K}')
  host_rows(out, A, B, 6e-3, f'gemm_nt variant {variant} {M}x{N}x{K}')
  alpha = torch.tensor(-0.5, device='cuda')
  wide = torch.zeros(M, N + 24, dtype=torch.bfloat16, device='cuda')  # padded rows (ldc > N)
  ops.gemm_nt(A, B, out=wide[:, :N], alpha=alpha, variant=variant)
  close(wide[:, :N].float(), -0.5 * ref, 6e-3, 'alpha / padded ldc')
  assert (wide[:, N:] == 0).all()


@pytest.mark.parametrize('M,N,K', [(32768, 768, 768), (32768, 768, 2048), (32700, 1032, 1024), (32768, 768, 50304), (20480, 1280, 640),
                                   (32768, 768, 576), (32768, 768, 4096)])
def test_gemm_nt_hybrid(ops, M, N, K, plm_env):
  """Shapes whose 256x256 tile count is a bad multiple of the CU count take the hybrid whole-K + stream-K schedule
  (workspace > 0 asserts that): ragged M/N tails inside the stream-K rows, runs that cross tile boundaries, alpha,
  padded ldc; plus bit-equality with the plain schedule on exactly-representable inputs."""
  from plainlm_amd import _lib
  plm_env('PLM_NT_HYBRID_MIN_K', '64')
  assert _lib.load().plm_gemm_nt_workspace_bytes(M, N, K) > 0, 'shape does not exercise the hybrid schedule'
  g = torch.Generator(device='cuda').manual_seed(M + N + K)
  A = bf(torch.randn(M, K, generator=g, device='cuda'))
  B = bf(torch.randn(N, K, generator=g, device='cuda'))
  ref = A.float() @ B.float().t()
  alpha = torch.tensor(0.5, device='cuda')
  wide = torch.zeros(M, N + 24, dtype=torch.bfloat16, device='cuda')
  ops.gemm_nt(A, B, out=wide[:, :N], alpha=alpha)
  close(wide[:, :N].float(), 0.5 * ref, 6e-3, f'gemm_nt hybrid {M}x{N}x{K}')
  assert (wide[:, N:] == 0).all()
  del ref, wide
  Ai = bf(torch.randint(-3, 4, (M, K), generator=g, device='cuda').float())
  Bi = bf(torch.randint(-3, 4, (N, K), generator=g, device='cuda').float())
  hyb = ops.gemm_nt(Ai, Bi)
  plain = ops.gemm_nt(Ai, Bi, variant=3)
  assert torch.equal(hyb, plain)  # integer inputs: fp32 sums are exact in any order


def test_gemms_with_cu_reserve(ops, plm_env):
  """Multi-GPU runs reserve 16 CUs for RCCL while gradient buckets are in flight (ddp.GradReducer -> ops.set_cu_reserve): the persistent grids shrink to 240
  workgroups and every plan (tile choice, hybrid stream-K, TN split) is recomputed for that count.  Same answers required."""
  plm_env('PLM_NT_HYBRID_MIN_K', '64')
  g = torch.Generator(device='cuda').manual_seed(11)
  M, d, h = 16384, 768, 2048
  fx = bf(torch.randn(M, d, generator=g, device='cuda'))
  fw1 = bf(0.05 * torch.randn(2 * h, d, generator=g, device='cuda'))
  fw2t = bf(0.05 * torch.randn(h, d, generator=g, device='cuda'))
  fwq = bf(0.05 * torch.randn(3 * d, d, generator=g, device='cuda'))
  fcos, fsin = (t.cuda() for t in O.rope_table(64, 1024))
  fu, fact = ops.fc1_swiglu(fx, fw1)  # whole chip
  fref = (fu, fact, ops.fc2_dx_swiglu_bwd(fx, fw2t, fu), ops.qkv_rope(fx, fwq, fcos, fsin, 16, 1024, 12))
  try:
    ops.set_cu_reserve(16)
    for M, N, K in [(32768, 768, 2048), (32768, 2304, 768), (8192, 4096, 768), (32700, 1032, 1024), (4096, 50280, 768)]:
      A = bf(torch.randn(M, K, generator=g, device='cuda'))
      B = bf(torch.randn(N, K, generator=g, device='cuda'))
      close(ops.gemm_nt(A, B).float(), A.float() @ B.float().t(), 6e-3, f'gemm_nt with reserve {M}x{N}x{K}')
    for M, N, K in [(2304, 768, 32768), (768, 768, 8192), (50280, 768, 4096), (4096, 768, 32768)]:
      A = bf(torch.randn(K, M, generator=g, device='cuda'))
      B = bf(torch.randn(K, N, generator=g, device='cuda'))
      close(ops.gemm_tn(A, B), A.float().t() @ B.float(), 2e-5 * math.sqrt(K), f'gemm_tn with reserve {M}x{N}x{K}')
    # the launches with fused epilogues on 240 workgroups: the bits of the same launches on the whole chip (the schedule changes,
    # a tile's arithmetic does not)
    u, act = ops.fc1_swiglu(fx, fw1)
    assert torch.equal(u, fref[0]) and torch.equal(act, fref[1])
    assert torch.equal(ops.fc2_dx_swiglu_bwd(fx, fw2t, fref[0]), fref[2])
    assert torch.equal(ops.qkv_rope(fx, fwq, fcos, fsin, 16, 1024, 12), fref[3])
    K = 4096
    shapes = [(768, 2048), (4096, 768), (768, 768), (2304, 768)] * 3  # what a data-parallel run groups: three blocks
    As = [bf(torch.randint(-3, 4, (K, m), generator=g, device='cuda').float()) for m, _ in shapes]
    Bs = [bf(torch.randint(-3, 4, (K, n), generator=g, device='cuda').float()) for _, n in shapes]
    outs = [torch.empty((m, n), device='cuda') for m, n in shapes]
    assert ops.gemm_tn_grouped([(a, b, o, False, None) for a, b, o in zip(As, Bs, outs)])
    for a, b, o in zip(As, Bs, outs):
      assert torch.equal(o, a.float().t() @ b.float())  # small integers: exact in any summation order
  finally:
    ops.set_cu_reserve(0)


def test_gemm_nt_strided_operand(ops):
  """A is a column block of a wider buffer (the q|k|v and x|z cases)."""
  g = torch.Generator().manual_seed(5)
  big = bf(torch.randn(300, 3 * 128, generator=g)).cuda()
  A = big[:, 128:256]
  B = bf(torch.randn(72, 128, generator=g)).cuda()
  close(ops.gemm_nt(A, B, out_dtype=torch.float32), A.float() @ B.float().t(), 1e-4, 'gemm_nt strided A')


TN_SHAPES = [(128, 128, 64), (256, 128, 512), (136, 72, 200), (2304, 768, 4096), (768, 2048, 1000), (50280, 768, 256),
             (8, 8, 8), (768, 768, 32768), (520, 264, 192), (304, 1000, 1024), (4096, 768, 8192), (256, 256, 64), (50280, 768, 2048), (66000, 256, 1024),
             # the dW GEMMs of a 420M block at full size (contraction over M = 16384 tokens): w_qkv, w_out, fc1, fc2
             (3072, 1024, 16384), (1024, 1024, 16384), (5632, 1024, 16384), (1024, 2816, 16384)]


@pytest.mark.parametrize('M,N,K', TN_SHAPES)
def test_gemm_tn(ops, M, N, K):
  g = torch.Generator().manual_seed(M + 3 * N + K)
  A = bf(torch.randn(K, M, generator=g)).cuda()
  B = bf(torch.randn(K, N, generator=g)).cuda()
  ref = A.float().t() @ B.float()
  out = ops.gemm_tn(A, B)
  close(out, ref, 2e-5 * math.sqrt(K), f'gemm_tn {M}x{N}x{K}')
  host_rows(out, A, B, 2e-5 * math.sqrt(K), f'gemm_tn {M}x{N}x{K}', tn=True)
  acc = torch.full((M, N), 2.0, device='cuda')
  alpha = torch.tensor(0.25, device='cuda')
  ops.gemm_tn(A, B, out=acc, accumulate=True, alpha=alpha)
  close(acc, 2 + 0.25 * ref, 2e-5 * math.sqrt(K), 'gemm_tn accumulate/alpha')


@pytest.mark.parametrize('shapes,K', [
    ([(768, 2048), (4096, 768), (768, 768), (2304, 768)], 32768),   # the four dW GEMMs of a 160M block (fc2, fc1, w_out, w_qkv)
    ([(136, 72), (520, 264)], 1024),                                 # ragged tiles
    ([(256, 256)], 64),                                              # one problem, one K-tile
    ([(768, 768), (8, 8), (1000, 392)], 4096),
    # 24 problems = six 160M blocks: 648 tiles = two whole-K rounds written directly + 136 tiles split over K
    ([(768, 2048), (4096, 768), (768, 768), (2304, 768)] * 6, 2048),
    # 48 problems = all twelve 160M blocks: 1296 tiles = five whole-K rounds + 16 tiles split over K
    ([(768, 2048), (4096, 768), (768, 768), (2304, 768)] * 12, 1024),
    # 6 ragged problems whose 262 tiles straddle the whole-K / split boundary inside one problem
    ([(1000, 1032), (520, 264), (2304, 768), (136, 72), (3000, 1544), (4096, 776)], 1024),
    # the four dW GEMMs of three 420M blocks (fc2, fc1, w_out, w_qkv; the DDP group size) at the full M = 16384
    ([(1024, 2816), (5632, 1024), (1024, 1024), (3072, 1024)] * 3, 16384),
])
def test_gemm_tn_grouped(ops, shapes, K):
  """Grouped dW launch (whole-K tiles for the full rounds + split-K remainder) == the individual GEMMs: every problem, overwrite
  and accumulate, device alpha; bit-equal to the single-problem kernel on exactly-representable inputs."""
  g = torch.Generator(device='cuda').manual_seed(K + len(shapes))
  As = [bf(torch.randn(K, M, generator=g, device='cuda')) for M, _ in shapes]
  Bs = [bf(torch.randn(K, N, generator=g, device='cuda')) for _, N in shapes]
  refs = [a.float().t() @ b.float() for a, b in zip(As, Bs)]
  outs = [torch.full((M, N), 3.0, device='cuda') for M, N in shapes]
  assert ops.gemm_tn_grouped([(a, b, o, False, None) for a, b, o in zip(As, Bs, outs)])
  for o, r, (M, N) in zip(outs, refs, shapes):
    close(o, r, 2e-5 * math.sqrt(K), f'grouped tn {M}x{N}x{K}')
  alpha = torch.tensor(0.25, device='cuda')
  outs = [torch.full((M, N), 2.0, device='cuda') for M, N in shapes]
  assert ops.gemm_tn_grouped([(a, b, o, True, alpha) for a, b, o in zip(As, Bs, outs)])
  for o, r in zip(outs, refs):
    close(o, 2 + 0.25 * r, 2e-5 * math.sqrt(K), 'grouped tn accumulate/alpha')
  Ai = [bf(torch.randint(-3, 4, (K, M), generator=g, device='cuda').float()) for M, _ in shapes]
  Bi = [bf(torch.randint(-3, 4, (K, N), generator=g, device='cuda').float()) for _, N in shapes]
  outs = [torch.empty((M, N), device='cuda') for M, N in shapes]
  assert ops.gemm_tn_grouped([(a, b, o, False, None) for a, b, o in zip(Ai, Bi, outs)])
  for a, b, o in zip(Ai, Bi, outs):
    assert torch.equal(o, ops.gemm_tn(a, b))  # small integers: fp32 sums are exact in any order


@pytest.mark.parametrize('M,h,K', [(2048, 2048, 768), (1000, 1024, 256), (4096, 128, 128), (300, 72, 200), (16384, 2816, 1024)])  # last: 420M at full size
def test_fc1_swiglu_fused_epilogue(ops, M, h, K):
  """fc1 + SwiGLU in one launch (gate / up half-tiles, activation in the GEMM epilogue) == GEMM followed by the stand-alone
  kernel, bit for bit (u and act); the (4096, 128, 128) and (300, 72, 200) shapes take the unfused fallback of the same entry point."""
  g = torch.Generator(device='cuda').manual_seed(M + h)
  x = bf(torch.randn(M, K, generator=g, device='cuda'))
  w = bf(torch.randn(2 * h, K, generator=g, device='cuda') * 0.05)
  u, act = ops.fc1_swiglu(x, w)
  u_ref = ops.gemm_nt(x, w)
  assert torch.equal(u, u_ref)
  assert torch.equal(act, ops.swiglu_fwd(u_ref))
  close(u.float(), x.float() @ w.float().t(), 6e-3, 'fc1_swiglu u vs fp32 matmul')


@pytest.mark.parametrize('M,h,K', [(2048, 2048, 768), (1000, 512, 256), (4096, 128, 128), (300, 72, 200), (16384, 2816, 1024)])  # last: 420M at full size
def test_fc2_dx_swiglu_bwd_fused_epilogue(ops, M, h, K):
  """dX of fc2 + SwiGLU backward in one launch == GEMM followed by the stand-alone kernel, bit for bit; the (4096, 128, 128) and
  (300, 72, 200) shapes take the two-launch fallback of the same entry point."""
  g = torch.Generator(device='cuda').manual_seed(M + 3 * h)
  dy = bf(torch.randn(M, K, generator=g, device='cuda'))
  w2t = bf(torch.randn(h, K, generator=g, device='cuda') * 0.05)
  u = bf(torch.randn(M, 2 * h, generator=g, device='cuda'))
  du = ops.fc2_dx_swiglu_bwd(dy, w2t, u)
  assert torch.equal(du, ops.swiglu_bwd(ops.gemm_nt(dy, w2t), u))


def test_short_batch_tile_128x192(ops):
  """Round 5: the 128x192 tile of the persistent NT kernel (variant 7; 8 waves of 32x96) - the shape the automatic policy takes at the
  reference's document-mask micro-batch (config_doc_mask.yaml:35: B = 8 -> M = 8192: N = 768 is ONE round of 256 tiles, N = 2304 three).
  Bit-equal to the 256x256 tile on exactly representable inputs (any summation order), the automatic policy's result equals the explicit
  variant's bits on random inputs, ragged M / N, and the RoPE epilogue on this tile gives the bits of GEMM + stand-alone pass."""
  g = torch.Generator(device='cuda').manual_seed(75)
  for M, N, K in ((8192, 768, 768), (8192, 2304, 768), (8192, 768, 4096), (4100, 1160, 192), (16384, 1024, 1024)):
    Ai = bf(torch.randint(-3, 4, (M, K), generator=g, device='cuda').float())
    Bi = bf(torch.randint(-3, 4, (N, K), generator=g, device='cuda').float())
    assert torch.equal(ops.gemm_nt(Ai, Bi, variant=7), ops.gemm_nt(Ai, Bi, variant=4)), (M, N, K)
    assert torch.equal(ops.gemm_nt(Ai, Bi), ops.gemm_nt(Ai, Bi, variant=4)), (M, N, K)
    A = bf(torch.randn(M, K, generator=g, device='cuda'))
    B = bf(torch.randn(N, K, generator=g, device='cuda'))
    close(ops.gemm_nt(A, B, variant=7).float(), A.float() @ B.float().t(), 6e-3, f'128x192 tile {M}x{N}x{K}')
    del Ai, Bi, A, B
  # M = 8192: the automatic policy is on the 128x192 tile for N = 768 / 2304 (same bits as the explicit variant: same kernel)
  A = bf(torch.randn(8192, 768, generator=g, device='cuda'))
  for N in (768, 2304):
    B = bf(torch.randn(N, 768, generator=g, device='cuda'))
    assert torch.equal(ops.gemm_nt(A, B), ops.gemm_nt(A, B, variant=7)), N
  for B_, T, nh, K in ((8, 1024, 12, 768), (8, 1000, 12, 768), (16, 512, 12, 768)):
    d = nh * 64
    x = bf(torch.randn(B_ * T, K, generator=g, device='cuda'))
    w = bf(0.1 * torch.randn(3 * d, K, generator=g, device='cuda'))
    cos, sin = (t.cuda() for t in O.rope_table(64, T))
    got = ops.qkv_rope(x, w, cos, sin, B_, T, nh)
    two = ops.gemm_nt(x, w)
    if B_ * T == 8192:
      assert torch.equal(two, ops.gemm_nt(x, w, variant=7))
    ops.rope_qk_(two, cos, sin, B_, T, nh)
    assert torch.equal(got, two), (B_, T, nh, K)


@pytest.mark.parametrize('variant', [4, 5, 6, 7])
def test_overlapped_epilogue_exact(ops, variant):
  """Round 5: in the last K-tile of an interior tile the persistent NT kernels hand finished accumulator quadrants to the epilogue while
  the other quadrants are still being multiplied (csrc/gemm_big.hip, OVL), with the stores counted into the LDS-DMA ring's vmcnt waits - of
  that K-tile and of the next tile's first one.  A miscounted wait reads a half-tile that has not landed: exactness on integer-valued
  operands (fp32 sums exact in any order, one bf16 rounding) against the device's fp32 matmul, for two K-tiles (first K-tile directly in
  front of the overlapped one), many, one (no overlap), edge tiles in M and N, 1 - 4 tiles per workgroup, three runs each."""
  g = torch.Generator(device='cuda').manual_seed(50 + variant)
  for M, N, K in ((32768, 768, 768), (9000, 1160, 128), (65536, 768, 192), (4096, 4096, 2048), (16384, 1024, 64), (33000, 2304, 320)):
    A = bf(torch.randint(-2, 3, (M, K), generator=g, device='cuda').float())
    B = bf(torch.randint(-2, 3, (N, K), generator=g, device='cuda').float())
    ref = (A.float() @ B.float().t()).to(torch.bfloat16)
    for _ in range(3):
      out = ops.gemm_nt(A, B, variant=variant)
      assert torch.equal(out, ref), (variant, M, N, K, (out.float() - ref.float()).abs().max().item())
    del A, B, ref, out


def test_fused_entry_points_fall_back_under_gemm_v1(ops, plm_env):
  """PLM_GEMM_V1=1 forces the register-staged 128x128 GEMMs: the three entry points with fused epilogues must then take their
  two-launch paths (the backward one asks for its d(act) scratch with PLM_E_WORKSPACE and gets it) and still be right."""
  plm_env('PLM_GEMM_V1', '1')
  g = torch.Generator(device='cuda').manual_seed(5)
  M, d, h = 1024, 256, 512
  x = bf(torch.randn(M, d, generator=g, device='cuda'))
  w1 = bf(0.05 * torch.randn(2 * h, d, generator=g, device='cuda'))
  u, act = ops.fc1_swiglu(x, w1)
  close(u.float(), x.float() @ w1.float().t(), 6e-3, 'fc1 (v1)')
  assert torch.equal(act, ops.swiglu_fwd(u))
  w2t = bf(0.05 * torch.randn(h, d, generator=g, device='cuda'))
  du = ops.fc2_dx_swiglu_bwd(x, w2t, u)
  assert torch.equal(du, ops.swiglu_bwd(ops.gemm_nt(x, w2t), u))
  cos, sin = (t.cuda() for t in O.rope_table(64, 256))
  wq = bf(0.05 * torch.randn(3 * 128, d, generator=g, device='cuda'))
  two = ops.gemm_nt(x, wq)
  ops.rope_qk_(two, cos, sin, 4, 256, 2)
  assert torch.equal(ops.qkv_rope(x, wq, cos, sin, 4, 256, 2), two)


def test_gemm_linearity(ops):
  """Size-independent property at a full-size shape: G(a1+a2) == G(a1)+G(a2) for exactly-representable inputs."""
  g = torch.Generator().manual_seed(9)
  M, N, K = 32768, 768, 768
  A1 = torch.randint(-4, 5, (M, K), generator=g).float()
  A2 = torch.randint(-4, 5, (M, K), generator=g).float()
  B = torch.randint(-4, 5, (N, K), generator=g).float()
  f = lambda a: ops.gemm_nt(bf(a).cuda(), bf(B).cuda(), out_dtype=torch.float32)
  assert torch.equal(f(A1 + A2), f(A1) + f(A2))  # small integers: every partial sum is exact in fp32


# --------------------------------------------------------------------------------------
# attention (with in-kernel RoPE) vs the oracle
# --------------------------------------------------------------------------------------
def _attn_ref(qkv, B, T, nh, doc_start, hd=64):
  cos, sin = O.rope_table(hd, T)
  q, k, v = (t.reshape(B, T, nh, hd) for t in qkv.float().split(nh * hd, dim=1))
  qr = O.rope_apply(q, cos, sin)
  kr = O.rope_apply(k, cos, sin)
  return O.attention(qr, kr, v, doc_start).reshape(B * T, nh * hd)


def _random_docs(B, T, seed):
  rng = np.random.default_rng(seed)
  out = []
  for _ in range(B):
    lens, tot = [], 0
    while tot < T + 1:
      n = int(min(rng.integers(1, max(2, T // 3)), T + 1 - tot))
      lens.append(n)
      tot += n
    out.append(lens)
  return out


@pytest.mark.parametrize('B,T,nh', [(2, 64, 2), (1, 128, 1), (2, 256, 3), (1, 1024, 2), (1, 200, 2), (1, 2048, 1), (2, 320, 2), (3, 512, 1), (1, 836, 2),
                                    (100, 1024, 1), (1, 4096, 2)])  # 800 tiles: a plan in tile order (past the sort's limit); T = 4096
@pytest.mark.parametrize('masked', [False, True])
def test_attention_fwd_bwd(ops, B, T, nh, masked):
  g = torch.Generator().manual_seed(T * nh + masked)
  d = nh * 64
  qkv = bf(torch.randn(B * T, 3 * d, generator=g))
  dout = bf(torch.randn(B * T, d, generator=g))
  ds = O.doc_start_from_lengths(_random_docs(B, T, T), T) if masked else None
  leaf = qkv.float().requires_grad_(True)
  ref = _attn_ref(leaf, B, T, nh, ds)
  ref.backward(dout.float())
  cos, sin = (t.cuda() for t in O.rope_table(64, T))
  dsg = None if ds is None else ds.cuda()
  qrot = ops.rope_qk_(qkv.cuda(), cos, sin, B, T, nh)
  out, lse = ops.attn_fwd(qrot, B, T, nh, dsg)
  close(out.float(), ref, 1.6e-2, 'attention out')
  dqkv = ops.attn_bwd(qrot, out, dout.cuda(), lse, cos, sin, B, T, nh, dsg)
  gq, gk, gv = (t for t in leaf.grad.split(d, dim=1))
  dq, dk, dv = (t.float() for t in dqkv.split(d, dim=1))
  close(dv, gv, 2e-2, 'attention dV')
  close(dk, gk, 2e-2, 'attention dK')
  close(dq, gq, 2e-2, 'attention dQ')


def _plan_ref(ds, T, nh, split_min=8):
  """numpy restatement of the plan (include/plainlm_hip.h, attn_common.h): header counts, doc_end, and the two item lists - the n / 4 heaviest
  tiles with cost >= split_min as two 64-row halves when the grid is resident at once - in stable order of expected duration."""
  B = ds.shape[0]
  nt = (T + 127) // 128
  n = B * nt
  de = np.empty((B, T), dtype=np.int64)
  for b in range(B):
    for j in range(T):
      later = np.nonzero(ds[b, j + 1:] > j)[0]
      de[b, j] = j + 1 + later[0] if len(later) else T
  up = lambda x: (x + 63) // 64
  tiles = [[], []]  # per list: (b, r0, cost, (cost0, bound0), (cost1, bound1), whole-tile bound)
  for b in range(B):
    for t in range(nt):
      r0, r1 = t * 128, min(t * 128 + 64, T - 1)
      lo0, lo1 = int(ds[b, r0]) // 64, int(ds[b, r1]) // 64
      hi0, hi1 = up(min(T, r0 + 64)), up(min(T, r0 + 128))
      e0, e1 = up(int(de[b, min(r0 + 63, T - 1)])), up(int(de[b, min(r0 + 127, T - 1)]))
      tiles[0].append((b, r0, hi1 - lo0, (hi0 - lo0, lo0), (hi1 - lo1, lo1), lo0))
      tiles[1].append((b, r0, e1 - r0 // 64, (e0 - r0 // 64, e0), (e1 - (r0 + 64) // 64, e1), e1))
  may_split = split_min > 0 and n * nh <= 1024 and n <= 768
  lists = []
  for l, tl in enumerate(tiles):
    by_cost = sorted(range(n), key=lambda i: (-tl[i][2], i))
    split = {i for k, i in enumerate(by_cost) if may_split and l == 0 and k < n // 4 and tl[i][2] >= split_min and tl[i][1] + 64 < T}  # key tiles are never split
    items = []  # (est, tile, half, record)
    for i, (b, r0, c, h0, h1, wb) in enumerate(tl):
      if i in split:
        for a, (ca, ba) in enumerate((h0, h1)):
          items.append(((ca + 1) // 2 + 1, i, a, (b, r0 + 64 * a, ba, (1 << 30) | ca)))
      else:
        items.append((c, i, 0, (b, r0, wb, c)))
    if n <= 768:  # DOC_PLAN_SORT_MAX: longer lists stay in tile order (and unsplit)
      items.sort(key=lambda it: (-it[0], it[1], it[2]))
    recs = [it[3] for it in items]
    lists.append(recs)
  return de, lists[0], lists[1]


@pytest.mark.parametrize('B,T,nh', [(2, 64, 2), (3, 200, 1), (8, 1024, 12), (2, 2048, 16), (5, 836, 3), (32, 1024, 12), (48, 2048, 12), (50, 2048, 1)])  # last two: 768 (the sort's limit) and 800 tiles
def test_attention_doc_plan(ops, B, T, nh):
  """plm_attn_doc_plan against its numpy restatement: header, doc_end[], both item lists (which tiles are split, bounds, order)."""
  ds = O.doc_start_from_lengths(_random_docs(B, T, 3 * T + B), T)
  plan = ops.attn_doc_plan(ds.cuda(), nh).cpu().numpy()
  de, iq, ik = _plan_ref(ds.numpy(), T, nh)
  n = B * ((T + 127) // 128)
  cap = n + n // 4
  pq = (8 + B * T + 3) // 4 * 4
  assert plan.shape[0] == pq + 8 * cap
  assert plan[0] == len(iq) and plan[1] == len(ik) and (plan[2:8] == 0).all()
  assert np.array_equal(plan[8:8 + B * T].reshape(B, T), de)
  assert np.array_equal(plan[pq:pq + 4 * len(iq)].reshape(-1, 4), np.array(iq))
  assert np.array_equal(plan[pq + 4 * cap:pq + 4 * cap + 4 * len(ik)].reshape(-1, 4), np.array(ik))
  again = ops.attn_doc_plan(ds.cuda(), nh).cpu().numpy()
  live = np.r_[0:8 + B * T, pq:pq + 4 * len(iq), pq + 4 * cap:pq + 4 * cap + 4 * len(ik)]
  assert np.array_equal(plan[live], again[live])


@pytest.mark.parametrize('kind', ['one_doc', 'singletons', 'tile_aligned_64', 'tile_aligned_128', 'off_by_one', 'geometric_256', 'long_then_short'])
def test_attention_doc_mask_structures(ops, kind):
  """Document layouts that sit on the kernels' segment boundaries (a wave's idle / masked / unmasked tile classes, the per-lane thresholds,
  the plan's bounds): one document per row (== causal), one-token documents, documents that start exactly on / one off 64- and 128-row tile
  edges, the bench's geometric lengths, a long document followed by short ones.  Forward + backward vs the oracle, two runs bit-equal,
  and the plan-less call (plan built inside ops) gives the same bits."""
  B, T, nh = 2, 512, 2
  rng = np.random.default_rng(5)
  if kind == 'one_doc':
    docs = [[T + 1]] * B
  elif kind == 'singletons':
    docs = [[1] * (T + 1)] * B
  elif kind == 'tile_aligned_64':
    docs = [[64] * 8 + [1], [192, 64, 128, 128, 1]]
  elif kind == 'tile_aligned_128':
    docs = [[128, 256, 128, 1], [256, 256, 1]]
  elif kind == 'off_by_one':
    docs = [[63, 65, 127, 129, 128, 1], [1, 127, 1, 255, 129]]
  elif kind == 'geometric_256':
    docs = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
  else:
    docs = [[400, 30, 30, 30, 23], [7, 500, 6]]
  assert all(sum(d) == T + 1 for d in docs)
  g = torch.Generator().manual_seed(len(kind))
  d = nh * 64
  qkv = bf(torch.randn(B * T, 3 * d, generator=g))
  dout = bf(torch.randn(B * T, d, generator=g))
  ds = O.doc_start_from_lengths(docs, T)
  leaf = qkv.float().requires_grad_(True)
  ref = _attn_ref(leaf, B, T, nh, ds)
  ref.backward(dout.float())
  cos, sin = (t.cuda() for t in O.rope_table(64, T))
  dsg = ds.cuda()
  plan = ops.attn_doc_plan(dsg, nh)
  qrot = ops.rope_qk_(qkv.cuda(), cos, sin, B, T, nh)
  out, lse = ops.attn_fwd(qrot, B, T, nh, dsg, plan)
  close(out.float(), ref, 1.6e-2, f'attention out [{kind}]')
  dqkv = ops.attn_bwd(qrot, out, dout.cuda(), lse, cos, sin, B, T, nh, dsg, plan)
  for name, got, want in zip('qkv', dqkv.split(d, dim=1), leaf.grad.split(d, dim=1)):
    if kind == 'singletons' and name != 'v':  # one visible key per row: dS = P (dP - delta) is exactly zero, so dQ = dK = 0 up to bf16(O)'s rounding in delta
      assert got.float().abs().max().item() <= 1e-2 * leaf.grad[:, 2 * d:].abs().max().item()
      continue
    close(got.float(), want, 2e-2, f'attention d{name} [{kind}]')
  out2, lse2 = ops.attn_fwd(qrot, B, T, nh, dsg)
  dqkv2 = ops.attn_bwd(qrot, out2, dout.cuda(), lse2, cos, sin, B, T, nh, dsg)
  assert torch.equal(out, out2) and torch.equal(lse, lse2) and torch.equal(dqkv, dqkv2)
  if kind == 'one_doc':  # a single document per row is the causal mask: the two kernel families agree closely
    outc, _ = ops.attn_fwd(qrot, B, T, nh)
    close(out.float(), outc.float(), 8e-3, 'one document per row vs the causal kernels')


@pytest.mark.parametrize('B,T', [(2, 64), (3, 200), (4, 1024)])
def test_doc_start_from_the_references_bool_mask(ops, B, T):
  """The reference passes a bool [B, T, T] mask (engine/engine.py:21-23, data_prep_utils.py:7-23); plm_attn_doc_start_from_mask turns it into
  doc_start and CHECKS every row: a block-diagonal causal mask converts exactly (status 0), a mask doc_start cannot express (a hole inside a
  document, a key ahead of the query, an empty row) raises the status flag."""
  ds = O.doc_start_from_lengths(_random_docs(B, T, 11 * T + B), T)
  mask = O.mask_from_doc_start(ds)
  got, status = ops.doc_start_from_mask(mask.cuda())
  assert torch.equal(got.cpu(), ds) and status.item() == 0
  bi = [(b, i) for b in range(B) for i in range(T) if int(ds[b, i]) <= i - 2]  # rows that see at least three keys
  for what in ('hole', 'future', 'empty'):
    bad = mask.clone()
    b, i = bi[len(bi) // 2]
    if what == 'hole':
      bad[b, i, i - 1] = False  # a key inside the row's document is missing
    elif what == 'future':
      bad[b, i, min(i + 4, T - 1)] = True if i + 1 < T else bad[b, i, T - 1]
      if i + 1 >= T:
        bad[b, i - 1, i] = True
    else:
      bad[b, i, :] = False
    _, status = ops.doc_start_from_mask(bad.cuda())
    assert status.item() == 1, what
  # through the model API: Transformer._doc_start refuses such a mask at the next conversion
  import plainlm_amd as P
  P.Transformer._mask_status = None
  bad = mask.clone()
  bad[0, 5, 9] = True
  P.Transformer._doc_start(bad.cuda(), B, T)
  with pytest.raises(ValueError, match='block-diagonal'):
    P.Transformer._doc_start(mask.cuda(), B, T)
  assert torch.equal(P.Transformer._doc_start(mask.cuda(), B, T).cpu(), ds)
  P.Transformer.check_mask_status()


def test_attention_doc_masks_random_shapes(ops):
  """Seeded sweep over shapes no hand-picked case has: T from 4 to ~700 in steps of 4 (shorter than a key tile, one row over a tile edge, ragged
  last tiles of 128-row items and of their 64-row halves), 1-4 sequences, 1-3 heads, document lengths from one token to the whole row - forward +
  backward against the oracle, causal twin (one document per row) against the causal kernels."""
  rng = np.random.default_rng(20261002)
  for case in range(24):
    B, nh = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    T = 4 * int(rng.integers(1, 176)) if case % 3 else 4 * int(rng.integers(1, 20))
    mean = float(rng.choice([1.5, 7, 40, 150, 400, 2000]))
    docs = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / mean), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
    g = torch.Generator().manual_seed(case)
    d = nh * 64
    qkv = bf(torch.randn(B * T, 3 * d, generator=g))
    dout = bf(torch.randn(B * T, d, generator=g))
    ds = O.doc_start_from_lengths(docs, T)
    leaf = qkv.float().requires_grad_(True)
    ref = _attn_ref(leaf, B, T, nh, ds)
    ref.backward(dout.float())
    cos, sin = (t.cuda() for t in O.rope_table(64, T))
    qrot = ops.rope_qk_(qkv.cuda(), cos, sin, B, T, nh)
    out, lse = ops.attn_fwd(qrot, B, T, nh, ds.cuda())
    tag = f'case {case}: B={B} T={T} nh={nh} mean doc {mean}'
    close(out.float(), ref, 1.6e-2, 'attention out, ' + tag)
    dqkv = ops.attn_bwd(qrot, out, dout.cuda(), lse, cos, sin, B, T, nh, ds.cuda())
    scale = leaf.grad.abs().max().item()
    for name, got, want in zip('qkv', dqkv.split(d, dim=1), leaf.grad.split(d, dim=1)):
      err = (got.float().cpu() - want).abs().max().item()
      assert err <= 2e-2 * max(want.abs().max().item(), 0.05 * scale), (tag, name, err)  # (short documents: dQ / dK are nearly zero, judged on dV's scale)


@pytest.mark.parametrize('hd', [32, 128])
@pytest.mark.parametrize('B,T,nh,masked', [(2, 64, 2, False), (1, 200, 3, True), (2, 512, 2, True), (1, 1024, 2, False), (3, 328, 1, True)])
def test_attention_other_head_dims(ops, hd, B, T, nh, masked):
  """models/transformer.py:34 allows any dim // n_heads; 64 is what every shipped config has and what the tuned kernels serve.  Head dims 32 and
  128 take the plain kernel family of csrc/attn_generic.hip behind the same entry points (RoPE by the stand-alone pass, element-wise masks, the
  inverse rotation as an in-place pass over dQ | dK): forward + backward vs the oracle, causal and with document masks, ragged T, bit-reproducible."""
  g = torch.Generator().manual_seed(hd * T + nh + masked)
  d = nh * hd
  qkv = bf(torch.randn(B * T, 3 * d, generator=g))
  dout = bf(torch.randn(B * T, d, generator=g))
  ds = O.doc_start_from_lengths(_random_docs(B, T, T + hd), T) if masked else None
  leaf = qkv.float().requires_grad_(True)
  ref = _attn_ref(leaf, B, T, nh, ds, hd=hd)
  ref.backward(dout.float())
  cos, sin = (t.cuda() for t in O.rope_table(hd, T))
  dsg = None if ds is None else ds.cuda()
  qrot = ops.rope_qk_(qkv.cuda(), cos, sin, B, T, nh)
  out, lse = ops.attn_fwd(qrot, B, T, nh, dsg)
  close(out.float(), ref, 1.6e-2, f'attention out (hd {hd})')
  dqkv = ops.attn_bwd(qrot, out, dout.cuda(), lse, cos, sin, B, T, nh, dsg)
  for name, got, want in zip('qkv', dqkv.split(d, dim=1), leaf.grad.split(d, dim=1)):
    close(got.float(), want, 2e-2, f'attention d{name} (hd {hd})')
  out2, lse2 = ops.attn_fwd(qrot, B, T, nh, dsg)
  assert torch.equal(out, out2) and torch.equal(dqkv, ops.attn_bwd(qrot, out2, dout.cuda(), lse2, cos, sin, B, T, nh, dsg))


def test_attention_doc_requires_plan_at_the_c_abi(ops):
  """The C entry points refuse a document mask without its plan (no silent slow path)."""
  import ctypes as C
  from plainlm_amd import _lib
  B, T, nh = 1, 128, 1
  qkv = torch.zeros(B * T, 192, dtype=torch.bfloat16, device='cuda')
  ds = torch.zeros(B, T, dtype=torch.int32, device='cuda')
  out = torch.empty(B * T, 64, dtype=torch.bfloat16, device='cuda')
  lse = torch.empty(B, nh, T, device='cuda')
  p = lambda t: C.c_void_p(t.data_ptr())
  rc = _lib.load().plm_attn_fwd(p(qkv), p(ds), C.c_void_p(0), p(out), p(lse), B, T, nh, 64, C.c_void_p(torch.cuda.current_stream().cuda_stream))
  assert rc != 0 and b'plan' in _lib.load().plm_last_error_string()


def test_rope_qk_golden(ops, golden_dir):
  """In-place RoPE against the reference's apply_rotary_emb_complex_like outputs."""
  z = {k: torch.from_numpy(v) for k, v in np.load(f'{golden_dir}/ops.npz').items() if k.startswith('rope_')}
  B, T, nh, hd = z['rope_q'].shape
  d = nh * hd
  qkv = bf(torch.cat([z['rope_q'].reshape(B * T, d), z['rope_k'].reshape(B * T, d), torch.ones(B * T, d)], dim=1)).cuda()
  cos, sin = (t.cuda() for t in O.rope_table(hd, T))
  ref_q = O.rope_apply(qkv[:, :d].float().cpu().reshape(B, T, nh, hd), cos.cpu(), sin.cpu()).reshape(B * T, d)
  ops.rope_qk_(qkv, cos, sin, B, T, nh)
  close(qkv[:, :d].float(), z['rope_qr'].reshape(B * T, d), 8e-3, 'rope q vs reference')
  close(qkv[:, d:2 * d].float(), z['rope_kr'].reshape(B * T, d), 8e-3, 'rope k vs reference')
  assert (qkv[:, :d].float().cpu() - ref_q).abs().max() <= 2 ** -7 * ref_q.abs().max()
  assert (qkv[:, 2 * d:] == 1).all()  # v untouched


@pytest.mark.parametrize('B,T,nh,K', [(4, 256, 2, 128), (8, 1024, 12, 768), (3, 100, 1, 64), (5, 200, 2, 128), (3, 344, 5, 64), (8, 2048, 16, 1024)])  # last: 420M at full size
def test_qkv_projection_with_rope(ops, B, T, nh, K):
  """Projection GEMM + in-place RoPE pass (plm_qkv_rope_bf16) vs the oracle."""
  g = torch.Generator().manual_seed(B * T + nh)
  d = nh * 64
  x = bf(torch.randn(B * T, K, generator=g))
  w = bf(0.1 * torch.randn(3 * d, K, generator=g))
  cos, sin = O.rope_table(64, T)
  y = x.float() @ w.float().t()
  q, k, v = (t.reshape(B, T, nh, 64) for t in y.split(d, dim=1))
  ref = torch.cat([O.rope_apply(q, cos, sin).reshape(B * T, d), O.rope_apply(k, cos, sin).reshape(B * T, d), v.reshape(B * T, d)], dim=1)
  got = ops.qkv_rope(x.cuda(), w.cuda(), cos.cuda(), sin.cuda(), B, T, nh)
  close(got.float(), ref, 8e-3, 'qkv projection + rope')
  # the rotation in the GEMM epilogue (big shapes) and the stand-alone pass (small ones / fallback) give the same bits
  two = ops.gemm_nt(x.cuda(), w.cuda())
  ops.rope_qk_(two, cos.cuda(), sin.cuda(), B, T, nh)
  assert torch.equal(got, two)


def test_attention_softmax_rescale_branch(ops):
  """Force a large running-max jump late in the row (guide rule: rare data-dependent branch needs its own test)."""
  B, T, nh, d = 1, 256, 1, 64
  g = torch.Generator().manual_seed(3)
  qkv = 0.3 * torch.randn(B * T, 3 * d, generator=g)
  qkv[200, d:2 * d] = 6.0   # key 200 spikes
  qkv[230:, 0:d] += 2.0     # queries after it align with the spike
  qkv = bf(qkv)
  ref = _attn_ref(qkv, B, T, nh, None)
  cos, sin = (t.cuda() for t in O.rope_table(64, T))
  out, _ = ops.attn_fwd(ops.rope_qk_(qkv.cuda(), cos, sin, B, T, nh), B, T, nh)
  close(out.float(), ref, 1.6e-2, 'attention with max jump')


def test_attention_golden(ops, golden_dir):
  """SDPA vectors produced by the reference itself (no RoPE: identity table)."""
  z = {k: torch.from_numpy(v) for k, v in np.load(f'{golden_dir}/ops.npz').items() if k.startswith('att')}
  B, T, nh, hd = z['attc_q'].shape
  cos = torch.ones(T, hd // 2).cuda()
  sin = torch.zeros(T, hd // 2).cuda()
  for tag in 'cm':
    qkv = bf(torch.cat([z[f'att{tag}_{n}'].reshape(B * T, nh * hd) for n in 'qkv'], dim=1)).cuda()
    ds = None
    if tag == 'm':
      docs = [[int(v) for v in row if v > 0] for row in z['attm_docs_lengths']]
      ds = O.doc_start_from_lengths(docs, T).cuda()
    out, lse = ops.attn_fwd(qkv, B, T, nh, ds)  # identity RoPE: the golden SDPA vectors have none
    close(out.float(), z[f'att{tag}_o'].reshape(B * T, nh * hd), 2e-2, f'attention golden {tag}')
    dqkv = ops.attn_bwd(qkv, out, bf(z[f'att{tag}_do']).reshape(B * T, nh * hd).cuda(), lse, cos, sin, B, T, nh, ds)
    for i, n in enumerate('qkv'):
      close(dqkv[:, i * nh * hd:(i + 1) * nh * hd].float(), z[f'att{tag}_d{n}'].reshape(B * T, nh * hd), 3e-2, f'golden d{n} {tag}')


# --------------------------------------------------------------------------------------
# cross entropy
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,V', [(48, 777), (64, 256), (33, 50280), (8, 8192 * 8), (4, 70000)])
def test_cross_entropy(ops, M, V):
  g = torch.Generator().manual_seed(V)
  logits = bf(3 * torch.randn(M, V, generator=g))
  tgt = torch.randint(0, V, (M,), generator=g)
  leaf = logits.float().requires_grad_(True)
  loss = O.cross_entropy(leaf, tgt)
  loss.backward()
  buf = logits.cuda().clone()
  rows = ops.ce_fwd_bwd_(buf, tgt.cuda(), 1.0 / M)
  lm = ops.mean(rows)
  assert abs(lm.item() - loss.item()) <= 2e-6 * abs(loss.item()) + 1e-6
  close(buf.float(), leaf.grad, 8e-3, 'dlogits (bf16)')


def test_cross_entropy_padded_rows(ops):
  """Rows padded to a multiple of 64 columns (the lm_head layout): pads must come back as zeros."""
  M, V, ld = 40, 50280, 50304
  g = torch.Generator().manual_seed(11)
  logits = bf(3 * torch.randn(M, V, generator=g))
  tgt = torch.randint(0, V, (M,), generator=g)
  leaf = logits.float().requires_grad_(True)
  loss = O.cross_entropy(leaf, tgt)
  loss.backward()
  buf = torch.full((M, ld), 7.0, dtype=torch.bfloat16, device='cuda')
  buf[:, :V] = logits.cuda()
  rows = ops.ce_fwd_bwd_(buf, tgt.cuda(), 1.0 / M, V=V)
  assert abs(ops.mean(rows).item() - loss.item()) <= 2e-6 * abs(loss.item()) + 1e-6
  close(buf[:, :V].float(), leaf.grad, 8e-3, 'dlogits (padded rows)')
  assert (buf[:, V:] == 0).all()


def test_cast_transpose_padded(ops):
  x = torch.randn(200, 72, device='cuda')
  out_t = torch.zeros(72, 256, dtype=torch.bfloat16, device='cuda')
  y, yt = ops.cast_bf16_t(x, out_t=out_t)
  assert torch.equal(yt[:, :200], x.bfloat16().t()) and (yt[:, 200:] == 0).all()


def test_cross_entropy_golden(ops, golden_dir):
  z = np.load(f'{golden_dir}/ops.npz')
  logits = bf(torch.from_numpy(z['ce_logits'])).cuda()
  ref = O.cross_entropy(logits.float().cpu(), torch.from_numpy(z['ce_targets']))
  rows = ops.ce_fwd_bwd_(logits, torch.from_numpy(z['ce_targets']).cuda(), 1.0 / logits.shape[0])
  assert abs(ops.mean(rows).item() - ref.item()) < 1e-5
  assert abs(ref.item() - float(z['ce_loss'])) < 2e-2  # bf16 rounding of the logits only


def test_cast_bf16_t_multi(ops):
  """One launch for a list of weights == the per-weight cast, bit for bit (incl. a padded transposed buffer and > 56 items,
  which the C side splits into several launches)."""
  g = torch.Generator().manual_seed(5)
  shapes = [(2304, 768), (768, 768), (4096, 768), (768, 2048), (50280, 768), (8, 8), (72, 136)] + [(64, 64)] * 60
  items, want = [], []
  for R, Cc in shapes:
    src = torch.randn(R, Cc, generator=g).cuda()
    pad = -(-R // 64) * 64
    out, out_t = torch.empty(R, Cc, dtype=torch.bfloat16, device='cuda'), torch.zeros(Cc, pad, dtype=torch.bfloat16, device='cuda')
    items.append((src, out, out_t))
    want.append(ops.cast_bf16_t(src, out_t=torch.zeros(Cc, pad, dtype=torch.bfloat16, device='cuda')))
  ops.cast_bf16_t_multi(items)
  for (src, out, out_t), (w, wt) in zip(items, want):
    assert torch.equal(out, w) and torch.equal(out_t, wt), tuple(src.shape)


def test_scale_bf16_and_axpy_f32(ops):
  """The two device-scalar passes of the chunked lm_head + cross-entropy backward."""
  g = torch.Generator().manual_seed(11)
  for n in (8, 4099, 1 << 20):
    x = torch.randn(n, generator=g).to(torch.bfloat16).cuda()
    alpha = torch.tensor(0.3125, device='cuda')
    want = (x.float() * 0.3125).to(torch.bfloat16)
    assert torch.equal(ops.scale_bf16_(x.clone(), alpha), want), n
    a, b = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    want = torch.addcmul(a.double(), b.double(), torch.tensor(0.3125, dtype=torch.float64, device='cuda')).float()
    got = ops.axpy_f32_(a.clone(), b, alpha, accumulate=True)
    assert (got - want).abs().max().item() <= 1e-6, n
    assert torch.equal(ops.axpy_f32_(torch.full_like(a, float('nan')), b, alpha, accumulate=False), b * 0.3125), n
    assert torch.equal(ops.axpy_f32_(a.clone(), b, None, accumulate=True), a + b), n


# --------------------------------------------------------------------------------------
# optimizer tail
# --------------------------------------------------------------------------------------
def test_sumsq_and_adamw(ops):
  g = torch.Generator().manual_seed(1)
  n = 1_000_003
  x = torch.randn(n + 1, generator=g).cuda()[:n]  # odd length
  x = x.clone()
  got = ops.sumsq(x)
  assert abs(got.item() - (x.double() ** 2).sum().item()) < 1e-4 * n
  p = torch.randn(4096, generator=g).cuda()
  gr = torch.randn(4096, generator=g).cuda()
  ref = torch.nn.Parameter(p.clone())
  ref.grad = gr.clone() * 0.5
  opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, eps=1e-8)
  m, v = torch.zeros_like(p), torch.zeros_like(p)
  clip = torch.tensor(0.5, device='cuda')
  for step in (1, 2, 3):
    opt.step()
    ops.adamw_(p, gr, m, v, 1e-2, 0.9, 0.95, 1e-8, 0.1, step, clip)
  close(p, ref.detach(), 1e-5, 'adamw vs torch.optim.AdamW')

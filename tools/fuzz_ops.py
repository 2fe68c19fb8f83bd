"""Random-shape sweep of the non-GEMM entry points (RMSNorm fwd / bwd, SwiGLU and the plain activations, cross-entropy, weight casts, embedding
fwd / bwd, attention causal and with document masks at head dims 32 / 64 / 128) against fp32 torch arithmetic on the same device.  Run on an
MI355X:  python tools/fuzz_ops.py [seed] [cases]   - prints one line per failing case and a summary; exit code 1 if anything failed."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from plainlm_amd import ops  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402

BF = torch.bfloat16


def relmax(a, ref):
  return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


def case_rmsnorm(rnd, g):
  M, d = rnd.choice([1, 3, 37, 256, 1000, 4099]), 4 * rnd.randint(1, 512)
  x = torch.randn(M, d, device='cuda', generator=g)
  w = 1 + 0.1 * torch.randn(d, device='cuda', generator=g)
  br = torch.randn(M, d, device='cuda', generator=g).to(BF) if rnd.random() < 0.5 else None
  xo, y, rstd = ops.rmsnorm_fwd(x, w, 1e-6, branch=br)
  r = x + (br.float() if br is not None else 0)
  ref = r * torch.rsqrt((r * r).mean(-1, keepdim=True) + 1e-6) * w
  e = [relmax(y.float(), ref)]
  if br is not None:
    e.append(relmax(xo, r) * 1e3)  # fp32 exact up to rounding of the add
  dy = torch.randn(M, d, device='cuda', generator=g).to(BF)
  leaf = r.clone().requires_grad_(True)
  wl = w.clone().requires_grad_(True)
  (leaf * torch.rsqrt((leaf * leaf).mean(-1, keepdim=True) + 1e-6) * wl).backward(dy.float())
  dx, _, dw = ops.rmsnorm_bwd(dy, r.contiguous(), w, rstd)
  e += [relmax(dx, leaf.grad) * 2, relmax(dw, wl.grad) * 2]
  return f'rmsnorm M={M} d={d} branch={br is not None}', max(e), 1.2e-2


def case_ce(rnd, g):
  M, V = rnd.choice([1, 5, 48, 333, 2048]), rnd.choice([3, 8, 97, 777, 4096, 50280, 70001])
  Vp = -(-V // 64) * 64
  logits = torch.zeros(M, Vp, device='cuda', dtype=BF)
  logits[:, :V] = (3 * torch.randn(M, V, device='cuda', generator=g)).to(BF)
  tgt = torch.randint(0, V, (M,), device='cuda', generator=g)
  leaf = logits[:, :V].float().requires_grad_(True)
  ref = torch.nn.functional.cross_entropy(leaf, tgt, reduction='none')
  ref.mean().backward()
  buf = logits.clone()
  rows = ops.ce_fwd_bwd_(buf, tgt, 1.0 / M, V=V)
  e = max(relmax(rows, ref.detach()), relmax(buf[:, :V].float(), leaf.grad) / 2)
  pad = buf[:, V:].float().abs().max().item() if Vp > V else 0.0
  return f'ce M={M} V={V}', max(e, pad), 6e-3


def case_act(rnd, g):
  M, h = rnd.choice([1, 16, 333, 1000]), 8 * rnd.randint(1, 400)
  kind = rnd.choice(['swiglu', 'silu', 'relu2'])
  dy = torch.randn(M, h, device='cuda', generator=g).to(BF)
  if kind == 'swiglu':
    u = torch.randn(M, 2 * h, device='cuda', generator=g).to(BF)
    leaf = u.float().requires_grad_(True)
    ref = torch.nn.functional.silu(leaf[:, :h]).to(BF).float() * leaf[:, h:]
    ref.backward(dy.float())
    e = max(relmax(ops.swiglu_fwd(u).float(), ref.detach()), relmax(ops.swiglu_bwd(dy, u).float(), leaf.grad))
  else:
    k = sorted(ops.ACT_KINDS, key=ops.ACT_KINDS.get)[0 if kind == 'silu' else 1]  # the names ops.py gives kinds 0 (silu) and 1 (relu squared)
    u = torch.randn(M, h, device='cuda', generator=g).to(BF)
    leaf = u.float().requires_grad_(True)
    ref = torch.nn.functional.silu(leaf) if kind == 'silu' else torch.relu(leaf) ** 2
    ref.backward(dy.float())
    e = max(relmax(ops.act_fwd(u, k).float(), ref.detach()), relmax(ops.act_bwd(dy, u, k).float(), leaf.grad))
  return f'act {kind} M={M} h={h}', e, 1.6e-2


def case_embed(rnd, g):
  M, V, d = rnd.choice([1, 7, 1000, 40000, 70000, 140000]), rnd.choice([3, 256, 5000, 50280]), 4 * rnd.randint(1, 200)
  ids = torch.randint(0, V, (M,), device='cuda', generator=g)
  W = torch.randn(V, d, device='cuda', generator=g)
  e = [0.0 if torch.equal(ops.embed_fwd(ids, W), W[ids]) else 1.0]
  dout = torch.randn(M, d, device='cuda', generator=g)
  ref = torch.zeros(V, d, dtype=torch.float64, device='cuda').index_add_(0, ids, dout.double())
  dW = torch.full((V, d), 3.0, device='cuda')
  assert ops.embed_bwd_sorted(ids, dout, dW, accumulate=False)
  dW2 = torch.full((V, d), 5.0, device='cuda')
  ops.embed_bwd_sorted(ids, dout, dW2, accumulate=False)
  e += [relmax(dW, ref) * 1e3 / max(1.0, (M / V) ** 0.5), 0.0 if torch.equal(dW, dW2) else 1.0]  # fp32 sums of ~M / V rows each
  return f'embed M={M} V={V} d={d}', max(e), 2e-3


def case_cast(rnd, g):
  r, c = 8 * rnd.randint(1, 700), 8 * rnd.randint(1, 300)
  w = torch.randn(r, c, device='cuda', generator=g)
  out = torch.empty(r, c, device='cuda', dtype=BF)
  rp = -(-r // 64) * 64
  out_t = torch.zeros(c, rp, device='cuda', dtype=BF)
  ops.cast_bf16_t(w, out=out, out_t=out_t)
  ok = torch.equal(out, w.to(BF)) and torch.equal(out_t[:, :r], w.to(BF).t()) and (rp == r or out_t[:, r:].float().abs().max().item() == 0)
  return f'cast {r}x{c}', 0.0 if ok else 1.0, 0.5


def case_attn(rnd, g):
  hd = rnd.choice([32, 64, 64, 64, 128])
  B, T, nh = rnd.randint(1, 5), 4 * rnd.randint(2, 300), rnd.randint(1, 4)
  masked = rnd.random() < 0.6
  d = nh * hd
  qkv = torch.randn(B * T, 3 * d, generator=torch.Generator().manual_seed(rnd.randint(0, 1 << 30))).to(BF)
  dout = torch.randn(B * T, d, generator=torch.Generator().manual_seed(rnd.randint(0, 1 << 30))).to(BF)
  ds = None
  if masked:
    rows = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = min(int(np.random.default_rng(rnd.randint(0, 1 << 30)).geometric(1.0 / rnd.choice([3, 40, 300]))), T + 1 - tot)
        lens.append(n)
        tot += n
      rows.append(lens)
    ds = O.doc_start_from_lengths(rows, T)
  leaf = qkv.float().requires_grad_(True)
  q, k, v = (t.reshape(B, T, nh, hd) for t in leaf.split(d, dim=1))
  cos, sin = O.rope_table(hd, T)
  q, k, v = O.rope_apply(q, cos, sin).transpose(1, 2), O.rope_apply(k, cos, sin).transpose(1, 2), v.transpose(1, 2)
  idx = torch.arange(T)
  allow = idx[None, :] <= idx[:, None]
  mask = allow[None].expand(B, T, T) if ds is None else (allow[None] & (idx[None, None, :] >= ds.long()[:, :, None]))
  s = (q @ k.transpose(-1, -2)) / hd ** 0.5
  p = torch.softmax(s.masked_fill(~mask[:, None], float('-inf')), -1)
  ref = (p @ v).transpose(1, 2).reshape(B * T, d)
  ref.backward(dout.float())
  cosg, sing = cos.cuda(), sin.cuda()
  dsg = None if ds is None else ds.cuda()
  qrot = ops.rope_qk_(qkv.cuda(), cosg, sing, B, T, nh)
  out, lse = ops.attn_fwd(qrot, B, T, nh, dsg)
  dqkv = ops.attn_bwd(qrot, out, dout.cuda(), lse, cosg, sing, B, T, nh, dsg)
  e = max(relmax(out.float().cpu(), ref.detach()), relmax(dqkv.float().cpu(), leaf.grad))
  return f'attn hd={hd} B={B} T={T} nh={nh} masked={masked}', e, 2.5e-2


def main():
  seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  rnd = random.Random(seed)
  g = torch.Generator(device='cuda').manual_seed(seed)
  cases = [case_rmsnorm, case_ce, case_act, case_embed, case_cast, case_attn, case_attn]
  bad, worst = 0, {}
  for it in range(n):
    fn = rnd.choice(cases)
    try:
      name, e, tol = fn(rnd, g)
    except Exception as ex:  # noqa: BLE001 - a refusal is a finding too
      name, e, tol = f'{fn.__name__}: {type(ex).__name__}: {str(ex)[:160]}', float('inf'), 0.0
    worst[fn.__name__] = max(worst.get(fn.__name__, 0.0), e if e != float('inf') else 0.0)
    if not e <= tol:
      bad += 1
      print(f'FAIL [{it}] {name}: err {e:.3e} > {tol:.1e}', flush=True)
  print({'seed': seed, 'cases': n, 'failed': bad, 'worst_error_by_family': {k: float(f'{v:.2e}') for k, v in worst.items()}})
  sys.exit(1 if bad else 0)


if __name__ == '__main__':
  main()

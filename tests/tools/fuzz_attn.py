"""Random-shape sweep of attention fwd+bwd (causal and document masks) against the CPU oracle (run on an MI355X)."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from oracle import cpu_ref as O  # noqa: E402
from plainlm_amd import ops  # noqa: E402


def relmax(a, ref):
  return ((a.double().cpu() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


def main():
  seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
  rnd = random.Random(seed)
  g = torch.Generator().manual_seed(seed)
  bad = 0
  for it in range(n):
    B = rnd.choice([1, 2, 3])
    T = rnd.choice([4, 60, 64, 100, 128, 192, 260, 516, 1024, 1536])
    nh = rnd.choice([1, 2, 3])
    masked = rnd.random() < 0.5
    d = nh * 64
    qkv = (0.5 * torch.randn(B * T, 3 * d, generator=g)).bfloat16()
    dout = torch.randn(B * T, d, generator=g).bfloat16()
    ds = None
    if masked:
      docs = []
      for _ in range(B):
        lens, tot = [], 0
        while tot < T + 1:
          k = min(rnd.randint(1, max(2, T // 2)), T + 1 - tot)
          lens.append(k)
          tot += k
        docs.append(lens)
      ds = O.doc_start_from_lengths(docs, T)
    cos, sin = O.rope_table(64, T)
    x = qkv.float().requires_grad_(True)
    q, k, v = (t.reshape(B, T, nh, 64) for t in x.split(d, dim=1))
    ref = O.attention(O.rope_apply(q, cos, sin), O.rope_apply(k, cos, sin), v, ds).reshape(B * T, d)
    ref.backward(dout.float())
    dev = qkv.cuda()
    rot = ops.rope_qk_(dev.clone(), cos.cuda(), sin.cuda(), B, T, nh)
    dsg = None if ds is None else ds.cuda()
    out, lse = ops.attn_fwd(rot, B, T, nh, dsg)
    dqkv = ops.attn_bwd(rot, out, dout.cuda(), lse, cos.cuda(), sin.cuda(), B, T, nh, dsg)
    e1, e2 = relmax(out.float(), ref), relmax(dqkv.float(), x.grad)
    ok = e1 <= 1.6e-2 and e2 <= 2.5e-2
    bad += not ok
    print('ok  ' if ok else 'FAIL', B, T, nh, 'doc' if masked else 'causal', f'{e1:.2e} {e2:.2e}', flush=True)
  print('failures:', bad)
  sys.exit(1 if bad else 0)


if __name__ == '__main__':
  main()

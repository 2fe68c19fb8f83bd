"""Random-shape sweep of the GEMM entry points against fp32 matmul (run on an MI355X; not part of the test-suite budget)."""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from plainlm_amd import ops  # noqa: E402

BF = torch.bfloat16


def relerr(a, ref):
  return ((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


def main():
  seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
  rnd = random.Random(seed)
  g = torch.Generator(device='cuda').manual_seed(seed)
  bad = 0
  for it in range(n):
    kind = rnd.choice(['nt', 'nt', 'tn', 'grouped', 'glu', 'glub', 'rope'])
    # let the hybrid (split-K) schedule trigger on small K too - except where bit-equality with a whole-K launch is the test
    if kind in ('glu', 'glub', 'rope'):
      os.environ.pop('PLM_NT_HYBRID_MIN_K', None)
    else:
      os.environ['PLM_NT_HYBRID_MIN_K'] = '64'
    ops.reload_env()
    if kind == 'nt':
      M = rnd.choice([8, 136, 256, 1000, 4096, 8192, 20480, 32768, 33000]) + 8 * rnd.randint(0, 3)
      N = rnd.choice([8, 72, 256, 392, 768, 1032, 2304, 4096]) + 8 * rnd.randint(0, 2)
      K = rnd.choice([64, 128, 576, 768, 1024, 2048, 4096, 8192]) + rnd.choice([0, 0, 0, 8, 64])
      A = torch.randn(M, K, generator=g, device='cuda').to(BF)
      B = torch.randn(N, K, generator=g, device='cuda').to(BF)
      ref = A.float() @ B.float().t()
      e = relerr(ops.gemm_nt(A, B).float(), ref)
      tol = 6e-3
    elif kind == 'tn':
      M = rnd.choice([8, 136, 256, 768, 2304, 4096, 50280]) + 8 * rnd.randint(0, 2)
      N = rnd.choice([8, 72, 256, 768, 2048]) + 8 * rnd.randint(0, 2)
      K = rnd.choice([64, 200, 1024, 4096, 8192, 32768]) + rnd.choice([0, 0, 64, 8])
      A = torch.randn(K, M, generator=g, device='cuda').to(BF)
      B = torch.randn(K, N, generator=g, device='cuda').to(BF)
      ref = A.float().t() @ B.float()
      acc = rnd.random() < 0.5
      out = torch.full((M, N), 1.5, device='cuda')
      ops.gemm_tn(A, B, out=out, accumulate=acc)
      e = relerr(out, ref + (1.5 if acc else 0.0))
      tol = 2e-5 * math.sqrt(K) + 1e-6
    elif kind in ('glu', 'glub', 'rope'):
      # launches with fused epilogues must give the bits of GEMM + stand-alone kernel (shapes on both sides of the fused / fallback line)
      M = rnd.choice([300, 512, 1000, 4096, 8200, 32768])
      K = rnd.choice([64, 128, 576, 768, 1024]) + rnd.choice([0, 0, 8])
      if kind == 'rope':
        T = rnd.choice([100, 256, 344, 1024])
        Bt = max(1, M // T)
        M, nh = Bt * T, rnd.choice([1, 2, 5, 12])
        N = 3 * nh * 64
        from plainlm_amd.transformer import rope_tables
        cos, sin = (t.cuda() for t in rope_tables(64, T))
        x = torch.randn(M, K, generator=g, device='cuda').to(BF)
        w = (0.05 * torch.randn(N, K, generator=g, device='cuda')).to(BF)
        two = ops.gemm_nt(x, w)
        ops.rope_qk_(two, cos, sin, Bt, T, nh)
        e = 0.0 if torch.equal(ops.qkv_rope(x, w, cos, sin, Bt, T, nh), two) else 1.0
      else:
        h = rnd.choice([72, 128, 256, 1024, 2048, 2816])
        N = 2 * h
        x = torch.randn(M, K, generator=g, device='cuda').to(BF)
        if kind == 'glu':
          w = (0.05 * torch.randn(2 * h, K, generator=g, device='cuda')).to(BF)
          u, act = ops.fc1_swiglu(x, w)
          u2 = ops.gemm_nt(x, w)
          e = 0.0 if torch.equal(u, u2) and torch.equal(act, ops.swiglu_fwd(u2)) else 1.0
        else:
          w = (0.05 * torch.randn(h, K, generator=g, device='cuda')).to(BF)
          u = torch.randn(M, 2 * h, generator=g, device='cuda').to(BF)
          e = 0.0 if torch.equal(ops.fc2_dx_swiglu_bwd(x, w, u), ops.swiglu_bwd(ops.gemm_nt(x, w), u)) else 1.0
      tol = 0.0
    else:
      cnt = rnd.choice([1, 3, 8, 17, 31, 48])
      K = rnd.choice([64, 512, 4096, 32768])
      shp = [(rnd.choice([8, 136, 768, 2304, 4096]) + 8 * rnd.randint(0, 2), rnd.choice([8, 264, 768, 2048]) + 8 * rnd.randint(0, 2)) for _ in range(cnt)]
      As = [torch.randn(K, m, generator=g, device='cuda').to(BF) for m, _ in shp]
      Bs = [torch.randn(K, n_, generator=g, device='cuda').to(BF) for _, n_ in shp]
      accs = [rnd.random() < 0.5 for _ in shp]
      outs = [torch.full((m, n_), 0.5, device='cuda') for m, n_ in shp]
      assert ops.gemm_tn_grouped([(a, b, o, c, None) for a, b, o, c in zip(As, Bs, outs, accs)])
      e = max(relerr(o, a.float().t() @ b.float() + (0.5 if c else 0.0)) for a, b, o, c in zip(As, Bs, outs, accs))
      M, N = shp[0]
      tol = 2e-5 * math.sqrt(K) + 1e-6
    ok = e <= tol
    bad += not ok
    print(('ok  ' if ok else 'FAIL'), kind, M, N, K, f'{e:.2e}', flush=True)
  print('failures:', bad)
  sys.exit(1 if bad else 0)


if __name__ == '__main__':
  main()

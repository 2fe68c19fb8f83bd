"""Launch every MFMA kernel of the 160M step twice (warm + measured) in a fixed order, for rocprofv3 --pmc passes.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 tools/prof_kernels.py
  rocprofv3 --pmc WRITE_SIZE ...                                              (separate pass: TCC slots)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ... (MFMA utilisation)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT ...

The program prints one line ``ORDER <json>``: the csrc sha of the tree and the launch list (name, kernel-name substring,
shape, algorithmic bytes / flops).  tools/pmc_report.py joins it with the counter CSVs by dispatch order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import csrc_sha  # noqa: E402
from plainlm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
# PLM_PROF_CONFIG = 160m (default: the headline step) | 420m (B = 8, T = 2048, d = 1024, h = 2816, 16 heads) | 160m_b8 (the 160M shapes at the reference's
# document-mask micro-batch of 8); an environment variable, not a flag: nothing but the program may stand behind rocprofv3's "--"
_CFG = os.environ.get('PLM_PROF_CONFIG', '160m')
B, T, d, h, V, nh = {'160m': (32, 1024, 768, 2048, 50280, 12), '420m': (8, 2048, 1024, 2816, 50280, 16), '160m_b8': (8, 1024, 768, 2048, 50280, 12)}[_CFG]
N_BLOCKS = 12  # per grouped dW launch (48 problems = the library's maximum; the 420M step issues two such launches)
M = B * T
NT = {'nt qkv fwd': (M, 3 * d, d), 'nt out fwd': (M, d, d), 'nt fc1 fwd': (M, 2 * h, d), 'nt fc2 fwd': (M, d, h), 'nt head fwd': (M, V, d),
      'nt dX qkv': (M, d, 3 * d), 'nt dX fc1': (M, d, 2 * h), 'nt dX fc2': (M, h, d), 'nt dX head': (M, d, 50304)}
TN = {'tn dW qkv': (3 * d, d, M), 'tn dW out': (d, d, M), 'tn dW fc1': (2 * h, d, M), 'tn dW fc2': (d, h, M), 'tn dW head': (V, d, M)}


def main():
  lib = _lib.load()
  order = []

  def entry(name, match, flops, alg, **shape):
    order.append(dict(name=name, match=match, flops=flops, algorithmic_bytes=alg, **shape))

  for name, (m, n, k) in NT.items():
    A = torch.randn(m, k, device='cuda').to(BF)
    Bm = torch.randn(n, k, device='cuda').to(BF)
    out = torch.empty(m, n, device='cuda', dtype=BF)
    torch.cuda.synchronize()
    glu = name == 'nt fc1 fwd'   # in the step this launch carries the SwiGLU epilogue (writes act [M, h] as well)
    glub = name == 'nt dX fc2'   # ... and this one the SwiGLU backward (reads u [M, 2h], writes du [M, 2h] instead of d(act) [M, h])
    rope = name == 'nt qkv fwd'  # ... and this one RoPE on the q | k columns (tables stay in L2: no extra algorithmic bytes)
    Um = torch.randn(m, 2 * n, device='cuda').to(BF) if glub else None
    if rope:
      from plainlm_amd.transformer import rope_tables
      cos_, sin_ = (t_.cuda() for t_ in rope_tables(64, T))
    for _ in range(2):
      if glu:
        ops.fc1_swiglu(A, Bm)
      elif glub:
        ops.fc2_dx_swiglu_bwd(A, Bm, Um)
      elif rope:
        ops.qkv_rope(A, Bm, cos_, sin_, m // T, T, n // 192)
      else:
        ops.gemm_nt(A, Bm, out=out)
    torch.cuda.synchronize()
    alg = 2.0 * (m * k + n * k + m * n) + (m * n if glu else 0) + (2.0 * 3 * m * n if glub else 0)
    suffix = ' + swiglu (epilogue)' if glu else ' + swiglu bwd (epilogue)' if glub else ' + rope (epilogue)' if rope else ''
    entry(name + suffix, 'gemm_nt', 2.0 * m * n * k, alg, M=m, N=n, K=k)
    del Um
    if lib.plm_gemm_nt_workspace_bytes(m, n, k) > 0:
      entry(name + ' (stream-K reduce)', 'nt_streamk_reduce', 0.0, 0.0, M=m, N=n, K=k, part_of=name)
    if name == 'nt head fwd':
      # the same launch on 256x128 tiles - an XCD's 32-tile rectangle then needs 4 A + 8 B half-panels = 3.1 MB (fits the
      # 4 MB L2) instead of 4.7 MB; reported beside the shipped 256x256 launch (traffic / algorithmic and, in kbench --variants, its time)
      for _ in range(2):
        ops.gemm_nt(A, Bm, out=out, variant=6)
      torch.cuda.synchronize()
      entry('nt head fwd [256x128 tiles, not shipped]', 'gemm_nt', 2.0 * m * n * k, alg, M=m, N=n, K=k)
    del A, Bm, out
  for name, (m, n, k) in TN.items():
    A = torch.randn(k, m, device='cuda').to(BF)
    Bm = torch.randn(k, n, device='cuda').to(BF)
    out = torch.zeros(m, n, device='cuda')
    torch.cuda.synchronize()
    for _ in range(2):
      ops.gemm_tn(A, Bm, out=out, accumulate=True)
    torch.cuda.synchronize()
    entry(name, 'gemm_tn', 2.0 * m * n * k, 2.0 * (m * k + n * k) + 8.0 * m * n, M=m, N=n, K=k)
    if lib.plm_gemm_tn_workspace_bytes(m, n, k) > 0:
      entry(name + ' (split-K reduce)', 'splitk_reduce', 0.0, 0.0, M=m, N=n, K=k, part_of=name)
    del A, Bm, out
  # the dW GEMMs of all twelve blocks as the engine issues them on one GPU: one grouped launch (whole-K tiles + split remainder) + one reduce
  gp = []
  for _blk in range(N_BLOCKS):
    for name in ('tn dW fc2', 'tn dW fc1', 'tn dW out', 'tn dW qkv'):
      m, n, k = TN[name]
      gp.append((torch.randn(k, m, device='cuda').to(BF), torch.randn(k, n, device='cuda').to(BF), torch.zeros(m, n, device='cuda'), False, None))
  torch.cuda.synchronize()
  for _ in range(2):
    assert ops.gemm_tn_grouped(gp)
  torch.cuda.synchronize()
  fl = sum(2.0 * a.shape[1] * b.shape[1] * a.shape[0] for a, b, *_ in gp)
  alg = sum(2.0 * (a.numel() + b.numel()) + 4.0 * o.numel() for a, b, o, *_ in gp)
  gname = 'tn dW 12 blocks (grouped x48)'
  entry(gname, 'gemm_tn', fl, alg, K=M)
  entry('tn dW 12 blocks (grouped reduce)', 'tn_grouped_reduce', 0.0, 0.0, part_of=gname)
  del gp
  # attention: the step's inputs have the statistics of a freshly initialised model (projection outputs ~N(0, 0.4))
  qkv = (0.4 * torch.randn(M, 3 * d, device='cuda')).to(BF)
  dout = (0.01 * torch.randn(M, d, device='cuda')).to(BF)
  from plainlm_amd.transformer import rope_tables
  cos, sin = (t_.cuda() for t_ in rope_tables(64, T))
  ds = None
  if _CFG == '160m_b8':  # the reference's document-mask config: random documents, mean length 256 (bench.py --doc-mask)
    import numpy as np
    from plainlm_amd.engine import doc_start_from_lengths
    rng = np.random.default_rng(1234)
    docs = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
    ds = doc_start_from_lengths(docs, T).cuda()
  plan = ops.attn_doc_plan(ds, nh) if ds is not None else None
  torch.cuda.synchronize()
  for _ in range(2):
    out, lse = ops.attn_fwd(qkv, B, T, nh, ds, plan)
  torch.cuda.synchronize()
  att_fl = ops.attn_flops(B, T, nh, 64, ds)  # the pairs the mask leaves (causal: T (T + 1) / 2 per head and sequence)
  entry('attn fwd', 'attn_fwd', att_fl, 2.0 * (M * 3 * d + M * d), B=B, T=T, nh=nh)
  for _ in range(2):
    ops.attn_bwd(qkv, out, dout, lse, cos, sin, B, T, nh, ds, plan)
  torch.cuda.synchronize()
  for kname in ('attn_bwd_dq', 'attn_bwd_dkdv', 'attn_bwd_fused', 'attn_bwd_dq_reduce'):
    order.append(dict(name=kname, match=kname, flops=2.0 * att_fl if kname in ('attn_bwd_fused',) else att_fl, algorithmic_bytes=2.0 * (M * 3 * d * 2 + M * d * 2),
                      B=B, T=T, nh=nh, optional=True))
  print('ORDER ' + json.dumps({'csrc_sha': csrc_sha(), 'launches': order}))


if __name__ == '__main__':
  main()

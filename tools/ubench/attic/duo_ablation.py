"""Timing-only ablations of the two-workgroups-per-CU SwiGLU-backward launch (gemm_duo.hip, PLM_DUO_DBG bits) at the 160M in-step shape.
Usage (GPU box): python tools/duo_ablation.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops
BF = torch.bfloat16
M, d, h = 32768, 768, 2048
A = torch.randn(M, d, device='cuda').to(BF)
W2T = (torch.randn(h, d, device='cuda') * 0.02).to(BF)
Um = torch.randn(M, 2 * h, device='cuda').to(BF)
def t(fn, it=20):
  for _ in range(8): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(it): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / it * 1000
def setenv(**kv):
  for k, v in kv.items():
    if v is None: os.environ.pop(k, None)
    else: os.environ[k] = str(v)
  ops.reload_env()
fn = lambda: ops.fc2_dx_swiglu_bwd(A, W2T, Um)
o = torch.empty(M, h, device='cuda', dtype=BF)
setenv(PLM_NT_DUO='0'); print('1wg glub', round(t(fn), 1))
print('1wg plain 256x256', round(t(lambda: ops.gemm_nt(A, W2T, out=o, variant=4)), 1))
for stg in ('0', '5'):
  setenv(PLM_NT_DUO='7', PLM_DUO_STAGGER_US=stg)
  print('duo plain stagger', stg, round(t(lambda: ops.gemm_nt(A, W2T, out=o, variant=7)), 1))
  for dbg, name in ((0, 'full'), (1, 'loads resident'), (2, 'no stores'), (4, 'stores resident'), (8, 'no exp/rcp'), (3, 'loads resident + no stores'), (11, 'resident loads, no stores, no exp'), (5, 'loads+stores resident')):
    os.environ['PLM_DUO_DBG'] = str(dbg); ops.reload_env()
    print(f'duo glub stagger {stg} dbg {dbg:2d} {name}:', round(t(fn), 1))
  os.environ.pop('PLM_DUO_DBG'); ops.reload_env()

"""Optimizer tail on its own: FlatAdamW.clip_and_step (||g||^2 sweep + AdamW, with and without the bf16 shadows written by the AdamW launch)
and the stand-alone weight cast it replaces, 160M model.  Usage (GPU box): python tools/optim_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import plainlm_amd as P
from plainlm_amd.optim import FlatAdamW


def t(fn, it=20):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(it): fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / it


for shadows in ('1', '0'):
  os.environ['PLM_ADAMW_SHADOWS'] = shadows
  model = bench.build_model(bench.CONFIGS['160m'], torch.device('cuda'))
  model.enable_main_grad()
  opt = FlatAdamW(model, P.get_param_groups(model, 0.1), lr=1e-4, betas=[0.9, 0.95], eps=1e-8, weight_decay=0.1)
  opt.flat_g.normal_()
  step = t(lambda: opt.clip_and_step(1.0))
  def cast():
    model.invalidate_shadows()
    model.refresh_shadows()
  print(f'PLM_ADAMW_SHADOWS={shadows}: clip_and_step {step:.3f} ms, stand-alone cast of all weights {t(cast):.3f} ms', flush=True)
  del model, opt

"""Learning-rate sequences of the reference's four schedulers (optim/init_optim.py:73-137 -> optim/lr_schedule.py), produced by the REFERENCE's
own initialize_scheduler on a dummy optimizer (imported in the build container) for a handful of configs - integer and fractional warm-up /
cool-down lengths, lr_end and lr_end_pct, steps beyond the budget - and frozen as data: the learning rate before the first step and after each
of `n` scheduler steps, plus the state_dict after the last one.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_schedules.py      (writes tests/golden/schedules.json)
"""
import json
import os
import sys
from types import SimpleNamespace

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _install_stubs  # noqa: E402

CASES = [
  dict(scheduler='warmup_cosine', lr=3e-3, lr_start=0.0, lr_end=1e-5, lr_end_pct=None, warmup_steps=2, cooldown_steps=None, steps_budget=8, n=11),
  dict(scheduler='warmup_cosine', lr=1e-3, lr_start=1e-5, lr_end=None, lr_end_pct=0.1, warmup_steps=0.1, cooldown_steps=None, steps_budget=50, n=55),
  dict(scheduler='warmup_constant', lr=2e-3, lr_start=1e-4, lr_end=None, lr_end_pct=None, warmup_steps=5, cooldown_steps=None, steps_budget=20, n=24),
  dict(scheduler='wsd', lr=3e-3, lr_start=0.0, lr_end=1e-5, lr_end_pct=None, warmup_steps=3, cooldown_steps=4, steps_budget=16, n=18),
  dict(scheduler='wsd', lr=6e-4, lr_start=6e-6, lr_end=None, lr_end_pct=0.01, warmup_steps=0.05, cooldown_steps=0.2, steps_budget=100, n=104),
  dict(scheduler='linear_cooldown', lr=3e-3, lr_start=0.0, lr_end=1e-5, lr_end_pct=None, warmup_steps=None, cooldown_steps=6, steps_budget=30, resume_step=4, n=14),
  dict(scheduler='linear_cooldown', lr=1e-3, lr_start=0.0, lr_end=None, lr_end_pct=0.1, warmup_steps=None, cooldown_steps=0.25, steps_budget=40, resume_step=30, n=42),
]


def main():
  _install_stubs()
  sys.path.insert(0, REF)
  from optim import initialize_scheduler
  out = []
  for case in CASES:
    cfg = SimpleNamespace(**{k: v for k, v in case.items() if k != 'n'})
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=cfg.lr)
    sch = initialize_scheduler(opt, cfg)
    lrs = [opt.param_groups[0]['lr']]
    for _ in range(case['n']):
      sch.step()
      lrs.append(opt.param_groups[0]['lr'])
    out.append({'cfg': case, 'lrs': lrs, 'state': sch.state_dict()})
  with open(os.path.join(HERE, 'schedules.json'), 'w') as f:
    json.dump(out, f, indent=1)
  print('wrote', len(out), 'cases')


if __name__ == '__main__':
  main()

"""cfg-1 at its literal shape (SURVEY.md section 8c's fixture list; BASELINE.json configs[0]): the REFERENCE engine on the CPU - 2 layers,
d = 128, 2 heads, seq_len 128, the real vocabulary (50 280), micro_batch_size 1 x grad_accumulation_steps 4, 6 optimizer steps
(= steps_budget; config/config.yaml's optimizer / schedule keys otherwise: lr 3e-3, cosine to 1e-5, clip 1.0, AdamW 0.9 / 0.95 / 0.1; the
warm-up is 2 steps - the yaml's fraction 0.1 of a 6-step budget would be none) - imported in the build container, its 24 micro-step losses, 6 learning rates, per-step parameter norms and slices of the final weights
frozen as data.  The initial weights (2 x 50 280 x 128 floats: too big to commit) are oracle.cpu_ref.init_params(seed 7) on both sides; the
fixture carries their checksums so a drifting RNG is noticed instead of mis-read as a parity failure.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_cfg1.py      (writes tests/golden/cfg1.npz)
"""
import os
import sys
from collections import namedtuple

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, ROOT, _install_stubs  # noqa: E402

CFG1 = dict(model='transformer', vocab_size=50280, seq_len=128, d_model=128, expand='8/3', n_layers=2, n_heads=2,
            mlp_class='glu', tie_embeddings=False, torch_compile=False, micro_batch_size=1,
            grad_accumulation_steps=4, dtype='bfloat16', optim='adamw', fused_optim=False, lr=3e-3,
            weight_decay=0.1, beta1=0.9, beta2=0.95, grad_clip=1.0, scheduler='warmup_cosine', warmup_steps=2,
            cooldown_steps=None, lr_start=0.0, lr_end=1e-5, lr_end_pct=None, steps_budget=6, resume=False, seed=100)
INIT_SEED, N_OPT = 7, 6
SLICES = {'layers.1.mlp.fc2.weight': None, 'out_norm.weight': None, 'layers.0.attn_norm.weight': None, 'layers.0.attn.w_qkv.weight': (0, 32)}


def checksums(params):
  return np.array([[float(p.double().sum()), float(p.double().abs().sum())] for p in params.values()])


def main():
  _install_stubs()
  sys.path.insert(0, REF)
  sys.path.insert(1, ROOT)
  from models import construct_model
  from engine.engine import TorchEngine
  from oracle import cpu_ref as O
  ocfg = O.OracleConfig(vocab_size=50280, seq_len=128, dim=128, n_layers=2, n_heads=2)
  weights = O.init_params(ocfg, seed=INIT_SEED)
  ecfg = namedtuple('Config', CFG1.keys())(**CFG1)
  torch.manual_seed(100)
  model, _ = construct_model(ecfg)
  model.load_state_dict(weights)
  eng = TorchEngine(model, ecfg, 'cpu', None, None)
  rng = np.random.default_rng(4321)
  tok = torch.from_numpy(rng.integers(0, 50280, size=(N_OPT * 4, 1, 129)))
  losses, lrs, norms = [], [], []
  for i in range(N_OPT * 4):
    losses.append(eng.step({'input_ids': tok[i]}).item())
    if (i + 1) % 4 == 0:
      lrs.append(eng.optimizer.param_groups[0]['lr'])
      norms.append([p.detach().norm().item() for _, p in model.named_parameters()])
  final = {n: p.detach().clone() for n, p in model.named_parameters()}
  used = torch.unique(tok[:, 0, :128])[:64]
  out = {'tokens': tok, 'losses': np.array(losses), 'lrs': np.array(lrs), 'param_norms': np.array(norms), 'init_checksums': checksums(weights),
         'rows': used, 'final:embed_rows': final['embed_tokens.weight'][used], 'final:lm_head_rows': final['lm_head.weight'][used]}
  for n, sl in SLICES.items():
    out['final:' + n] = final[n] if sl is None else final[n][sl[0]:sl[1]]
  np.savez_compressed(os.path.join(HERE, 'cfg1.npz'), **{k: np.asarray(v) for k, v in out.items()})
  # the oracle's own engine on the same run, printed at generation time (tests/test_oracle_golden.py re-checks it on every CPU run)
  oe = O.OracleEngine(weights, ocfg, lr=3e-3, weight_decay=0.1, beta1=0.9, beta2=0.95, grad_clip=1.0, accum=4, steps_budget=6, warmup_steps=2)
  ol = [oe.step({'input_ids': tok[i]}).item() for i in range(N_OPT * 4)]
  print('reference losses', np.round(losses, 5))
  print('oracle - reference (max rel)', float(np.max(np.abs(np.array(ol) - np.array(losses)) / np.array(losses))))
  print('lrs', lrs)
  print('cfg1.npz', os.path.getsize(os.path.join(HERE, 'cfg1.npz')) // 1024, 'KiB')


if __name__ == '__main__':
  main()

"""The reference's two plain MLP classes (models/transformer.py:26 MLP_CLASSES: 'mlp' = fc2(silu(fc1 x)), 'mlp_relu_sq' = fc2(relu(fc1 x)^2);
models/components.py:31-40, :59-70) - off every shipped config's path, but part of the module API: the REFERENCE model (imported in the build
container, CPU fp32) on a tiny config, its loss and all 15 gradients frozen as data, the weights being oracle.cpu_ref.init_params(seed 3) on
both sides (checksums in the fixture).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_mlp_classes.py      (writes tests/golden/mlp_classes.npz)
"""
import os
import sys

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, ROOT, _install_stubs  # noqa: E402


def main():
  _install_stubs()
  sys.path.insert(0, REF)
  sys.path.insert(1, ROOT)
  from models.transformer import Transformer, ModelConfig
  from oracle import cpu_ref as O
  rng = np.random.default_rng(77)
  tok = torch.from_numpy(rng.integers(0, 256, size=(2, 65)))
  ids, tgt = tok[:, :64], tok[:, 1:]
  out = {'tokens': tok}
  for kind in ('mlp', 'mlp_relu_sq'):
    ocfg = O.OracleConfig(vocab_size=256, seq_len=64, dim=128, n_layers=2, n_heads=2, mlp=kind)
    w = O.init_params(ocfg, seed=3)
    m = Transformer(ModelConfig(vocab_size=256, seq_len=64, dim=128, expand=8 / 3, n_layers=2, n_heads=2, mlp=kind))
    m.load_state_dict(w)
    logits = m(ids, None)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 256), tgt.reshape(-1))
    loss.backward()
    out[f'{kind}:loss'] = loss.detach()
    out[f'{kind}:logits'] = logits.detach()
    out[f'{kind}:init_checksums'] = np.array([[float(p.double().sum()), float(p.double().abs().sum())] for p in w.values()])
    for n, p in m.named_parameters():
      out[f'{kind}:g:{n}'] = p.grad.detach()
    ol, og = O.loss_and_grads(w, ocfg, ids, tgt)
    err = max(((og[n] - p.grad).abs().max() / p.grad.abs().max()).item() for n, p in m.named_parameters())
    print(kind, 'loss', loss.item(), 'oracle - reference: loss', abs(ol.item() - loss.item()) / loss.item(), 'worst gradient (rel-to-max)', err)
  np.savez_compressed(os.path.join(HERE, 'mlp_classes.npz'), **{k: np.asarray(v) for k, v in out.items()})
  print('mlp_classes.npz', os.path.getsize(os.path.join(HERE, 'mlp_classes.npz')) // 1024, 'KiB')


if __name__ == '__main__':
  main()

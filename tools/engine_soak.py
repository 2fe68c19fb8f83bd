"""Soak of the drop-in engine: N optimizer steps of `plainlm_amd.TorchEngine.step` on the 160M config (config_doc_mask.yaml's shape by default:
micro-batch 8, accumulation 2, document masks with fresh random documents every micro-step - a new plan, new host batches) and what a long run
must keep flat: device memory (allocated / reserved), host RSS, step time; plus the loss on a FIXED small pool of batches, which must fall.

  python tools/engine_soak.py [--steps 300] [--micro-batch 8] [--accum 2] [--no-doc-mask]
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import plainlm_amd as P  # noqa: E402


def rss_mb():
  with open('/proc/self/status') as f:
    for l in f:
      if l.startswith('VmRSS:'):
        return int(l.split()[1]) / 1024.0
  return float('nan')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=300, help='optimizer steps')
  ap.add_argument('--accum', type=int, default=2)
  ap.add_argument('--micro-batch', type=int, default=8)
  ap.add_argument('--no-doc-mask', action='store_true')
  a = ap.parse_args()
  T = 1024
  cfg = SimpleNamespace(model='transformer', vocab_size=50280, d_model=768, expand='8/3', n_layers=12, n_heads=12, mlp_class='glu',
                        seq_len=T, tie_embeddings=False, dtype='bfloat16', optim='adamw', fused_optim=True, lr=3e-4, beta1=0.9,
                        beta2=0.95, weight_decay=0.1, eps=1e-8, scheduler='warmup_cosine', warmup_steps=20, lr_start=0.0, lr_end=1e-5,
                        lr_end_pct=None, steps_budget=a.steps, grad_accumulation_steps=a.accum, grad_clip=1.0,
                        intra_doc_masking=not a.no_doc_mask, resume=False, seed=100, micro_batch_size=a.micro_batch)
  torch.manual_seed(cfg.seed)
  model, _ = P.construct_model(cfg)
  eng = P.TorchEngine(model, cfg, 'cuda', None, None)
  rng = np.random.default_rng(0)
  B = a.micro_batch
  pool = rng.integers(0, 512, size=(8, B, T + 1))  # a small fixed pool over 512 token ids: the loss must fall well below ln(50280)

  def batch(i):
    bt = {'input_ids': torch.from_numpy(pool[i % 8])}
    if not a.no_doc_mask:  # fresh documents every micro-step: every step builds a new plan
      rows = []
      for _ in range(B):
        lens, tot = [], 0
        while tot < T + 1:
          n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
          lens.append(n)
          tot += n
        rows.append(lens)
      bt['docs_lengths'] = rows
    return bt

  recs = []
  t_last = time.perf_counter()
  n_micro = a.steps * a.accum
  every = max(1, n_micro // 10)
  for i in range(n_micro):
    loss = eng.step(batch(i))
    if (i + 1) % every == 0:
      torch.cuda.synchronize()
      now = time.perf_counter()
      recs.append({'micro_step': i + 1, 'loss': round(float(loss), 4), 'ms_per_micro_step': round(1e3 * (now - t_last) / every, 3),
                   'dev_alloc_mb': round(torch.cuda.memory_allocated() / 2**20, 1), 'dev_reserved_mb': round(torch.cuda.memory_reserved() / 2**20, 1),
                   'host_rss_mb': round(rss_mb(), 1)})
      print(recs[-1], flush=True)
      t_last = time.perf_counter()
  first, last = recs[1], recs[-1]  # the first record includes warm-up allocations
  ok = (last['dev_alloc_mb'] <= first['dev_alloc_mb'] * 1.01 + 1 and last['dev_reserved_mb'] <= first['dev_reserved_mb'] * 1.05 + 64
        and last['host_rss_mb'] <= first['host_rss_mb'] * 1.03 + 64 and last['loss'] < recs[0]['loss'])
  print({'flat_memory_and_falling_loss': ok, 'first': first, 'last': last})
  sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()

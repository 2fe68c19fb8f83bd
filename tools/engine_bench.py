"""Throughput of the drop-in engine (`plainlm_amd.TorchEngine.step`, the call train.py:73 makes) on the 160M config:
host batches in, loss tensor out, clip + AdamW + LR schedule at every window end.  Compares with bench.py's raw
fwd+bwd figure to show what the engine layer (host->device copy, NaN check sync, optimizer) costs."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import plainlm_amd as P  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--accum', type=int, default=1)
  ap.add_argument('--micro-batch', type=int, default=32)
  ap.add_argument('--doc-mask', action='store_true', help="intra_doc_masking with random documents (mean length 256): config_doc_mask.yaml is --micro-batch 8 --accum 2 --doc-mask")
  ap.add_argument('--nan-check-lag', type=int, default=None, help='0 = the reference order (host read before backward); default: the engine default')
  a = ap.parse_args()
  cfg = SimpleNamespace(model='transformer', vocab_size=50280, d_model=768, expand='8/3', n_layers=12, n_heads=12, mlp_class='glu',
                        seq_len=1024, tie_embeddings=False, dtype='bfloat16', optim='adamw', fused_optim=True, lr=3e-4, beta1=0.9,
                        beta2=0.95, weight_decay=0.1, eps=1e-8, scheduler='warmup_cosine', warmup_steps=10, lr_start=0.0, lr_end=1e-5,
                        lr_end_pct=None, steps_budget=1000, grad_accumulation_steps=a.accum, grad_clip=1.0, intra_doc_masking=a.doc_mask,
                        resume=False, seed=100, micro_batch_size=a.micro_batch)
  if a.nan_check_lag is not None:
    cfg.nan_check_lag = a.nan_check_lag
  torch.manual_seed(cfg.seed)
  model, _ = P.construct_model(cfg)
  eng = P.TorchEngine(model, cfg, 'cuda', None, None)
  rng = np.random.default_rng(0)
  B = a.micro_batch
  batches = [{'input_ids': torch.from_numpy(rng.integers(0, cfg.vocab_size, size=(B, 1025)))} for _ in range(4)]
  if a.doc_mask:  # docs_lengths per row, summing to T + 1 (data_prep_utils.py:52-77)
    for bt in batches:
      rows = []
      for _ in range(B):
        lens, tot = [], 0
        while tot < 1025:
          n = int(min(rng.geometric(1.0 / 256.0), 1025 - tot))
          lens.append(n)
          tot += n
        rows.append(lens)
      bt['docs_lengths'] = rows
  # 24 untimed steps (the allocator and the runtime settle in the first dozen)
  for i in range(24):
    eng.step(batches[i % 4])
  torch.cuda.synchronize()
  if os.environ.get('PLM_SYNC_DEBUG'):
    torch.cuda.set_sync_debug_mode('warn')
  t0 = time.perf_counter()
  host = 0.0
  for i in range(a.steps * a.accum):
    h0 = time.perf_counter()
    loss = eng.step(batches[i % 4])
    host += time.perf_counter() - h0
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  n = a.steps * a.accum
  print({'engine_ms_per_micro_step': round(1e3 * dt / n, 3), 'tokens_per_s': round(B * 1024 * n / dt, 1), 'micro_batch': B, 'accum': a.accum, 'doc_mask': a.doc_mask,
         'ms_inside_step_calls_per_micro_step': round(1e3 * host / n, 3), 'nan_check_lag': eng.nan_check_lag, 'loss': float(loss)})


if __name__ == '__main__':
  main()

"""Host-side profile of TorchEngine.step (is the drop-in engine launch-bound?): cProfile over a few steps, top entries by
cumulative and by own time (Event.synchronize = time the host waits for the GPU: the engine is GPU-bound when it dominates)."""
import cProfile
import io
import os
import pstats
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import plainlm_amd as P  # noqa: E402


def main():
  cfg = SimpleNamespace(model='transformer', vocab_size=50280, d_model=768, expand='8/3', n_layers=12, n_heads=12, mlp_class='glu',
                        seq_len=1024, tie_embeddings=False, dtype='bfloat16', optim='adamw', fused_optim=True, lr=3e-4, beta1=0.9,
                        beta2=0.95, weight_decay=0.1, eps=1e-8, scheduler='warmup_cosine', warmup_steps=10, lr_start=0.0, lr_end=1e-5,
                        lr_end_pct=None, steps_budget=1000, grad_accumulation_steps=1, grad_clip=1.0, intra_doc_masking=False,
                        resume=False, seed=100, micro_batch_size=32)
  torch.manual_seed(cfg.seed)
  model, _ = P.construct_model(cfg)
  eng = P.TorchEngine(model, cfg, 'cuda', None, None)
  rng = np.random.default_rng(0)
  batches = [{'input_ids': torch.from_numpy(rng.integers(0, cfg.vocab_size, size=(32, 1025)))} for _ in range(4)]
  for i in range(6):
    eng.step(batches[i % 4])
  torch.cuda.synchronize()
  pr = cProfile.Profile()
  pr.enable()
  for i in range(10):
    eng.step(batches[i % 4])
  pr.disable()
  torch.cuda.synchronize()
  for key in ('cumulative', 'tottime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == '__main__':
  main()

"""GPU-side duration of every TorchEngine.step over a longer run (events around each call): trend, spikes, allocator events."""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import plainlm_amd as P  # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
  fresh = 'fresh' in sys.argv[2:]  # a fresh random batch every step instead of four repeating ones
  if 'nogc' in sys.argv[2:]:
    import gc
    gc.disable()
  cfg = SimpleNamespace(model='transformer', vocab_size=50280, d_model=768, expand='8/3', n_layers=12, n_heads=12, mlp_class='glu',
                        seq_len=1024, tie_embeddings=False, dtype='bfloat16', optim='adamw', fused_optim=True, lr=3e-4, beta1=0.9,
                        beta2=0.95, weight_decay=0.1, eps=1e-8, scheduler='warmup_cosine', warmup_steps=10, lr_start=0.0, lr_end=1e-5,
                        lr_end_pct=None, steps_budget=1000, grad_accumulation_steps=1, grad_clip=1.0, intra_doc_masking=False,
                        resume=False, seed=100, micro_batch_size=32)
  torch.manual_seed(0)
  model, _ = P.construct_model(cfg)
  eng = P.TorchEngine(model, cfg, 'cuda', None, None)
  rng = np.random.default_rng(0)
  batches = [{'input_ids': torch.from_numpy(rng.integers(0, cfg.vocab_size, size=(32, 1025)))} for _ in range(4)]
  if 'nosync' in sys.argv[2:]:  # what-if: no host wait inside step() at all (flags checked 4 micro-steps late)
    eng.nan_check_lag = 4
    orig = eng.check_losses
    eng.check_losses = lambda keep=4: orig(keep=max(keep, 4))
  for i in range(5):
    eng.step(batches[i % 4])
  torch.cuda.synchronize()
  if 'freeze' in sys.argv[2:]:
    import gc
    gc.collect()
    gc.freeze()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
  losses = []
  for i in range(n):
    b = {'input_ids': torch.from_numpy(rng.integers(0, cfg.vocab_size, size=(32, 1025)))} if fresh else batches[i % 4]
    ev[i][0].record()
    losses.append(eng.step(b))
    ev[i][1].record()
  torch.cuda.synchronize()
  t = [a.elapsed_time(b) for a, b in ev]
  import gc
  print('gc enabled', gc.isenabled(), 'gc counts', gc.get_count(), 'gc stats', gc.get_stats())
  st = torch.cuda.memory_stats()
  print({'steps': n, 'fresh_batches': fresh, 'mean_ms': round(sum(t) / n, 2), 'median_ms': round(sorted(t)[n // 2], 2), 'max_ms': round(max(t), 1),
         'first10': [round(x, 1) for x in t[:10]], 'last10': [round(x, 1) for x in t[-10:]], 'spikes>40ms': [(i, round(x, 1)) for i, x in enumerate(t) if x > 40],
         'loss_first_last': (round(float(losses[0]), 3), round(float(losses[-1]), 3)), 'device_allocs': st['num_device_alloc'], 'alloc_retries': st['num_alloc_retries']})


if __name__ == '__main__':
  main()

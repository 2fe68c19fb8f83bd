"""How long does the HOST need to enqueue one fwd+bwd step (Python + launches), next to what the GPU needs to run it?  The host runs ahead of the
GPU, so the wall time of a few steps WITHOUT a synchronisation at the end is the enqueue time (as long as it is shorter than the GPU time and
the runtime's queue does not fill).  Usage (GPU box): python tools/host_enqueue_time.py [--micro-batch 8] [--doc-mask] [--config 420m]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--config', default='160m')
  ap.add_argument('--micro-batch', type=int, default=0)
  ap.add_argument('--doc-mask', action='store_true')
  a = ap.parse_args()
  c = dict(bench.CONFIGS[a.config])
  if a.micro_batch:
    c['micro_batch'] = a.micro_batch
  B, T, V = c['micro_batch'], c['seq_len'], c['vocab_size']
  dev = torch.device('cuda', 0)
  model = bench.build_model(c, dev)
  model.enable_main_grad()
  rng = np.random.default_rng(0)
  tok = torch.from_numpy(rng.integers(0, V, size=(B, T + 1)))
  ids, tgt = tok[:, :T].contiguous().to(dev), tok[:, 1:].contiguous().to(dev)
  ds = None
  if a.doc_mask:
    from plainlm_amd.engine import doc_start_from_lengths
    docs = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
    ds = doc_start_from_lengths(docs, T).to(dev)

  def step():
    model.sink.begin_window()
    model.invalidate_shadows()
    model.loss(ids, tgt, ds).backward()

  for _ in range(6):
    step()
  torch.cuda.synchronize()
  res = []
  for n in (3, 3, 3):
    t0 = time.perf_counter()
    for _ in range(n):
      step()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    res.append((round(1e3 * t_host, 2), round(1e3 * t_all, 2)))
  print({'config': a.config, 'micro_batch': B, 'doc_mask': a.doc_mask, 'host_enqueue_ms_per_step / wall_ms_per_step (3 steps, GPU idle at start)': res})


if __name__ == '__main__':
  main()

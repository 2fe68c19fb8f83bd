"""Does a hipGraph of the whole fwd+bwd step run faster than the eager launches?  torch.cuda.CUDAGraph captures every kernel launched on the capturing
stream - including the ctypes launches of libplainlm_hip.so, which take torch's current stream.  Usage (GPU box): python tools/graph_probe.py [--micro-batch 8] [--doc-mask]"""
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
  ap.add_argument('--steps', type=int, default=20)
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
    loss = model.loss(ids, tgt, ds)
    loss.backward()
    return loss

  def timeit(fn, n):
    for _ in range(5):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

  for _ in range(6):
    step()
  torch.cuda.synchronize()
  eager = [timeit(step, a.steps) for _ in range(2)]
  ref_loss = float(step())
  ref_grad = model._flat_grad.clone()
  g = torch.cuda.CUDAGraph()
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    for _ in range(3):
      step()
  torch.cuda.current_stream().wait_stream(s)
  torch.cuda.synchronize()
  with torch.cuda.graph(g):
    gl = step()
  torch.cuda.synchronize()
  graph = [timeit(g.replay, a.steps) for _ in range(2)]
  g.replay()
  torch.cuda.synchronize()
  same = bool(torch.equal(model._flat_grad, ref_grad)) and float(gl) == ref_loss
  eager2 = [timeit(step, a.steps)]
  print({'config': a.config, 'micro_batch': B, 'doc_mask': a.doc_mask, 'eager_ms': [round(x, 3) for x in eager + eager2], 'graph_replay_ms': [round(x, 3) for x in graph],
         'graph_result_bit_equal_to_eager': same})


if __name__ == '__main__':
  main()

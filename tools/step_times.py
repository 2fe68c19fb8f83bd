"""Wall time of every single fwd+bwd step (synchronised), to see spikes the bench's mean hides.
  python tools/step_times.py [--doc-mask] [--micro-batch 8] [--steps 40]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--doc-mask', action='store_true')
  ap.add_argument('--micro-batch', type=int, default=32)
  ap.add_argument('--steps', type=int, default=40)
  a = ap.parse_args()
  from plainlm_amd.engine import doc_start_from_lengths
  c = dict(Bn.CONFIGS['160m'])
  B, T, V = a.micro_batch, c['seq_len'], c['vocab_size']
  dev = torch.device('cuda', 0)
  model = Bn.build_model(c, dev)
  model.enable_main_grad()
  rng = np.random.default_rng(1234)
  tok = torch.from_numpy(rng.integers(0, V, size=(B, T + 1)))
  ids, tgt = tok[:, :T].contiguous().to(dev), tok[:, 1:].contiguous().to(dev)
  ds = None
  if a.doc_mask:
    docs = []
    for _ in range(B):
      lens, tot = [], 0
      while tot < T + 1:
        n = int(min(rng.geometric(1.0 / 256.0), T + 1 - tot))
        lens.append(n)
        tot += n
      docs.append(lens)
    ds = doc_start_from_lengths(docs, T).to(dev)
  ts = []
  for i in range(a.steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.sink.begin_window()
    model.invalidate_shadows()
    loss = model.loss(ids, tgt, ds)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ts.append((1e3 * (t3 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
  for i, (tot, hf, hb) in enumerate(ts):
    print(f'step {i:2d}: {tot:7.3f} ms   host enqueue fwd {hf:6.3f} bwd {hb:6.3f}')


if __name__ == '__main__':
  main()

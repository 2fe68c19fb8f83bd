"""Average board power and shader clock over whole training steps (fwd+bwd of the bench's model) next to the same readings for a loop of one
GEMM and a loop of one HBM-bound kernel: is the 1400 W limit met by the step's AVERAGE or by each kernel on its own?
Usage (GPU box): python tools/step_power.py [--seconds 6]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import power_probe as pp  # noqa: E402
import bench  # noqa: E402
from plainlm_amd import ops  # noqa: E402


def sample(name, fn, seconds, files, unit_work=None):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  smp = pp.Sampler(files, period=0.01)
  t0 = time.time()
  smp.start()
  n = 0
  while time.time() - t0 < seconds:
    fn()
    n += 1
    if n % 4 == 0:
      torch.cuda.synchronize()
  torch.cuda.synchronize()
  dt = time.time() - t0
  smp.stop_flag = True
  smp.join()
  out = {'case': name, 'ms_per_call': round(1e3 * dt / n, 3), **smp.summary(skip=0.4)}
  print(json.dumps(out), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--seconds', type=float, default=6.0)
  a = ap.parse_args()
  files = pp.hwmon_files()
  dev = torch.device('cuda', 0)
  c = dict(bench.CONFIGS['160m'])
  B, T, V = c['micro_batch'], c['seq_len'], c['vocab_size']
  model = bench.build_model(c, dev)
  model.enable_main_grad()
  rng = np.random.default_rng(0)
  tok = torch.from_numpy(rng.integers(0, V, size=(B, T + 1))).to(dev)
  ids, tgt = tok[:, :T].contiguous(), tok[:, 1:].contiguous()

  def step():
    model.sink.begin_window()
    model.invalidate_shadows()
    model.loss(ids, tgt, None).backward()
    model.sink.flush_dw()
  sample('training step fwd+bwd (160M, B=32)', step, a.seconds, files)
  M, d, h = B * T, 768, 2048
  A = torch.randn(M, 4096, device=dev).bfloat16(); W = (torch.randn(d, 4096, device=dev) * 0.02).bfloat16(); O = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
  sample('loop of nt dX fc1 (K = 4096)', lambda: ops.gemm_nt(A, W, out=O), a.seconds, files)
  x = torch.randn(M, d, device=dev); w = torch.ones(d, device=dev); br = torch.randn(M, d, device=dev).bfloat16()
  sample('loop of add + rmsnorm fwd', lambda: ops.rmsnorm_fwd(x, w, 1e-6, branch=br), a.seconds, files)
  sample('training step fwd+bwd again', step, a.seconds, files)


if __name__ == '__main__':
  main()

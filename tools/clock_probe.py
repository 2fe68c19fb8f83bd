"""Sample rocm-smi clocks while one GEMM variant loops (diagnostic: is a timing difference a clock difference?).
usage: clock_probe.py <variant> [n] [seconds]"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from plainlm_amd import ops  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 9
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
BF = torch.bfloat16
A = (torch.rand(n, n, device='cuda') * 2 - 1).to(BF)
B = (torch.rand(n, n, device='cuda') * 2 - 1).to(BF)
out = torch.empty(n, n, device='cuda', dtype=BF)
stop, clocks = False, []


def sampler():
  while not stop:
    r = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True).stdout
    for l in r.splitlines():
      if 'sclk' in l:
        clocks.append(int(l.split('(')[1].split('Mhz')[0]))
    time.sleep(0.2)


for _ in range(50):
  ops.gemm_nt(A, B, out=out, variant=variant)
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
time.sleep(0.5)
clocks.clear()
t0 = time.time()
it = 0
while time.time() - t0 < secs:
  for _ in range(100):
    ops.gemm_nt(A, B, out=out, variant=variant)
  torch.cuda.synchronize()
  it += 100
dt = time.time() - t0
mid = list(clocks)
stop = True
th.join()
tf = 2.0 * n ** 3 * it / dt / 1e12
clk = sum(mid) / max(1, len(mid))
print({'variant': variant, 'xp': os.environ.get('PLM_W4_XP'), 'TFLOP/s': round(tf, 1), 'sclk_MHz_avg': round(clk), 'samples': len(mid),
       'MFMA_util_at_clock': round(tf / (2500.0 * clk / 2400.0), 3)})

"""Is the MFMA plateau an issue limit or the board's power limit?  Loops one kernel for a few seconds per case while a thread
samples the GPU's hwmon files (socket power, shader clock) and prints TFLOP/s next to average watts / MHz.  The same GEMM is run
on random operands, on zeros and on a constant: equal instruction streams, different switching activity - if the zero run is
much faster at a higher clock, the random run is clock-limited by power, not by what the kernel issues.
Usage: python tools/power_probe.py [--seconds 2.5]   (ordinary user; reads /sys/class/drm/card*/device/hwmon and rocm-smi)"""

import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plainlm_amd import ops  # noqa: E402

BF = torch.bfloat16


def device_bus_id():
  """PCI bus id of HIP device 0 ('0000:c1:00.0'): the box has eight GPUs and /sys lists all of them."""
  import ctypes
  hip = ctypes.CDLL('libamdhip64.so')
  buf = ctypes.create_string_buffer(64)
  if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0:
    return None
  return buf.value.decode().lower()


def hwmon_files():
  out = {}
  bus = device_bus_id()
  for card in sorted(glob.glob('/sys/class/drm/card*/device')):
    if bus and not os.path.realpath(card).lower().endswith(bus):
      continue
    for hm in glob.glob(card + '/hwmon/hwmon*'):
      for key in ('power1_average', 'power1_input', 'freq1_input', 'power1_cap'):
        p = os.path.join(hm, key)
        if os.path.exists(p):
          out.setdefault(card, {})[key] = p
    p = os.path.join(card, 'pp_dpm_sclk')
    if os.path.exists(p):
      out.setdefault(card, {})['pp_dpm_sclk'] = p
  return out


def read(path):
  try:
    return open(path).read().strip()
  except OSError:
    return None


class Sampler(threading.Thread):
  def __init__(self, files, period=0.02):
    super().__init__(daemon=True)
    self.files, self.period, self.rows, self.stop_flag = files, period, [], False

  def run(self):
    while not self.stop_flag:
      row = {}
      for card, f in self.files.items():
        for k in ('power1_average', 'power1_input', 'freq1_input'):
          if k in f:
            v = read(f[k])
            if v and v.lstrip('-').isdigit():
              row[k] = max(row.get(k, 0), int(v))  # the busy card is the one that counts
        if 'pp_dpm_sclk' in f:
          v = read(f['pp_dpm_sclk'])
          if v:
            for line in v.splitlines():
              if '*' in line:
                try:
                  row['dpm_sclk_mhz'] = max(row.get('dpm_sclk_mhz', 0), int(line.split(':')[1].strip().split('M')[0]))
                except (ValueError, IndexError):
                  pass
      self.rows.append(row)
      time.sleep(self.period)

  def summary(self, skip=0.3):
    rows = self.rows[int(len(self.rows) * skip):]
    out = {'samples': len(rows)}
    for k in ('power1_average', 'power1_input'):
      v = [r[k] for r in rows if k in r]
      if v:
        out[k + '_W'] = round(sum(v) / len(v) / 1e6, 1)
        out[k + '_max_W'] = round(max(v) / 1e6, 1)
    v = [r['freq1_input'] for r in rows if 'freq1_input' in r]
    if v:
      out['sclk_MHz'] = round(sum(v) / len(v) / 1e6, 0)
      out['sclk_min_MHz'] = round(min(v) / 1e6, 0)
    v = [r['dpm_sclk_mhz'] for r in rows if 'dpm_sclk_mhz' in r]
    if v:
      out['dpm_sclk_MHz'] = round(sum(v) / len(v), 0)
    return out


def smi(args):
  try:
    return subprocess.run(args, capture_output=True, text=True, timeout=20).stdout.strip()
  except Exception as e:  # noqa: BLE001
    return f'({args[0]}: {e})'


def run_case(name, fn, flops, seconds, files):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  fn()
  e.record()
  torch.cuda.synchronize()
  n = max(8, int(seconds * 1e3 / max(s.elapsed_time(e), 1e-3)))
  smp = Sampler(files)
  smp.start()
  s.record()
  for _ in range(n):
    fn()
  e.record()
  torch.cuda.synchronize()
  smp.stop_flag = True
  smp.join()
  ms = s.elapsed_time(e) / n
  row = {'case': name, 'ms': round(ms, 4), 'launches': n}
  if flops:
    row['TFLOP/s'] = round(flops / ms / 1e9, 1)
  row.update(smp.summary())
  print(json.dumps(row), flush=True)
  return row


def operands(kind, m, k, dev):
  if kind == 'randn':
    return torch.randn(m, k, device=dev).to(BF)
  if kind == 'zeros':
    return torch.zeros(m, k, device=dev, dtype=BF)
  if kind == 'ones':
    return torch.ones(m, k, device=dev, dtype=BF)
  if kind == 'pm1':  # +-1: one sign bit toggles, exponent / mantissa constant
    return (torch.randint(0, 2, (m, k), device=dev) * 2 - 1).to(BF)
  if kind == 'small':  # weights-like magnitudes
    return (torch.randn(m, k, device=dev) * 0.02).to(BF)
  raise ValueError(kind)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--seconds', type=float, default=2.5)
  a = ap.parse_args()
  dev = 'cuda'
  torch.cuda.init()
  files = hwmon_files()
  print(json.dumps({'hwmon': {c: sorted(f) for c, f in files.items()}, 'power_cap': {c: read(f['power1_cap']) for c, f in files.items() if 'power1_cap' in f}}), flush=True)
  print(smi(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower']), flush=True)
  M, d, h, V = 32768, 768, 2048, 50304
  time.sleep(1.0)
  idle = Sampler(files)
  idle.start()
  time.sleep(1.0)
  idle.stop_flag = True
  idle.join()
  print(json.dumps({'case': 'idle', **idle.summary(0.0)}), flush=True)
  for shape_name, (m, n, k) in {'nt dX head': (M, d, V), 'nt fc1 fwd': (M, 2 * h, d), 'nt 8192^3': (8192, 8192, 8192)}.items():
    out = torch.empty(m, n, device=dev, dtype=BF)
    for kind in ('randn', 'small', 'pm1', 'ones', 'zeros'):
      A, Bm = operands(kind, m, k, dev), operands(kind, n, k, dev)
      run_case(f'{shape_name} [{kind}]', lambda: ops.gemm_nt(A, Bm, out=out), 2.0 * m * n * k, a.seconds, files)
      del A, Bm
    A, Bm = operands('randn', m, k, dev), operands('zeros', n, k, dev)
    run_case(f'{shape_name} [A randn, B zeros]', lambda: ops.gemm_nt(A, Bm, out=out), 2.0 * m * n * k, a.seconds, files)
    del A, Bm
    # the vendor library under the same cap (comparator only: torch.matmul -> hipBLASLt; never on the product path)
    for kind in ('randn', 'zeros'):
      A, Bm = operands(kind, m, k, dev), operands(kind, n, k, dev)
      run_case(f'{shape_name} [{kind}] torch.matmul (hipBLASLt)', lambda: torch.matmul(A, Bm.t(), out=out), 2.0 * m * n * k, a.seconds, files)
      del A, Bm
    del out
  m, n, k = V, d, M
  for kind in ('randn', 'zeros'):
    A, Bm = operands(kind, k, m, dev), operands(kind, k, n, dev)
    out = torch.zeros(m, n, device=dev)
    run_case(f'tn dW head [{kind}]', lambda: ops.gemm_tn(A, Bm, out=out, accumulate=True), 2.0 * m * n * k, a.seconds, files)
    del A, Bm, out
  # attention and an HBM-bound kernel for comparison
  from plainlm_amd.transformer import rope_tables
  B, T, nh = 32, 1024, 12
  cos, sin = (t.to(dev) for t in rope_tables(64, T))
  for kind in ('randn', 'zeros'):
    qkv = operands(kind, M, 3 * d, dev)
    dout = operands(kind, M, d, dev)
    o, lse = ops.attn_fwd(qkv, B, T, nh)
    fl = 2.0 * 2 * B * nh * T * (T + 1) / 2 * 64
    run_case(f'attn fwd [{kind}]', lambda: ops.attn_fwd(qkv, B, T, nh), fl, a.seconds, files)
    run_case(f'attn bwd [{kind}]', lambda: ops.attn_bwd(qkv, o, dout, lse, cos, sin, B, T, nh), 2.0 * fl, a.seconds, files)
  big = torch.randn(M, 2 * h, device=dev).to(BF)
  r = run_case('swiglu fwd (HBM bound)', lambda: ops.swiglu_fwd(big), None, a.seconds, files)
  print(json.dumps({'swiglu GB/s': round(M * 2 * h * 2 * 1.5 / r['ms'] / 1e6, 1)}))
  print(smi(['rocm-smi', '--showpower', '--showclocks']), flush=True)


if __name__ == '__main__':
  main()

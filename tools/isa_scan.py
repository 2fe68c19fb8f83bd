"""What did hipcc do to the hot loops?  Compiles one csrc/*.hip file for gfx950 with the library's flags and prints, per kernel:
registers, spills, scratch, and - inside the span from its first to its last MFMA - the things that cost round 2 its time before
anyone looked: `s_waitcnt vmcnt(0)` that is not behind a branch (a drained LDS-DMA ring every K-tile), v_readlane / v_writelane
(SGPR spills), scratch accesses, and vector loads that are not LDS-DMA.
Usage: python tools/isa_scan.py [gemm_big.hip] [--filter nt_big]"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-munsafe-fp-atomics', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-Wno-unused-function']


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('file', nargs='?', default='gemm_big.hip')
  ap.add_argument('--filter', default='')
  a = ap.parse_args()
  src = os.path.join(ROOT, 'plainlm_amd', 'csrc', a.file)
  with tempfile.TemporaryDirectory() as td:
    extra = ['-fno-slp-vectorize'] if a.file in ('attn_causal.hip', 'attn_doc.hip') else []  # as csrc/Makefile builds them
    r = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + extra + ['-c', src, '-o', os.path.join(td, 'x.o'), '-save-temps=obj'], cwd=td, capture_output=True, text=True)
    if r.returncode != 0:
      sys.exit(r.stderr[-2000:])
    asm = [f for f in os.listdir(td) if f.endswith('gfx950.s')][0]
    txt = open(os.path.join(td, asm)).read()
  meta = {}
  for blk in txt.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1))
    meta[name] = dict(vgpr=g('vgpr_count'), sgpr=g('sgpr_count'), vspill=g('vgpr_spill_count'), sspill=g('sgpr_spill_count'), scratch=g('private_segment_fixed_size'))
  lines = txt.split('\n')
  starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
  for k, (i, name) in enumerate(starts):
    if a.filter and a.filter not in name:
      continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = [l.split(';')[0].rstrip() for l in lines[i:end]]
    mf = [j for j, l in enumerate(body) if 'v_mfma' in l]
    demangled = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    print(demangled[:110])
    print('   ', meta.get(name, {}))
    if not mf:
      continue
    span = body[mf[0]:mf[-1] + 1]
    drains = 0
    for j, l in enumerate(span):
      if 'vmcnt(0)' in l and not any('cbranch' in x for x in span[max(0, j - 3):j]):
        drains += 1
    print('    MFMA span: %d lines, %d MFMAs | unconditional vmcnt(0): %d | v_readlane/v_writelane: %d | scratch_: %d | vector loads (no LDS-DMA): %d'
          % (len(span), len(mf), drains, sum('v_readlane' in l or 'v_writelane' in l for l in span), sum('scratch_' in l for l in span),
             sum(bool(re.search(r'\b(global|flat|buffer)_load_(dword|ushort|ubyte|short)', l)) and 'lds' not in l for l in span)))


if __name__ == '__main__':
  main()

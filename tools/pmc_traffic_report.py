"""Join the FETCH_SIZE / WRITE_SIZE counter CSVs of two `rocprofv3 --pmc` passes over tools/prof_traffic.py with the
shape list it printed.  Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters
are in KiB; on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads, so it is doubled;
WRITE_SIZE matched known byte counts in our own kernels (swiglu_fwd: 131072 KiB for a 128 MiB output)."""
import csv
import glob
import json
import sys


def rows(d):
  out = []
  for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      out.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
  out.sort()
  return [(n, v) for _, n, v in out if n.startswith('void gemm_') or n.startswith('gemm_')]


def main(fetch_dir, write_dir, order_log, out_json):
  order = None
  for l in open(order_log):
    if l.startswith('ORDER '):
      order = json.loads(l[6:])
  f, w = rows(fetch_dir), rows(write_dir)
  main_f = [(n, v) for n, v in f if 'reduce' not in n]
  main_w = [(n, v) for n, v in w if 'reduce' not in n]
  res = []
  for i, (name, m, n, k, alg) in enumerate(order):
    fv = [main_f[2 * i][1], main_f[2 * i + 1][1]]
    wv = [main_w[2 * i][1], main_w[2 * i + 1][1]]
    fetch = 2.0 * 1024.0 * fv[1]
    write = 1024.0 * wv[1]
    res.append({'gemm': name, 'M': m, 'N': n, 'K': k, 'kernel': main_f[2 * i + 1][0].split('(')[0], 'algorithmic_bytes': alg,
                'fetch_bytes': fetch, 'write_bytes': write, 'traffic_bytes': fetch + write, 'traffic_over_algorithmic': round((fetch + write) / alg, 2)})
  json.dump(res, open(out_json, 'w'), indent=1)
  for r in res:
    print(r['gemm'].ljust(12), 'alg %.0f MB  fetch %.0f MB  write %.0f MB  x%.2f  %s' % (r['algorithmic_bytes'] / 1e6, r['fetch_bytes'] / 1e6, r['write_bytes'] / 1e6, r['traffic_over_algorithmic'], r['kernel'][:60]))


if __name__ == '__main__':
  main(*sys.argv[1:5])

"""Join rocprofv3 --pmc counter CSVs of tools/prof_kernels.py passes with the launch list it printed.

  python tools/pmc_report.py <order.log> <out.json> <pass_dir> [<pass_dir> ...]

Every pass directory holds one *counter_collection.csv (any counters).  Launches are matched to dispatches IN ORDER: for
each ORDER entry the next two dispatches whose kernel name contains its `match` string; the SECOND (warm) one is kept.
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950
FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads, so it is doubled; both are L2-miss side
(Infinity-Cache hits included).  SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16, summed over the chip:
MFMA utilisation = busy / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) (the counter is summed over the XCDs: GRBM / 8 / duration gives the ~2 GHz clock); the expected busy count from the algorithmic flops is printed
beside it as a calibration of that reading (ratio > 1: recomputed work, e.g. the two-pass attention backward)."""
import csv
import glob
import json
import sys


def dispatches(d):
  rows = {}
  for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      k = int(r['Dispatch_Id'])
      rows.setdefault(k, {'name': r['Kernel_Name'], 'c': {}})
      rows[k]['c'][r['Counter_Name']] = rows[k]['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
  return [rows[k] for k in sorted(rows)]


def main(order_log, out_json, *dirs):
  order = None
  for l in open(order_log):
    if l.startswith('ORDER '):
      order = json.loads(l[6:])
  res = {'csrc_sha': order['csrc_sha'], 'rows': []}
  per_pass = []
  for d in dirs:
    disp = dispatches(d)
    pos, got = 0, []
    for e in order['launches']:
      found = []
      p = pos
      while p < len(disp) and len(found) < 2:
        if e['match'] in disp[p]['name'] and not (e['match'].startswith('gemm_') and 'reduce' in disp[p]['name']):
          found.append(p)
        p += 1
      if 'part_of' in e:  # a reduce kernel that follows each launch of its GEMM: the (warm) one right behind the kept dispatch
        prev = next((q for q in range(pos - 1, len(disp)) if e['match'] in disp[q]['name']), None)
        got.append(disp[prev] if prev is not None else None)
        continue
      if e.get('optional'):  # kernels of one multi-kernel op (attention backward): interleaved dispatches, keep the last (warm) one
        allm = [q for q in range(pos, len(disp)) if e['match'] in disp[q]['name']]
        got.append(disp[allm[-1]] if allm else None)
        continue
      if len(found) < 2:
        raise SystemExit(f'{d}: launch {e["name"]} not found after dispatch {pos}')
      pos = found[1] + 1
      got.append(disp[found[1]])
    per_pass.append(got)
  for i, e in enumerate(order['launches']):
    c, kname = {}, None
    for got in per_pass:
      if got[i] is not None:
        c.update(got[i]['c'])
        kname = got[i]['name'].split('(')[0]
    if kname is None:
      continue
    row = {'gemm': e['name'], 'kernel': kname, **{k: e[k] for k in ('M', 'N', 'K', 'B', 'T', 'nh', 'part_of') if k in e},
           'algorithmic_bytes': e['algorithmic_bytes'], 'flops': e['flops'], 'counters': c}
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
      row['fetch_bytes'] = 2.0 * 1024.0 * c['FETCH_SIZE']
      row['write_bytes'] = 1024.0 * c['WRITE_SIZE']
      row['traffic_bytes'] = row['fetch_bytes'] + row['write_bytes']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c.get('GRBM_GUI_ACTIVE'):
      row['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4)  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
      if e['flops']:
        row['mfma_busy_over_algorithmic'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (32.0 * e['flops'] / (2 * 32 * 32 * 16)), 3)
    res['rows'].append(row)
  # fold reduce kernels' traffic into their GEMM
  by = {r['gemm']: r for r in res['rows']}
  for r in res['rows']:
    if 'part_of' in r and r['part_of'] in by and 'traffic_bytes' in r:
      p = by[r['part_of']]
      for k in ('fetch_bytes', 'write_bytes', 'traffic_bytes'):
        p[k] += r[k]
  for r in res['rows']:
    if 'traffic_bytes' in r and r['algorithmic_bytes']:
      r['traffic_over_algorithmic'] = round(r['traffic_bytes'] / r['algorithmic_bytes'], 2)
  json.dump(res, open(out_json, 'w'), indent=1)
  for r in res['rows']:
    s = r['gemm'].ljust(30)
    if 'traffic_bytes' in r:
      s += ' alg %6.0f MB fetch %6.0f MB write %6.0f MB x%-5s' % (r['algorithmic_bytes'] / 1e6, r['fetch_bytes'] / 1e6, r['write_bytes'] / 1e6, r.get('traffic_over_algorithmic', '-'))
    if 'mfma_util' in r:
      s += ' mfma_util %.3f (busy/alg %s)' % (r['mfma_util'], r.get('mfma_busy_over_algorithmic', '-'))
    extra = {k: v for k, v in r['counters'].items() if k.startswith('SQ_') and k not in ('SQ_VALU_MFMA_BUSY_CYCLES',)}
    if extra:
      s += ' ' + ' '.join(f'{k[3:]}={v:.3g}' for k, v in sorted(extra.items()))
    print(s + '  ' + r['kernel'][:50])


if __name__ == '__main__':
  main(*sys.argv[1:])

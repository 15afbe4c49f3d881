#!/usr/bin/env python3
"""Summarises rocprofv3 output directories into profiles/:
  kernel stats  (--kernel-trace --stats)      -> per-kernel avg duration
  PMC passes    (--pmc FETCH_SIZE / WRITE_SIZE) -> HBM bytes per launch
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: counters
are in KiB-ish units of 1024 B per the guide's hbm_bytes recipe
((FETCH_SIZE + WRITE_SIZE) * 1024) and on gfx950 FETCH_SIZE reads exactly half
the bytes of a wide (16 B/lane) coalesced stream, so it is doubled."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_counters(d):
  per = defaultdict(lambda: defaultdict(list))
  for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'),
                        recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        per[row['Kernel_Name']][row['Counter_Name']].append(
            float(row['Counter_Value']))
  return per


def read_stats(d):
  out = {}
  for path in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'),
                        recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        out[row['Name']] = dict(calls=int(row['Calls']),
                                avg_us=float(row['AverageNs']) / 1e3,
                                min_us=float(row['MinNs']) / 1e3,
                                max_us=float(row['MaxNs']) / 1e3)
  return out


def main():
  trace_dir, fetch_dir, write_dir, out_json = sys.argv[1:5]
  stats = read_stats(trace_dir)
  fetch = read_counters(fetch_dir)
  write = read_counters(write_dir)
  valu = read_counters(sys.argv[5]) if len(sys.argv) > 5 else {}
  result = {}
  for name, st in stats.items():
    if not any(w in name for w in ('march', 'direct', 'copy', 'ldswin', 'lds2d')):
      continue
    entry = dict(st)
    f = fetch.get(name, {}).get('FETCH_SIZE')
    w = write.get(name, {}).get('WRITE_SIZE')
    if f and w:
      fb = sum(f) / len(f) * 1024 * 2     # gfx950: FETCH_SIZE counts half
      wb = sum(w) / len(w) * 1024
      entry.update(fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                   hbm_bytes_per_launch=fb + wb,
                   source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate '
                   'passes; FETCH_SIZE x2 (gfx950), x1024 B')
    v = valu.get(name, {})
    if v.get('SQ_INSTS_VALU'):
      entry['valu_wave_instructions_per_launch'] = (
          sum(v['SQ_INSTS_VALU']) / len(v['SQ_INSTS_VALU']))
    if v.get('SQ_WAVES'):
      entry['waves_per_launch'] = sum(v['SQ_WAVES']) / len(v['SQ_WAVES'])
    if v.get('GRBM_GUI_ACTIVE'):
      # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
      entry['gui_active_cycles_per_xcd'] = (
          sum(v['GRBM_GUI_ACTIVE']) / len(v['GRBM_GUI_ACTIVE']) / 8)
    result[name] = entry
  # tie every measurement to the KERNEL it was taken on: bench.py reports, per
  # kernel, a hash of its machine code + descriptor (isa_key, soda_amd/isa.py)
  # and drops a traffic figure whose key differs from what it built (the
  # module-source key of rounds 1-4 is kept beside it)
  if len(sys.argv) > 6:
    with open(sys.argv[6]) as f:
      line = [l for l in f.read().splitlines() if l.startswith('{')][-1]
    bench = json.loads(line)
    roof = bench['roofline']
    triples = [(k.get('kernel'), k.get('isa_key'), roof.get('kernel_key'))
               for k in roof.get('scheduled_kernels', [])]
    triples.append((roof.get('kernel'), roof.get('isa_key'),
                    roof.get('kernel_key')))
    si = bench.get('single_iter') or {}
    triples.append((si.get('kernel'), si.get('isa_key'), si.get('kernel_key')))
    for name, isa_key, key in triples:
      if name in result:
        if isa_key:
          result[name]['isa_key'] = isa_key
        if key:
          result[name]['kernel_key'] = key
        result[name]['compiler'] = roof.get('compiler')
  with open(out_json, 'w') as f:
    json.dump(result, f, indent=1, sort_keys=True)
  print(json.dumps(result, indent=1, sort_keys=True))


if __name__ == '__main__':
  main()

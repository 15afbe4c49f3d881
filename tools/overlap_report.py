#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace --memory-copy-trace directory of a slab
group run and reports how much of the halo copies' time ran UNDER stencil
kernels: for every copy (memory-copy records and the runtime's copyBuffer
kernels), the stencil kernels whose execution overlaps it in time.

  python tools/overlap_report.py <trace dir> [out.json]
"""
import csv
import glob
import json
import os
import sys


def rows(d, pattern):
  out = []
  for path in glob.glob(os.path.join(d, '**', pattern), recursive=True):
    with open(path) as f:
      out.extend(csv.DictReader(f))
  return out


def main():
  d = sys.argv[1]
  kernels, copies = [], []
  for r in rows(d, '*kernel_trace.csv'):
    name = r.get('Kernel_Name') or r.get('Name') or ''
    t0, t1 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    (copies if 'copyBuffer' in name or 'copy' in name.lower() and
     'march' not in name else kernels).append((t0, t1, name))
  for r in rows(d, '*memory_copy_trace.csv'):
    copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                   r.get('Direction', 'copy')))
  kernels = [k for k in kernels if 'march' in k[2] or 'direct' in k[2]]
  kernels.sort()
  total = under = 0
  examples = []
  for c0, c1, what in copies:
    if 'HOST' in what.upper():
      continue          # scatter / gather, not the exchange
    covered = 0
    names = set()
    for k0, k1, name in kernels:
      if k1 <= c0:
        continue
      if k0 >= c1:
        break
      covered += min(c1, k1) - max(c0, k0)
      names.add(name)
    covered = min(covered, c1 - c0)
    total += c1 - c0
    under += covered
    if len(examples) < 12:
      examples.append({'copy': what[:60], 'start_ns': c0, 'ns': c1 - c0,
                       'ns_under_kernels': covered,
                       'kernels': sorted(names)[:4]})
  out = {'copies': len([c for c in copies if 'HOST' not in c[2].upper()]),
         'stencil_kernels': len(kernels),
         'copy_ns_total': total, 'copy_ns_under_stencil_kernels': under,
         'fraction_under_kernels': under / total if total else None,
         'examples': examples}
  text = json.dumps(out, indent=1)
  print(text)
  if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
      f.write(text + '\n')


if __name__ == '__main__':
  main()

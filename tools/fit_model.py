#!/usr/bin/env python3
"""Fits the constants of the library's launch-time model (soda_hip.cpp
kernel_geometry: issue shares per resident wave count, the rule that combines
issue time and memory time, the fixed launch cost) to measured pass times
(tools/model_check.py output, profiles/r03_model_data.jsonl).  Offline, no GPU:
prints the fitted constants and the residuals; the constants are then pasted
into soda_hip.cpp and checked on the GPU by tests/test_hip_parity.py
test_model_schedule_is_close_to_the_calibrated_one."""
import json
import math
import sys

import numpy as np
from scipy.optimize import least_squares

SIMDS = 1024
NS_PER_OP_IN_DATA = 1.4   # runtime.NS_PER_VALU_OP when the data was taken


def waves_per_simd(v):
  a = max(8, -(-v // 8) * 8)
  return max(1, min(8, 512 // a))


def points(path):
  out = []
  for line in open(path):
    r = json.loads(line)
    ext = r['extent']
    for t, p in r['passes'].items():
      d = p['desc']
      tile = p['tile']
      ax = len(ext) - 1
      others = 1
      for i in range(len(ext)):
        if i != ax:
          others *= -(-ext[i] // tile[i])
      wpb = (d['block'][0] * d['block'][1] * d['block'][2] + 63) // 64
      along = max(1, d['waves_along'])
      per_wave = tile[ax] // along
      chunks = -(-ext[ax] // per_wave)
      waves = others * (-(-chunks // along)) * wpb
      cells = 1
      for e in ext:
        cells *= e
      out.append(dict(ext=ext, T=int(t), waves=waves, cap=waves_per_simd(p['vgpr']),
                      steps=max(1.0, per_wave + d['warm'] - d['warm_saved']),
                      step_ops=d['step_ns'] / d.get('ns_per_op', NS_PER_OP_IN_DATA),
                      mix=d.get('shift') == 'mixh', per_wave=per_wave,
                      warm=d['warm'], lanes=max(1.0, d['lanes']), cells=cells,
                      bpc=d['bytes_per_cell'], pipe=max(1, d['pipe']),
                      measured=p['measured_us'] * 1e3, old=p['model_us'] * 1e3,
                      dim=len(ext)))
  return out


def model(pt, th):
  ns_op, s1, s2, s3, s4, rate, p, launch, wave_ns, mix = th
  share_of = {1: s1, 2: s2, 3: s3}
  def sh(k):
    return share_of.get(k, s4)
  slots = pt['cap'] * SIMDS
  full, rem = divmod(pt['waves'], slots)
  share = full * pt['cap'] * sh(pt['cap'])
  if rem > 0 or full == 0:
    rk = max(1, -(-rem // SIMDS))
    share += rk * sh(rk)
  valu = share * pt['steps'] * pt['step_ops'] * ns_op / pt['pipe']
  if pt.get('mix'):
    valu *= mix     # runtime.MIXH_FACTOR
  rows_factor = (pt['per_wave'] + pt['warm']) / pt['per_wave']
  bytes_ = pt['cells'] * pt['bpc'] * (0.5 * pt['lanes'] * rows_factor + 0.5)
  mem = bytes_ / rate
  both = (valu ** p + mem ** p) ** (1.0 / p)
  return both + launch + wave_ns * pt['waves'] / SIMDS


def main():
  pts = []
  for path in (sys.argv[1].split(',') if len(sys.argv) > 1 else
               ['profiles/r03_model_data.jsonl']):
    pts.extend(points(path))
  pts2 = pts     # 2-D and 3-D together
  x0 = np.array([1.4, 2.0, 1.2, 1.0, 1.0, 6300.0, 3.0, 2000.0, 0.0, 0.8])
  lo = np.array([0.5, 1.0, 0.8, 0.8, 0.8, 4000.0, 1.0, 0.0, 0.0, 0.4])
  hi = np.array([3.0, 4.0, 2.5, 2.5, 2.5, 8000.0, 8.0, 10000.0, 3000.0, 1.2])

  def resid(th):
    return [math.log(model(q, th) / q['measured']) for q in pts2]

  fit = least_squares(resid, x0, bounds=(lo, hi))
  th = fit.x
  names = ['ns_per_op', 'share1', 'share2', 'share3', 'share4+', 'bytes_per_ns',
           'p', 'launch_ns', 'ns_per_wave_per_simd', 'mixh_factor']
  print({n: round(float(v), 3) for n, v in zip(names, th)})
  r = np.array(resid(th))
  old = np.array([math.log(q['old'] / q['measured']) for q in pts2])
  print('points %d: rms log error new %.3f (max %.3f), old %.3f (max %.3f)' %
        (len(pts2), math.sqrt((r ** 2).mean()), abs(r).max(),
         math.sqrt((old ** 2).mean()), abs(old).max()))
  for q, e in zip(pts2, r):
    print(q['ext'], 'T', q['T'], 'measured %.1f model %.1f (%+.0f %%) old %.1f' %
          (q['measured'] / 1e3, model(q, th) / 1e3, 100 * (math.exp(e) - 1),
           q['old'] / 1e3))
  for q in pts:
    if q['dim'] == 3:
      print(q['ext'], 'T', q['T'], 'measured %.1f model %.1f old %.1f' %
            (q['measured'] / 1e3, model(q, th) / 1e3, q['old'] / 1e3))


if __name__ == '__main__':
  main()


def knap(cost, iterate):
  best = [0.0] + [1e18] * iterate
  pick = [None] * (iterate + 1)
  for n in range(1, iterate + 1):
    for t, c in cost.items():
      if t <= n and best[n - t] + c < best[n]:
        best[n], pick[n] = best[n - t] + c, t
  out, n = {}, iterate
  while n:
    out[pick[n]] = out.get(pick[n], 0) + 1
    n -= pick[n]
  return out


def schedules(path, th, passes=(12, 8, 4, 1), iterate=100):
  by = {}
  for q in points(path):
    by.setdefault(tuple(q['ext']), {})[q['T']] = q
  for ext, d in by.items():
    if not all(t in d for t in passes):
      continue
    meas = {t: d[t]['measured'] for t in passes}
    for label, cost in (('new', {t: model(d[t], th) for t in passes}),
                        ('old', {t: d[t]['old'] for t in passes})):
      s = knap(cost, iterate)
      ideal = knap(meas, iterate)
      t_s = sum(meas[t] * c for t, c in s.items())
      t_i = sum(meas[t] * c for t, c in ideal.items())
      print(ext, label, s, 'costs %.1f us, calibrated %s %.1f us: %+.1f %%' %
            (t_s / 1e3, ideal, t_i / 1e3, 100 * (t_s / t_i - 1)))


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'sched':
  pts_ = []
  for path in sys.argv[1].split(','):
    pts_.extend(points(path))
  x0 = np.array([1.4, 2.0, 1.2, 1.0, 1.0, 6300.0, 3.0, 2000.0, 0.0, 0.8])
  lo = np.array([0.5, 1.0, 0.8, 0.8, 0.8, 4000.0, 1.0, 0.0, 0.0, 0.4])
  hi = np.array([3.0, 4.0, 2.5, 2.5, 2.5, 8000.0, 8.0, 10000.0, 3000.0, 1.2])
  fit = least_squares(lambda th: [math.log(model(q, th) / q['measured'])
                                  for q in pts_], x0, bounds=(lo, hi))
  schedules(sys.argv[1].split(',')[-1], fit.x)

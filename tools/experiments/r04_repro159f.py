#!/usr/bin/env python3
"""Delta-reduces tools/fuzz_scan.py deep seed 159."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import c_oracle
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
HEAD = 'kernel: d159\nburst width: 64\nunroll factor: 2\niterate: 1\ninput uint8: in0(32, *)\ninput uint8: in1\n'
T = {'T1': 'in1(2, 0) * 43 * (min(16, 32) * in0(1, 0))', 'T2': '(in1(-1, -1) + 10) * in1(0, -2)',
     'T3': '(in0(2, -2) - in0(2, 1)) * 40', 'T4': 'in0(1, 2)'}
U = {'U1': 'min(int32(in0(-2, 1)), 4)', 'U2': 'in1(-2, -2) * in1(2, -1)', 'U3': '(in0(2, -2) + 8 * 16)',
     'U4': 'in0(1, -2) * in1(0, 1)'}
OUT0 = 'output uint8: out0(0, 0) = in1(2, 1) * 1\n'
extent = (520, 291)
rng = np.random.default_rng(4401)
ins0 = {n: rng.integers(0, 201, extent[::-1]).astype(np.uint8) for n in ('in0', 'in1')}

def run(text):
  st = core.from_text(text)
  want = c_oracle.COracle(st).run(ins0)
  with runtime.Program(st, lower.LowerOptions(peel=0), extent=extent) as prog:
    got = prog.run(ins0)
  bad = 0
  for o in st.output_names:
    lo, hi = st.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    bad += int((got[o][idx] != want[o][idx]).sum())
  return bad

SIGN = {'T1': '+', 'T2': '-', 'T3': '+', 'T4': '+', 'U1': '+', 'U2': '+', 'U3': '-', 'U4': '-'}

def chain(terms, table):
  out = ''
  for i, t in enumerate(terms):
    sg = SIGN[t]
    out += (('0 - ' if sg == '-' else '') if i == 0 else ' %s ' % sg) + table[t]
  return out

def prog(ts, us, out0=OUT0, div=' / 6', loc_t='uint16'):
  loc = 'local %s: loc0(0, 0) = %s\n' % (loc_t, chain(ts, T)) if ts else ''
  return HEAD + loc + out0 + 'output uint8: out1(0, 0) = (%s)%s\n' % (chain(us, U), div)

print('full', run(prog(list(T), list(U))))
for n in (1, 2, 3):
  for ts in itertools.combinations(T, n):
    print('loc0 =', ts, 'out1 full:', run(prog(ts, list(U))), flush=True)
for n in (1, 2, 3):
  for us in itertools.combinations(U, n):
    print('loc0 full, out1 =', us, ':', run(prog(list(T), us)), flush=True)
print('no division', run(prog(list(T), list(U), div='')))
print('out0 = in1(0,0)', run(prog(list(T), list(U), out0='output uint8: out0(0, 0) = in1(0, 0)\n')))
print('loc int16', run(prog(list(T), list(U), loc_t='int16')))

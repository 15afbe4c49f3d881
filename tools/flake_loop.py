#!/usr/bin/env python3
"""Round-3 flake, reproduced and closed: loops the overlapped rank-per-GPU
exchange (tests/overlap_case.py: dist.StreamOverlap + soda_hip_run_device_slab,
ranks as threads on the one GPU, messages by tests/fabric.py) and counts the
trials whose stitched result differs from the C oracle.

  python tools/flake_loop.py --trials 200 --out profiles/r04_flake.json

One process per (fabric lifetime, GPU_MAX_HW_QUEUES) setting -- both are read
when HIP / tests/fabric.py load -- each looping every parametrisation of
tests/test_dist.py::test_exchange_hidden_under_the_compute under four skews:
none; one rank spinning on the GPU in front of every interval; one rank
sleeping on the host; both.  `unsafe` brings back round 3's fabric (staging
block handed back to the allocator before the receiver's copy has run): with it
mismatches are expected, without it any mismatch is the PRODUCT's."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

UPWIND = """kernel: upwind
burst width: 64
unroll factor: 2
iterate: 4
input float: u(64, *)
output float: v(0, 0) = (u(0, 0) + u(0, 1) + u(-1, 0) + u(1, 1)) * 0.25f
"""

CASES = [
    # (label, program, extent, iterate, every, fuse, world)
    ('jacobi2d-1024x1600-40-12-world4', 'jacobi2d.soda', (1024, 1600), 40, 12,
     (12, 4), 4),
    ('jacobi2d-512x480-23-4-world8', 'jacobi2d.soda', (512, 480), 23, 4, (4,),
     8),
    ('heat3d-64x48x96-9-2-world4', 'heat3d.soda', (64, 48, 96), 9, 2, (2,), 4),
    ('upwind-512x227-10-1-world6', UPWIND, (512, 227), 10, 1, (), 6),
]


def skews(world):
  mid = world // 2
  return [
      ('none', {}, {}),
      ('gpu_spin', {1: 2_000_000}, {}),
      ('host_sleep', {}, {mid: 0.001}),
      ('both', {world - 2: 1_000_000, 0: 300_000}, {1: 0.0005}),
  ]


def worker(args):
  import numpy as np
  import overlap_case
  import fabric
  from conftest import soda_path
  from soda_amd import core
  from oracle import c_oracle
  rows = []
  for label, prog, extent, iterate, every, fuse, world in CASES:
    if args.only and args.only not in label:
      continue

    def load(iters):
      if prog.endswith('.soda'):
        return core.from_file(soda_path(prog), iterate=iters)
      return core.from_text(prog, iterate=iters)

    stencil, again = load(iterate), load(2 * iterate)
    rng = np.random.default_rng(5)
    fields = {n: rng.random(tuple(extent[::-1]), dtype=np.float32)
              for n in stencil.input_names}
    want = c_oracle.COracle(again).run(fields)
    with overlap_case.Case(stencil, extent, every, fuse, world,
                           calibrate=args.calibrate) as case:
      for skew, spin, sleep in skews(world):
        bad_trials = bad_cells = 0
        t0 = time.time()
        for _ in range(args.trials):
          got, _ = case.trial(fields, iterate, runs=2, spin=spin, sleep=sleep)
          bad = overlap_case.mismatches(again, extent, got, want, 2 * iterate)
          bad_trials += bad > 0
          bad_cells += bad
        row = {'case': label, 'skew': skew, 'trials': args.trials,
               'mismatching_trials': int(bad_trials),
               'mismatching_cells': int(bad_cells),
               'fabric': 'unsafe (round 3)' if fabric.UNSAFE_LIFETIME else
                         'payload lives until the copy ran',
               'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES',
                                                   'default'),
               'calibrate': bool(args.calibrate),
               'seconds': round(time.time() - t0, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
  return rows


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--trials', type=int, default=200)
  ap.add_argument('--out', default=None)
  ap.add_argument('--only', default=None, help='substring of a case label')
  ap.add_argument('--calibrate', action='store_true')
  ap.add_argument('--queues', nargs='*', default=['default', '2', '8'])
  ap.add_argument('--unsafe-trials', type=int, default=None,
                  help='trials of the round-3 fabric (default: --trials)')
  ap.add_argument('--worker', action='store_true')
  args = ap.parse_args()
  if args.worker:
    worker(args)
    return
  import __graft_entry__ as entry
  entry.build_library()
  rows = []
  settings = [(False, q) for q in args.queues] + [(True, 'default')]
  for unsafe, queues in settings:
    env = dict(os.environ)
    env.pop('SODA_FABRIC_UNSAFE_LIFETIME', None)
    env.pop('GPU_MAX_HW_QUEUES', None)
    if unsafe:
      env['SODA_FABRIC_UNSAFE_LIFETIME'] = '1'
    if queues != 'default':
      env['GPU_MAX_HW_QUEUES'] = queues
    trials = args.unsafe_trials if unsafe and args.unsafe_trials else args.trials
    cmd = [sys.executable, os.path.abspath(__file__), '--worker', '--trials',
           str(trials)] + (['--only', args.only] if args.only else []) + (
               ['--calibrate'] if args.calibrate else [])
    run = subprocess.run(cmd, env=env, capture_output=True, text=True)
    sys.stderr.write(run.stderr[-2000:])
    for line in run.stdout.splitlines():
      if line.startswith('{'):
        rows.append(json.loads(line))
        print(line, flush=True)
    if run.returncode:
      rows.append({'error': run.stderr[-500:], 'unsafe': unsafe,
                   'queues': queues})
  summary = {
      'what': __doc__.split('\n\n')[0],
      'product_mismatching_trials': sum(
          r.get('mismatching_trials', 0) for r in rows
          if not str(r.get('fabric', '')).startswith('unsafe')),
      'round3_fabric_mismatching_trials': sum(
          r.get('mismatching_trials', 0) for r in rows
          if str(r.get('fabric', '')).startswith('unsafe')),
      'rows': rows,
  }
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(summary, f, indent=1)
  print(json.dumps({k: v for k, v in summary.items() if k != 'rows'}))


if __name__ == '__main__':
  main()

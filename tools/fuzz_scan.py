#!/usr/bin/env python3
"""One-off scan of random programs beyond the seeds the test suite holds: the
generic generator (tests/fuzz.py program) and the window generator
(window_program), GPU kernels (auto and direct) against the C oracle, bit for
bit; `group`: the same programs cut into virtual slabs (group_scan).
`options`: random backend knobs on larger grids (options_scan).
`wire`: behind the reference host's stream format (wire_scan).
`deep`: 8-26 iterations at fusion depths up to 13 (deep_scan).
`ranks`: thread-ranks with the overlapped exchange under skew (ranks_scan).
Usage: python tools/fuzz_scan.py window|generic|rich|group|options|wire|deep|ranks|wide FIRST LAST"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
  import fuzz
  from oracle import c_oracle
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
  gen = (fuzz.window_program if kind == 'window' else
         (lambda s: fuzz.program(s, rich=True)) if kind == 'rich' else
         fuzz.program)
  ext_for = fuzz.window_extent_for if kind == 'window' else fuzz.extent_for
  ran = failed = 0
  t0 = time.time()
  for seed in range(first, last):
    text, dim, _ = gen(seed)
    try:
      stencil = core.from_text(text)
    except util.SodaError:
      continue
    extent = ext_for(seed, dim)
    lo, hi = stencil.valid_box(extent)
    if not all(h > l for l, h in zip(lo, hi)):
      continue
    ins = fuzz.inputs_for(stencil, extent, seed)
    want = c_oracle.COracle(stencil, openmp=False).run(ins)
    ran += 1
    for strategy in ('auto', 'direct'):
      try:
        with runtime.Program(stencil, lower.LowerOptions(strategy=strategy,
                                                         fuse=(2,)),
                             extent=extent) as prog:
          got = prog.run(ins)
      except Exception as e:  # noqa
        failed += 1
        print('seed %d %s: %s: %s\n%s' % (seed, strategy, type(e).__name__,
                                          str(e)[:300], text), flush=True)
        continue
      for o in stencil.output_names:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        if not np.array_equal(got[o][idx], want[o][idx], equal_nan=True):
          failed += 1
          print('seed %d %s output %s: %d cells differ\n%s' %
                (seed, strategy, o, int((got[o][idx] != want[o][idx]).sum()),
                 text), flush=True)
    if ran % 25 == 0:
      print('... %d programs, %d failures, %.0f s' % (ran, failed,
                                                     time.time() - t0),
            flush=True)
  print('%s seeds [%d, %d): %d programs run, %d failures' %
        (kind, first, last, ran, failed))
  return 1 if failed else 0


def group_scan(first, last):
  """Random iterable programs on 2-6 virtual slabs of the one GPU (random
  exchange interval, fusion depth, enqueueing threads, `border: preserve`)
  through soda_hip_group_* against the C oracle, bit for bit."""
  import fuzz
  from oracle import c_oracle
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  ran = failed = skipped = 0
  t0 = time.time()
  for seed in range(first, last):
    rng = np.random.default_rng(seed + 31000)
    kind = ['plain', 'window', 'rich'][int(rng.integers(3))]
    text, dim, _ = (fuzz.window_program(seed) if kind == 'window' else
                    fuzz.program(seed, rich=kind == 'rich'))
    if dim == 1:
      continue
    iterate = int(rng.integers(1, 8))
    if 'int32(' in text and kind == 'rich':
      iterate = 1      # (float -> int32 of values that grow: undefined in C)
    border = 'preserve' if rng.random() < 0.3 else None
    try:
      stencil = core.from_text(text, iterate=iterate,
                               **({'border': border} if border else {}))
      if border:
        stencil.check_preserve()
    except util.SodaError:
      continue
    extent = ((int(rng.choice([64, 130, 258, 300])), int(rng.integers(150, 420)))
              if dim == 2 else
              (int(rng.choice([40, 64, 130])), int(rng.integers(10, 24)),
               int(rng.integers(60, 130))))
    lo, hi = stencil.valid_box(extent)
    if not border and not all(h > l for l, h in zip(lo, hi)):
      continue
    slabs = int(rng.integers(2, 7))
    every = int(rng.integers(0, iterate + 1))
    fuse = [(), (2,), (3, 2), (4,)][int(rng.integers(4))]
    threads = bool(rng.random() < 0.3)
    overlap = bool(rng.random() < 0.8)
    ins = fuzz.inputs_for(stencil, extent, seed)
    want = c_oracle.COracle(stencil, openmp=False).run(ins)
    what = ('seed %d %s dim %d iterate %d border %s extent %s slabs %d every %d '
            'fuse %s threads %s overlap %s' %
            (seed, kind, dim, iterate, border, extent, slabs, every, fuse,
             threads, overlap))
    try:
      with runtime.Group(stencil, extent, [0] * slabs,
                         lower.LowerOptions(fuse=fuse), exchange_every=every,
                         overlap=overlap, threads=threads) as group:
        got = group.run_host(ins)
    except util.SodaError as e:
      if 'thinner' in str(e) or 'ghost' in str(e):
        skipped += 1
        continue
      failed += 1
      print('%s: %s: %s\n%s' % (what, type(e).__name__, str(e)[:300], text),
            flush=True)
      continue
    ran += 1
    for o in stencil.output_names:
      if border:
        g, w = got[o], want[o]
      else:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        g, w = got[o][idx], want[o][idx]
      if not np.array_equal(g, w, equal_nan=True):
        failed += 1
        print('%s output %s: %d cells differ\n%s' %
              (what, o, int((g != w).sum()), text), flush=True)
    if ran % 25 == 0:
      print('... %d groups, %d failures, %d skipped, %.0f s' %
            (ran, failed, skipped, time.time() - t0), flush=True)
  print('group seeds [%d, %d): %d groups run, %d skipped (slabs thinner than '
        'their ghosts), %d failures' % (first, last, ran, skipped, failed))
  return 1 if failed else 0


def options_scan(first, last, only=None):
  """Random programs on grids of several strips and chunks with random backend
  knobs: chunk length (the launch-time tuner may pick any), peeled warm-up,
  fusion depth, prefetch depth, lane-shift flavour, cells per lane, blocks that
  share rows, pipelined waves, the rewrites on / off."""
  import fuzz
  from oracle import c_oracle
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  ran = failed = refused = 0
  t0 = time.time()
  for seed in range(first, last):
    rng = np.random.default_rng(seed + 47000)
    kind = ['plain', 'window', 'window', 'rich'][int(rng.integers(4))]
    text, dim, iterate = (fuzz.window_program(seed) if kind == 'window' else
                          fuzz.program(seed, rich=kind == 'rich'))
    if only and only not in text:
      continue
    border = 'preserve' if rng.random() < 0.25 else None
    try:
      stencil = core.from_text(text, **({'border': border} if border else {}))
      if border:
        stencil.check_preserve()
    except util.SodaError:
      continue
    extent = ((int(rng.integers(900, 5000)),) if dim == 1 else
              (int(rng.choice([300, 520, 777, 1100])), int(rng.integers(60, 260)))
              if dim == 2 else
              (int(rng.choice([64, 130, 300])), int(rng.integers(12, 40)),
               int(rng.integers(20, 70))))
    lo, hi = stencil.valid_box(extent)
    if not border and not all(h > l for l, h in zip(lo, hi)):
      continue
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    kw = dict(strategy=pick(['auto', 'auto', 'auto', 'direct']) if only else 'auto',
              fuse=pick([(), (2,), (3,), (3, 2)]),
              chunk_rows=pick([None, None, 3, 5, 9, 16, 33]),
              peel=pick([None, None, 0, 1, -1]),
              prefetch=pick([None, None, 1, 2, 4]),
              lane_shift=pick([None, None, None, 'dpp', 'mixh', 'swzh',
                               'bperm']) if dim == 2 else None,
              windows=pick([None, None, False]),
              inline=pick([None, None, False]),
              xshare=pick([None, None, None, True]),
              nt_store=pick([None, None, True, False]),
              tile_rows=pick([None, None, 2, 4, 6]) if dim == 3 else None)
    if rng.random() < 0.15 and dim == 2 and kw['fuse'] in ((2,), ):
      kw['pipe'] = 2
    if rng.random() < 0.15:
      kw['waves_y'] = 2
    ins = fuzz.inputs_for(stencil, extent, seed)
    want = c_oracle.COracle(stencil, openmp=False).run(ins)
    what = 'seed %d %s%s extent %s %s' % (seed, kind, ' preserve' if border else '',
                                          extent, kw)
    try:
      with runtime.Program(stencil, lower.LowerOptions(**kw),
                           extent=extent) as prog:
        got = prog.run(ins)
    except (util.SodaError, ValueError) as e:
      refused += 1        # a knob the program / shape does not admit
      if not any(w in str(e) for w in ('cannot', 'need', 'fit', 'must',
                                       'support', 'not ', 'only')):
        print('%s: %s: %s' % (what, type(e).__name__, str(e)[:300]), flush=True)
      continue
    ran += 1
    for o in stencil.output_names:
      if border:                       # the whole grid is defined
        g, w = got[o], want[o]
      else:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        g, w = got[o][idx], want[o][idx]
      if not np.array_equal(g, w, equal_nan=True):
        failed += 1
        print('%s output %s: %d cells differ\n%s' %
              (what, o, int((g != w).sum()), text), flush=True)
    if ran % 25 == 0:
      print('... %d programs, %d failures, %d refused, %.0f s' %
            (ran, failed, refused, time.time() - t0), flush=True)
  print('options seeds [%d, %d): %d programs run, %d refused, %d failures' %
        (first, last, ran, refused, failed))
  return 1 if failed else 0


def wire_scan(first, last):
  """Random programs behind the reference host's wire format:
  scatter as the reference host does, run <app>_kernel's dense / linear forms
  on the banked streams, gather, compare with the stream-level restatement
  (oracle/frt_layout.kernel_on_streams) and -- single-tile layouts -- the n-D
  oracle on the common valid box."""
  import fuzz
  from oracle import frt_layout, numpy_oracle
  from soda_amd import core, stream, util
  ran = failed = refused = 0
  t0 = time.time()
  for seed in range(first, last):
    rng = np.random.default_rng(seed + 91000)
    kind = ['plain', 'plain', 'window', 'rich'][int(rng.integers(4))]
    text, dim, _ = (fuzz.window_program(seed) if kind == 'window' else
                    fuzz.program(seed, rich=kind == 'rich'))
    if dim == 1:
      continue
    try:
      stencil = core.from_text(text)
    except util.SodaError:
      continue
    tiles = rng.random() < 0.3
    if dim == 2:
      extent = (int(rng.integers(70, 130)) if tiles else int(rng.integers(20, 33)),
                int(rng.integers(12, 40)))
    else:
      extent = (int(rng.integers(40, 70)) if tiles else int(rng.integers(20, 33)),
                int(rng.integers(12, 33)), int(rng.integers(8, 16)))
    boxes = [stencil.valid_box(extent, o) for o in stencil.output_names]
    lo = [max(b[0][d] for b in boxes) for d in range(dim)]
    hi = [min(b[1][d] for b in boxes) for d in range(dim)]
    if not all(h > l for l, h in zip(lo, hi)):
      continue
    mode = 'dense' if rng.random() < 0.6 else 'linear'
    what = 'seed %d %s extent %s %s' % (seed, kind, extent, mode)
    ins = fuzz.inputs_for(stencil, extent, seed)
    try:
      layout = stream.WireLayout(stencil, extent)
      in_banks = frt_layout.scatter(layout, ins)
      out_banks = frt_layout.alloc(layout, stencil.output_names)
      prog = stream.StreamProgram(stencil, dense=mode != 'linear')
    except (util.SodaError, ValueError, NotImplementedError) as e:
      refused += 1
      continue
    try:
      prog.run_banked_host(out_banks, in_banks, layout.cycle_count)
      used = prog.last_mode
    except util.SodaError as e:
      failed += 1
      print('%s: %s: %s\n%s' % (what, type(e).__name__, str(e)[:300], text),
            flush=True)
      continue
    finally:
      prog.close()
    ran += 1
    got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
           for o, t in zip(stencil.output_names, stencil.output_types)}
    try:
      frt_layout.gather(layout, out_banks, got)
      ref = {o: np.zeros_like(got[o]) for o in got}
      frt_layout.gather(layout, frt_layout.kernel_on_streams(layout, in_banks),
                        ref)
    except Exception as e:   # noqa -- the restatement of the host, not the product
      refused += 1
      print('%s: oracle/frt_layout.py: %s: %s' % (what, type(e).__name__,
                                                  str(e)[:200]), flush=True)
      continue
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    # several outputs on several tiles: the host gathers every output over the
    # region of the PROGRAM's window, so an output with a wider window of its
    # own is read at tile-edge cells it cannot be computed at -- the 1-D form
    # wraps into the next row there, the dense form reads outside the array
    # (seed 333).  No kernel can be held to those: where the two restatements
    # of the contract differ, the cell is not compared.
    held = {o: np.ones(got[o][idx].shape, bool) for o in got}
    if len(stencil.output_names) > 1 and layout.tiles > 1:
      other = frt_layout.kernel_on_dense_view(layout, in_banks)
      if other is not None:
        ref2 = {o: np.zeros_like(got[o]) for o in got}
        frt_layout.gather(layout, other, ref2)
        held = {o: ref[o][idx] == ref2[o][idx] for o in got}
    bad = [o for o in stencil.output_names
           if not np.array_equal(got[o][idx][held[o]], ref[o][idx][held[o]],
                                 equal_nan=True)]
    # (several outputs: the host gathers each with ITS stencil offset over the
    # region of the program's window, and where the outputs' windows differ
    # the streams' cells and the n-D cells part ways -- the reference's layout,
    # not compared here)
    # (a delayed input on an array narrower than its tile reaches the kernel
    # misplaced -- the host delays in the caller's coordinates, the offset
    # counts tile rows, INTEGRATION.md 2b: the contract above is all there is)
    delayed = len(stencil.input_names) > 1 and \
        any(stencil.produce_offsets().values()) and \
        tuple(extent[:-1]) != tuple(stencil.tile_size[:-1])
    if not bad and layout.tiles == 1 and len(stencil.output_names) == 1 and \
        not delayed:
      want = numpy_oracle.run(stencil, ins)
      bad = [o + ' (n-D oracle)' for o in stencil.output_names
             if not np.array_equal(got[o][idx], want[o][idx], equal_nan=True)]
    if bad:
      failed += 1
      print('%s (ran %s, %d tiles) outputs %s differ\n%s' %
            (what, used, layout.tiles, bad, text), flush=True)
    if ran % 25 == 0:
      print('... %d programs, %d failures, %d refused, %.0f s' %
            (ran, failed, refused, time.time() - t0), flush=True)
  print('wire seeds [%d, %d): %d programs run, %d refused, %d failures' %
        (first, last, ran, refused, failed))
  return 1 if failed else 0


def ranks_scan(first, last):
  """The rank-per-GPU path under random programs (round 4): iterable 2-D / 3-D
  programs cut into 2-6 thread-ranks on the one GPU (tests/overlap_case.py:
  dist.StreamOverlap + soda_hip_run_device_slab, messages by tests/fabric.py),
  random exchange interval and fusion depth, two chained runs, one rank
  spinning on the GPU and one sleeping on the host; against the C oracle."""
  import fuzz
  import overlap_case
  from oracle import c_oracle
  from soda_amd import core, util
  ran = failed = skipped = 0
  t0 = time.time()
  for seed in range(first, last):
    rng = np.random.default_rng(seed + 67000)
    kind = ['plain', 'window', 'plain'][int(rng.integers(3))]
    text, dim, _ = (fuzz.window_program(seed) if kind == 'window' else
                    fuzz.program(seed))
    if dim == 1:
      continue
    iterate = int(rng.integers(2, 9))
    border = 'preserve' if rng.random() < 0.4 else None
    try:
      stencil = core.from_text(text, iterate=iterate,
                               **({'border': border} if border else {}))
      again = core.from_text(text, iterate=2 * iterate,
                             **({'border': border} if border else {}))
      if border:
        stencil.check_preserve()
    except util.SodaError:
      continue
    extent = ((int(rng.choice([64, 130, 258, 300])), int(rng.integers(150, 420)))
              if dim == 2 else
              (int(rng.choice([40, 64, 130])), int(rng.integers(10, 24)),
               int(rng.integers(60, 130))))
    lo, hi = again.valid_box(extent)
    if not border and not all(h > l for l, h in zip(lo, hi)):
      continue
    world = int(rng.integers(2, 7))
    every = int(rng.integers(1, iterate + 1))
    fuse = [(), (2,), (3, 2), (4,)][int(rng.integers(4))]
    spin = {int(rng.integers(world)): int(rng.integers(100_000, 1_500_000))} \
        if rng.random() < 0.6 else {}
    sleep = {int(rng.integers(world)): float(rng.uniform(0.0002, 0.002))} \
        if rng.random() < 0.4 else {}
    ins = fuzz.inputs_for(stencil, extent, seed)
    want = c_oracle.COracle(again, openmp=False).run(ins)
    what = ('seed %d %s dim %d iterate 2 x %d border %s extent %s world %d '
            'every %d fuse %s spin %s sleep %s' %
            (seed, kind, dim, iterate, border, extent, world, every, fuse,
             spin, sleep))
    try:
      with overlap_case.Case(stencil, extent, every, fuse, world) as case:
        got, _ = case.trial(ins, iterate, runs=2, spin=spin, sleep=sleep)
    except util.SodaError as e:
      if 'thinner' in str(e) or 'ghost' in str(e):
        skipped += 1
        continue
      failed += 1
      print('%s: %s: %s\n%s' % (what, type(e).__name__, str(e)[:300], text),
            flush=True)
      continue
    ran += 1
    bad = overlap_case.mismatches(again, extent, got, want, 2 * iterate,
                                  whole_grid=bool(border), by_value=True)
    if bad:
      failed += 1
      print('%s: %d cells differ\n%s' % (what, bad, text), flush=True)
    if ran % 25 == 0:
      print('... %d rank groups, %d failures, %d skipped, %.0f s' %
            (ran, failed, skipped, time.time() - t0), flush=True)
  print('ranks seeds [%d, %d): %d rank groups run, %d skipped (slabs thinner '
        'than their ghosts), %d failures' % (first, last, ran, skipped, failed))
  return 1 if failed else 0


def wide_scan(first, last):
  """Random wide-window programs (tests/fuzz_nest.py `wide`) through
  --hip-strategy ldswin -- 4 and 8 rows per step, random chunk lengths --
  against their own C++ nests (nothing shared with the product)."""
  import fuzz_nest
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  ran = failed = refused = 0
  t0 = time.time()
  for seed in range(first, last):
    rng = np.random.default_rng(seed + 71000)
    prog, extent = fuzz_nest.program(seed, 'wide')
    if fuzz_nest.has_empty_box(prog, extent):
      continue
    extent = (int(rng.choice([36, 260, 520, 1028, 1540, 2052])),
              int(rng.integers(40, 260)))
    if fuzz_nest.has_empty_box(prog, extent):
      continue
    stencil = core.from_text(prog.soda_text())
    ins = fuzz_nest.inputs_for(prog, extent, seed)
    want = prog.run(ins, extent)
    kw = dict(strategy='ldswin', waves_y=int(rng.choice([1, 4, 8])),
              chunk_rows=int(rng.choice([8, 24, 40, 64, 100])))
    try:
      with runtime.Program(stencil, lower.LowerOptions(**kw),
                           extent=extent) as hip:
        got = hip.run(ins)
    except util.SodaError:
      refused += 1
      continue
    ran += 1
    for o in stencil.output_names:
      if not np.array_equal(got[o], want[o], equal_nan=True):
        failed += 1
        print('seed %d extent %s %s output %s: %d cells differ\n%s' %
              (seed, extent, kw, o, int((got[o] != want[o]).sum()),
               prog.soda_text()), flush=True)
    if ran % 25 == 0:
      print('... %d programs, %d failures, %d refused, %.0f s' %
            (ran, failed, refused, time.time() - t0), flush=True)
  print('wide seeds [%d, %d): %d programs run, %d refused, %d failures' %
        (first, last, ran, refused, failed))
  return 1 if failed else 0


def deep_scan(first, last):
  """Deep temporal blocking on random programs (round 4: the scans above fuse
  at most 3 iterations, bench.py fuses 13): iterable 2-D programs of the plain
  and the independent-nest generators run 8-26 iterations with fusion depths
  drawn from {13, 12, 8, 4} (depths whose registers do not fit are dropped by
  the lowering) on grids of several strips and chunks, half of them with
  `border: preserve`; against the C oracle AND, for the nest programs, against
  their own C++ nests."""
  import fuzz
  import fuzz_nest
  from oracle import c_oracle
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  ran = failed = refused = 0
  t0 = time.time()
  for seed in range(first, last):
    rng = np.random.default_rng(seed + 53000)
    nest = None
    if rng.random() < 0.5:
      nest, _ = fuzz_nest.program(seed, 'plain')
      text, dim, iterable = nest.soda_text(), nest.dim, \
          len(nest.inputs) == len(nest.outputs) and \
          [t for _, t in nest.inputs] == [s.typ for s in nest.outputs]
    else:
      text, dim, _ = fuzz.program(seed)
      iterable = None
    if dim != 2:
      continue
    iterate = int(rng.choice([8, 9, 12, 13, 16, 26]))
    border = 'preserve' if rng.random() < 0.5 and nest is None else None
    text = '\n'.join('iterate: %d' % iterate if l.startswith('iterate:') else l
                     for l in text.splitlines()) + '\n'
    try:
      stencil = core.from_text(text, **({'border': border} if border else {}))
      if border:
        stencil.check_preserve()
    except util.SodaError:
      continue                    # not iterable
    if iterable is False:
      continue
    extent = (int(rng.choice([520, 777, 1100])), int(rng.integers(150, 420)))
    lo, hi = stencil.valid_box(extent)
    if not border and not all(h > l + 8 for l, h in zip(lo, hi)):
      continue
    depths = [t for t in (13, 12, 8, 4) if rng.random() < 0.6] or [13]
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    kw = dict(fuse=tuple(depths), chunk_rows=pick([None, None, 40, 64]),
              lane_shift=pick([None, None, 'dpp', 'mixh']))
    if nest is not None:
      nest.iterate = iterate
      ins = fuzz_nest.inputs_for(nest, extent, seed)
    else:
      ins = fuzz.inputs_for(stencil, extent, seed)
    what = 'seed %d %s iterate %d%s extent %s %s' % (
        seed, 'nest' if nest else 'plain', iterate,
        ' preserve' if border else '', extent, kw)
    try:
      with runtime.Program(stencil, lower.LowerOptions(**kw),
                           extent=extent) as prog:
        got = prog.run(ins)
        fused = max(p.fused_iters for p in prog.module.passes)
    except (util.SodaError, ValueError) as e:
      refused += 1
      print('%s: refused: %s' % (what, str(e)[:200]), flush=True)
      continue
    want = c_oracle.COracle(stencil).run(ins)
    own = nest.run(ins, extent) if nest is not None else None
    ran += 1
    for o in stencil.output_names:
      if border:
        g, w = got[o], want[o]
      else:
        lo, hi = stencil.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        g, w = got[o][idx], want[o][idx]
      bad = not np.array_equal(g, w, equal_nan=True)
      if own is not None and not np.array_equal(got[o], own[o],
                                                equal_nan=True):
        bad = True
      if bad:
        failed += 1
        print('%s (deepest pass %d) output %s: %d cells differ\n%s' %
              (what, fused, o, int((g != w).sum()), text), flush=True)
    if ran % 10 == 0:
      print('... %d programs, %d failures, %d refused, %.0f s' %
            (ran, failed, refused, time.time() - t0), flush=True)
  print('deep seeds [%d, %d): %d programs run, %d refused, %d failures' %
        (first, last, ran, refused, failed))
  return 1 if failed else 0


if __name__ == '__main__':
  if sys.argv[1] == 'wide':
    sys.exit(wide_scan(int(sys.argv[2]), int(sys.argv[3])))
  if sys.argv[1] == 'ranks':
    sys.exit(ranks_scan(int(sys.argv[2]), int(sys.argv[3])))
  if sys.argv[1] == 'deep':
    sys.exit(deep_scan(int(sys.argv[2]), int(sys.argv[3])))
  if sys.argv[1] == 'wire':
    sys.exit(wire_scan(int(sys.argv[2]), int(sys.argv[3])))
  if sys.argv[1] == 'options':     # optional 4th argument: only programs containing it
    sys.exit(options_scan(int(sys.argv[2]), int(sys.argv[3]),
                          sys.argv[4] if len(sys.argv) > 4 else None))
  if sys.argv[1] == 'group':
    sys.exit(group_scan(int(sys.argv[2]), int(sys.argv[3])))
  sys.exit(main())

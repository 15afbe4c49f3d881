"""One host thread, N GPUs: the slab group behind the C ABI (soda_hip_group_*,
soda_hip_run_device_slab; include/soda_hip.h).

The reference's host is one blocking sequence on one device (reference
src/soda/codegen/frt/host.py:319-322) that tiles with a replicated halo
(frt/host.py:124-128); the group is its N-GPU form.  CPU tests pin the launch
planning (which chunks of a pass may run while a halo exchange is in flight)
by brute force and the choice of the exchange interval; GPU tests run N
"virtual devices" on the one GPU of the test box -- the whole schedule with its
streams, events and copies -- against the single-GPU run and the oracle, bit
for bit.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, soda_path


def _plan(name, extent, fuse, iterate=None, **kw):
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), **(
      {'iterate': iterate} if iterate else {}))
  opts = runtime.resolve_options(stencil, lower.LowerOptions(fuse=fuse, **kw),
                                 extent)
  mod = lower.lower(stencil, opts)
  code = runtime.compile_source(mod.source, '%s.hip' % stencil.app_name)
  return stencil, runtime.make_plan(mod, runtime.kernel_resources(code))


# ---------------------------------------------------------------------------
# CPU: launch planning
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('name,extent,fuse,iterate,ghost', [
    ('jacobi2d.soda', (8192, 1224), (12, 8, 4), 100, 100),
    ('jacobi2d.soda', (8192, 1264), (12, 4), 120, 120),
    ('jacobi2d.soda', (512, 300), (4,), 9, 12),
    ('jacobi2d.soda', (512, 300), (), 5, 5),
    ('heat3d.soda', (512, 512, 80), (2,), 8, 8),
    ('heat3d.soda', (64, 48, 40), (2,), 3, 4),
    ('blur.soda', (2048, 600), (), 1, 2),
])
def test_split_passes_keep_clear_of_the_exchange(built, name, extent, fuse,
                                                 iterate, ghost):
  """Brute force over the chunks of every split pass: an interior chunk of the
  first pass reads no ghost row, an interior chunk of the last pass delivers no
  row a neighbour fetches and writes no ghost row of the result."""
  from soda_amd import runtime
  stencil, plan = _plan(name, extent, fuse, iterate if iterate > 1 else None)
  reach_lo, reach_hi = stencil.reach_along(stencil.dim - 1)
  rows = extent[-1]
  seen_split = 0
  for sides in ((1, 1), (0, 1), (1, 0)):
    g_lo, g_hi = ghost * sides[0] * min(1, reach_lo), \
        ghost * sides[1] * min(1, reach_hi)
    keep = (g_lo, rows - g_hi)
    for exchanged in (True, False):
      run = runtime.SlabRun(keep[0], keep[1], reach_lo, reach_hi,
                            g_lo if exchanged else 0, g_hi if exchanged else 0,
                            0, 0, 1 if exchanged else None, 1)
      # (send rows: what the neighbour on that side keeps as ITS ghosts)
      run.send_lo = ghost * reach_hi if sides[0] else 0
      run.send_hi = ghost * reach_lo if sides[1] else 0
      launches = runtime.plan_launches(plan, extent, iterate, run)
      assert sum(l['fused_iters'] for l in launches) == iterate
      assert launches[0]['wait'] == (1 if exchanged else 0)
      assert [l['wait'] for l in launches[1:]] == [0] * (len(launches) - 1)
      assert launches[-1]['record'] == 1
      assert [l['record'] for l in launches[:-1]] == [0] * (len(launches) - 1)
      for l in launches:
        if not l['split']:
          continue
        seen_split += 1
        assert 0 <= l['bnd_lo'] < l['bnd_hi'] <= l['chunks']
        assert l['bnd_lo'] > 0 or l['bnd_hi'] < l['chunks']
        n = l['hi'] - l['lo']
        assert l['chunks'] == -(-n // l['chunk'])
        t = l['fused_iters']
        for c in range(l['bnd_lo'], l['bnd_hi']):
          first = l['lo'] + c * l['chunk']
          end = l['lo'] + min((c + 1) * l['chunk'], n)
          if l['wait'] and run.ghost_lo:
            assert first - t * reach_lo >= run.ghost_lo
          if l['wait'] and run.ghost_hi:
            assert end + t * reach_hi <= rows - run.ghost_hi
          if l['record'] and run.send_lo:
            assert first >= keep[0] + run.send_lo
          if l['record'] and run.send_hi:
            assert end <= keep[1] - run.send_hi
          if l['record']:      # behind `sendable` only kept rows are written
            assert first >= keep[0] and end <= keep[1]
  if iterate > 1 or name == 'blur.soda':
    assert seen_split, 'no pass of any configuration was split'


def test_launch_planning_without_a_slab_is_the_plain_schedule(built):
  from soda_amd import runtime
  _, plan = _plan('jacobi2d.soda', (8192, 8192), (12, 8, 4), 100)
  launches = runtime.plan_launches(plan, (8192, 8192), 100)
  assert sum(l['fused_iters'] for l in launches) == 100
  assert all((l['lo'], l['hi'], l['split'], l['wait'], l['record']) ==
             (0, 8192, 0, 0, 0) for l in launches)
  counts = runtime.plan_schedule(plan, (8192, 8192), 100)
  assert len(launches) == sum(counts)


def _desc(extent, n, reach, iterate, every=0, flags=0):
  from soda_amd import runtime
  d = runtime.GroupDesc()
  d.num_slabs = n
  for i, e in enumerate(extent):
    d.extent[i] = e
  d.reach_lo, d.reach_hi = reach
  d.iterate = iterate
  d.exchange_every = every
  d.flags = flags
  return d


def _interval(plan, desc):
  from soda_amd import runtime
  k = ctypes.c_int32(-1)
  rc = runtime.library().soda_hip_group_plan(ctypes.byref(plan),
                                             ctypes.byref(desc),
                                             ctypes.byref(k))
  return rc, k.value


def test_exchange_interval_is_chosen_by_the_library(built):
  """BASELINE's multi-GPU configs on 8 slabs: the interval comes out of the
  per-extent pass times + the transfer model, is a whole number of the deepest
  pass, and never asks a neighbour for rows it does not own."""
  from soda_amd import runtime
  _, plan = _plan('jacobi2d.soda', (8192, 1224), (12, 8, 4), 100)
  rc, k = _interval(plan, _desc((8192, 8192), 8, (1, 1), 100))
  assert rc == 0 and k in (36, 48, 60, 100)        # C2: 1 to 3 exchanges
  rc, k = _interval(plan, _desc((8192, 8192), 8, (1, 1), 1000))
  assert rc == 0 and k % 12 == 0 and 24 <= k <= 1024      # C5
  _, plan3 = _plan('heat3d.soda', (512, 512, 80), (2,), 50)
  rc, k = _interval(plan3, _desc((512, 512, 512), 8, (1, 1), 50))
  assert rc == 0 and k % 2 == 0 and 2 <= k <= 50          # C4
  # one slab: nothing to exchange; an explicit interval is taken as given
  assert _interval(plan, _desc((8192, 8192), 1, (1, 1), 100)) == (0, 100)
  assert _interval(plan, _desc((8192, 8192), 8, (1, 1), 100, every=7)) == (0, 7)
  # too thin: 8192 rows over 64 slabs = 128 rows cannot serve 200 ghost rows
  rc, _ = _interval(plan, _desc((8192, 8192), 64, (1, 1), 1000, every=200))
  assert rc == 1 and 'thinner than' in runtime.last_error()
  # a program that cannot iterate
  _, blur = _plan('blur.soda', (2048, 600), ())
  rc, k = _interval(blur, _desc((2048, 2048), 4, (0, 2), 1))
  assert (rc, k) == (0, 1)
  _, den = _plan('denoise2d.soda', (512, 300), ())      # 2 inputs, 1 output
  rc, _ = _interval(den, _desc((512, 1200), 4, (2, 2), 2))
  assert rc == 1 and 'same as output' in runtime.last_error()


def test_group_needs_a_gpu(built):
  from soda_amd import core, runtime
  if runtime.device_count() > 0:
    pytest.skip('a GPU is present')
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=4)
  with pytest.raises(Exception) as err:
    runtime.Group(stencil, (256, 256), [0, 0])
  assert 'GPU' in str(err.value) or 'device' in str(err.value)


def test_geometry_of_a_tall_stream_is_cheap(built):
  """chunk sizing searched every chunk length, 59 ms of host time
  per run on an 8192 x 2M-row stream; only chunk COUNTS matter (O(sqrt n))."""
  import time
  from soda_amd import runtime
  _, plan = _plan('jacobi2d.soda', (8192, 8192), (12, 8, 4), 100)
  for rows in (8192, 65536, 1 << 21):
    t0 = time.perf_counter()
    tiles, _ = runtime.plan_geometry(plan, (1024, rows))
    dt = time.perf_counter() - t0
    assert dt < 0.01, (rows, dt)
    assert all(1 <= t[1] <= rows for t in tiles)


# ---------------------------------------------------------------------------
# GPU: N virtual devices on the one GPU
# ---------------------------------------------------------------------------

def _inputs(stencil, extent, seed=7):
  rng = np.random.default_rng(seed)
  shape = tuple(extent[::-1])
  out = {}
  for n, t in zip(stencil.input_names, stencil.input_types):
    if t.is_float:
      out[n] = rng.random(shape, dtype=np.float64).astype(t.np_name)
    else:
      out[n] = rng.integers(0, 30000, shape).astype(t.np_name)
  for pstmt in stencil.param_stmts:
    out[pstmt.name] = rng.random(pstmt.size or (1,), dtype=np.float64).astype(
        pstmt.haoda_type.np_name)
  return out


def _oracle(stencil, inputs, iterate):
  from oracle import c_oracle
  return c_oracle.COracle(stencil).run(inputs, iterate=iterate)


def _check(stencil, extent, got, want, iterate):
  for name in stencil.output_names:
    lo, hi = stencil.valid_box(extent, name, iterate)
    idx = tuple(slice(l, max(l, h)) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(got[name][idx], want[name][idx]), name


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent,iterate,fuse,slabs,every,border', [
    ('jacobi2d.soda', (512, 480), 23, (4,), 3, 4, None),
    ('jacobi2d.soda', (512, 480), 23, (4,), 8, 0, None),
    ('jacobi2d.soda', (1024, 1600), 40, (12, 4), 4, 12, None),
    ('jacobi2d.soda', (512, 480), 7, (), 2, 1, None),
    ('jacobi2d.soda', (512, 480), 9, (4,), 3, 4, 'preserve'),
    ('heat3d.soda', (64, 48, 96), 9, (2,), 4, 2, None),
    ('heat3d.soda', (64, 48, 96), 6, (), 3, 0, None),
    ('blur.soda', (2048, 600), 1, (), 4, 0, None),
    ('denoise2d.soda', (512, 300), 1, (), 3, 0, None),
    ('sobel2d.soda', (512, 300), 1, (), 5, 0, None),
    ('conv2d.soda', (512, 300), 1, (), 2, 0, None),
])
@pytest.mark.parametrize('overlap', [True, False])
def test_virtual_slabs_equal_one_gpu(built, name, extent, iterate, fuse, slabs,
                                     every, border, overlap):
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  kw = {'border': border} if border else {}
  stencil = core.from_file(soda_path(name), iterate=iterate, **kw)
  inputs = _inputs(stencil, extent)
  want = _oracle(stencil, inputs, iterate)
  with runtime.Group(stencil, extent, [0] * slabs,
                     lower.LowerOptions(fuse=fuse), exchange_every=every,
                     overlap=overlap) as group:
    got = group.run_host(inputs)
    st = group.stats()
  _check(stencil, extent, got, want, iterate)
  if iterate > 1:
    k = st['exchange_every']
    assert every in (0, k)
    assert st['intervals'] == -(-iterate // k)
    assert st['exchanges'] == st['intervals'] - 1     # fresh after load
    sides = 2 * (slabs - 1)
    assert st['copies'] == st['exchanges'] * sides * len(stencil.input_names)
    if overlap and fuse:
      assert st['split_passes'] > 0
  else:
    assert (st['intervals'], st['exchanges'], st['copies']) == (1, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent,iterate,fuse,slabs', [
    ('jacobi2d.soda', (512, 480), 9, (4,), 3),      # 2-D box: a column range
    ('heat3d.soda', (64, 48, 96), 4, (2,), 4),      # 3-D box
    ('blur.soda', (2048, 600), 1, (), 4),
    ('coupled2d.soda', (256, 500), 4, (2,), 2),     # two tensors each way
])
def test_group_on_registered_arrays(built, name, extent, iterate, fuse, slabs):
  """soda_hip_group_load / _store with the caller's arrays in memory pinned by
  soda_hip_host_register (runtime.PinnedBuffer): every slab's rows go by DMA
  from / to where they are -- N links at once on N GPUs, no host thread in
  between -- and the result is the pageable run's, bit for bit, nothing outside
  the valid box touched."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  inputs = _inputs(stencil, extent)
  shape = extent[::-1]
  room = 1 << 22
  with runtime.Group(stencil, extent, [0] * slabs,
                     lower.LowerOptions(fuse=fuse)) as group:
    plain = group.run_host(inputs)
    with runtime.PinnedBuffer(4 * room) as buf:
      pin_in = {}
      for k, n in enumerate(stencil.input_names):
        pin_in[n] = buf.array(shape, inputs[n].dtype, k * room)
        pin_in[n][...] = inputs[n]
      outs = {}
      for k, (n, t) in enumerate(zip(stencil.output_names,
                                     stencil.output_types)):
        outs[n] = buf.array(shape, np.dtype(t.np_name),
                            (len(pin_in) + k) * room)
        outs[n][...] = 77
      group.load(pin_in)
      group.run(iterate)
      got = group.store(iterate, outputs=outs)
      for o in stencil.output_names:
        lo, hi = stencil.valid_box(extent, o, iterate)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
        assert got[o] is outs[o]
        assert np.array_equal(got[o][idx].view(np.uint8),
                              plain[o][idx].view(np.uint8)), o
        mask = np.ones(shape, bool)
        mask[idx] = False
        assert (got[o][mask] == 77).all(), o
        assert (got[o][idx] != 77).any(), o


@pytest.mark.gpu
def test_chained_group_runs_continue_from_the_result(built):
  """run(a) then run(b) == one run of a + b iterations: the second run opens
  with a halo exchange (its ghosts are stale), device-resident state."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  extent = (512, 640)
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=30)
  inputs = _inputs(stencil, extent, 3)
  want = _oracle(stencil, inputs, 30)
  with runtime.Group(stencil, extent, [0] * 4, lower.LowerOptions(fuse=(4,)),
                     iterate=10, exchange_every=8) as group:
    group.load(inputs)
    for _ in range(3):
      group.run(10)
      st = group.stats()
      assert st['intervals'] == 2
    assert st['exchanges'] == 2          # ... the first of them at the start
    got = group.store(30)
    info = group.slab(1)
    assert (info.own_begin, info.own_end) == (160, 320)
    assert (info.ghost_lo, info.ghost_hi) == (8, 8)
    assert info.inputs[0] == info.outputs[0]     # the state IS the result
  _check(stencil, extent, got, want, 30)


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent,iterate,fuse,slabs,every', [
    ('jacobi2d.soda', (512, 960), 47, (4,), 8, 4),
    ('heat3d.soda', (64, 48, 96), 9, (2,), 4, 2),
    ('jacobi2d.soda', (512, 480), 5, (), 3, 1),
])
def test_one_enqueueing_thread_per_slab(built, name, extent, iterate, fuse,
                                        slabs, every):
  """SODA_HIP_GROUP_THREADS: the slabs' launches are enqueued by a thread each
  (a neighbour's `sendable` must be recorded before a copy is ordered behind
  it: the threads hand that over among themselves); same results, run after
  run."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  inputs = _inputs(stencil, extent, 11)
  want = _oracle(stencil, inputs, 2 * iterate)
  with runtime.Group(stencil, extent, [0] * slabs,
                     lower.LowerOptions(fuse=fuse), exchange_every=every,
                     threads=True) as group:
    group.load(inputs)
    group.run()
    first = group.stats()
    group.run()
    st = group.stats()
    got = group.store(2 * iterate)
  _check(stencil, extent, got, want, 2 * iterate)
  rounds = -(-iterate // every)
  assert (first['intervals'], first['exchanges']) == (rounds, rounds - 1)
  assert (st['intervals'], st['exchanges']) == (rounds, rounds)
  assert st['copies'] == rounds * 2 * (slabs - 1)


@pytest.mark.gpu
@pytest.mark.parametrize('threads', [True, False])
def test_many_short_intervals_stay_in_order(built, threads):
  """300 intervals of two iterations on 8 slabs, 60 chained runs: the event
  chain (and, threaded, the hand-over of recorded events between the slabs'
  enqueueing threads) must never let a copy or a pass overtake what it depends
  on.  `border: preserve` keeps the whole grid defined, so every cell of the
  600-iteration result is compared."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import c_oracle
  extent = (256, 640)
  runs, per_run = 60, 10
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=per_run,
                           border='preserve')
  inputs = _inputs(stencil, extent, 23)
  with runtime.Group(stencil, extent, [0] * 8, lower.LowerOptions(fuse=(2,)),
                     exchange_every=2, threads=threads) as group:
    group.load(inputs)
    for _ in range(runs):
      group.run()
    st = group.stats()
    got = group.store(per_run)
  assert (st['intervals'], st['exchanges']) == (5, 5)
  total = core.from_file(soda_path('jacobi2d.soda'), iterate=runs * per_run,
                         border='preserve')
  want = c_oracle.COracle(total).run(inputs)
  assert np.array_equal(got['t0'], want['t0'])


ONE_SIDED = """kernel: upwind
burst width: 64
unroll factor: 2
iterate: 4
input float: u(64, *)
output float: v(0, 0) = (u(0, 0) + u(0, 1) + u(-1, 0) + u(1, 1)) * 0.25f
"""


@pytest.mark.gpu
@pytest.mark.parametrize('threads', [True, False])
@pytest.mark.parametrize('overlap', [True, False])
def test_one_sided_reach_orders_writes_behind_the_neighbours_copies(
    built, threads, overlap):
  """A program that taps upward only: a slab fetches ghost rows from the slab
  above and never from the one below, so nothing in its own dependency chain
  keeps it from running ahead and overwriting rows the slab below has not
  copied yet (found by tools/fuzz_scan.py group: two wrong planes, one run in
  hundreds).  400 intervals of one iteration on 6 uneven slabs, the whole grid
  compared (`border: preserve`)."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import c_oracle
  extent = (512, 227)
  runs, per_run = 40, 10
  stencil = core.from_text(ONE_SIDED, iterate=per_run, border='preserve')
  assert stencil.reach_along(1) == (0, 1)
  inputs = _inputs(stencil, extent, 5)
  with runtime.Group(stencil, extent, [0] * 6, lower.LowerOptions(fuse=(2,)),
                     exchange_every=1, threads=threads,
                     overlap=overlap) as group:
    assert group.slab(0).ghost_lo == 0 and group.slab(0).ghost_hi == 1
    assert group.slab(5).ghost_lo == 0 and group.slab(5).ghost_hi == 0
    group.load(inputs)
    for _ in range(runs):
      group.run()
    got = group.store(per_run)
  total = core.from_text(ONE_SIDED, iterate=runs * per_run, border='preserve')
  want = c_oracle.COracle(total).run(inputs)
  assert np.array_equal(got['v'], want['v'])


C5 = ('jacobi2d.soda', (8192, 8192), 1000, (12, 4), 3)
C4 = ('heat3d.soda', (512, 512, 512), 50, (2,), 2)


@pytest.mark.gpu
@pytest.mark.parametrize('config', [C5, C4])
def test_baseline_multi_gpu_configs_as_eight_virtual_slabs(built, config):
  """BASELINE C4 / C5 as written, through the C entry: eight slabs on the one
  GPU, exchanges overlapped, equal to the oracle and to the single-GPU run."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  import test_baseline_configs as base
  name, extent, iterate, fuse, seed = config
  from soda_amd import core
  stencil = core.from_file(soda_path(name), iterate=iterate)
  field = base._field(extent, seed)
  inp, out = stencil.input_names[0], stencil.output_names[0]
  with runtime.Group(stencil, extent, [0] * 8, lower.LowerOptions(fuse=fuse),
                     calibrate=True) as group:
    got = group.run_host({inp: field})[out]
    st = group.stats()
  assert st['exchanges'] >= 1 and st['split_passes'] >= 2 * st['exchanges']
  idx, _, _ = base._box(stencil, extent)
  want = base._oracle(name, extent, iterate, seed, 'random')
  assert np.array_equal(got[idx], want[idx])
  with runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                       extent=extent) as prog:
    single = prog.run({inp: field})[out]
  assert np.array_equal(got[idx], single[idx])


def test_split_passes_with_a_one_sided_reach(built):
  """Launch planning for a program that reaches upward only (ONE_SIDED): a slab
  has ghost rows above and none below, sends rows below and none above; the
  boundary chunks of the first / last pass sit on ONE side of the grid, and by
  brute force no interior chunk of a first pass reads a ghost row, none of a
  last pass delivers a row the neighbour below fetches."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_text(ONE_SIDED, iterate=8)
  extent = (512, 300)
  opts = runtime.resolve_options(stencil, lower.LowerOptions(fuse=(4,)), extent)
  mod = lower.lower(stencil, opts)
  code = runtime.compile_source(mod.source, '%s.hip' % stencil.app_name)
  plan = runtime.make_plan(mod, runtime.kernel_resources(code))
  reach_lo, reach_hi = stencil.reach_along(1)
  assert (reach_lo, reach_hi) == (0, 1)
  rows, ghost = extent[-1], 8
  seen = 0
  for below, above, exchanged in ((True, True, True), (False, True, True),
                                  (True, False, True), (True, True, False),
                                  (False, True, False)):
    g_hi = ghost * reach_hi if above else 0        # ghosts: above only
    keep = (0, rows - g_hi)
    # `exchanged` False: the run opens on fresh ghosts -- nothing to wait for,
    # no ghost rows named -- but the rows above the kept range are still the
    # RESULT's ghost rows, which the exchange that follows overwrites: the
    # chunks that write them must go in front of `sendable` all the same (the
    # hole tools/flake_loop.py found in round 4: only the side that SENDS
    # pulled its end chunks into the boundary)
    run = runtime.SlabRun(keep[0], keep[1], reach_lo, reach_hi, 0,
                          g_hi if exchanged else 0, 0, 0,
                          1 if g_hi and exchanged else None, 1)
    run.send_lo = ghost * reach_hi if below else 0   # the slab below fetches
    run.send_hi = 0                                  # nobody above does
    launches = runtime.plan_launches(plan, extent, 8, run)
    assert sum(l['fused_iters'] for l in launches) == 8
    last = launches[-1]
    assert last['record']
    if g_hi:       # the top chunk writes ghost rows: never behind `sendable`
      assert last['split'] == 0 or last['bnd_hi'] < last['chunks']
    for l in launches:
      if not l['split']:
        continue
      seen += 1
      t, n = l['fused_iters'], l['hi'] - l['lo']
      for c in range(l['bnd_lo'], l['bnd_hi']):      # the interior chunks
        first = l['lo'] + c * l['chunk']
        end = l['lo'] + min((c + 1) * l['chunk'], n)
        if l['wait'] and g_hi:
          assert end + t * reach_hi <= rows - g_hi
        if l['record'] and run.send_lo:
          assert first >= keep[0] + run.send_lo
        if l['record']:
          assert first >= keep[0] and end <= keep[1]
      if l['wait'] and not l['record']:
        assert l['bnd_lo'] == 0          # nothing to wait for at the low end
  assert seen


@pytest.mark.gpu
def test_bench_group_mode_checks_its_result(built):
  """`bench.py --group --virtual`: one process drives four slabs on the one
  GPU through soda_hip_group_*, both exchange modes timed, and the JSON line
  carries the result check -- two chained steps (the second opens with the
  exchange) against the C oracle."""
  import json
  import subprocess
  import sys
  from conftest import ROOT
  run = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--group', '--virtual',
       '--gpus', '4', '--steps', '3', '--warmup', '1', '--extent', '2048',
       '1600', '--iterate', '24', '--fuse', '12', '4', '--exchange-every',
       '12'], capture_output=True, text=True, timeout=600)
  assert run.returncode == 0, run.stderr[-3000:]
  out = json.loads([l for l in run.stdout.splitlines()
                    if l.startswith('{')][-1])
  assert out['n_gpus'] == 4 and out['config']['exchanges_per_step'] == 2
  assert out['parity']['mismatches'] == 0
  assert out['parity']['cells'] == (2048 - 96) * (1600 - 96)

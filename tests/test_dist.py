"""Slab decomposition + halo exchange over gloo (world_size 2 and 3, CPU).

The decomposition/exchange code is the one the GPU ranks run; only the local
compute engine is swapped for the CPU oracle (test infrastructure), since this
box has no GPU.  The gathered result must equal the single-process oracle on
the global valid box, bit for bit."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, soda_path


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, name, extent, iterate, every, out_dir,
            border=None):
  import sys
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import torch
  import torch.distributed as tdist
  from soda_amd import core, dist as sdist
  from oracle import numpy_oracle
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  tdist.init_process_group('gloo', rank=rank, world_size=world)
  stencil = core.from_file(soda_path(name), iterate=iterate, border=border)
  slab = sdist.Slab(stencil, extent, world, rank, every)
  rng = np.random.default_rng(42)
  full = {}
  for n, t in zip(stencil.input_names, stencil.input_types):
    if t.is_float:
      full[n] = rng.random(tuple(extent[::-1]), dtype=np.float32)
    else:
      full[n] = rng.integers(0, 60000, tuple(extent[::-1])).astype(t.np_name)
  cur = [torch.from_numpy(full[n][slab.begin:slab.end].copy())
         for n in stencil.input_names]
  keep = [t.clone() for t in cur]
  work_a = [torch.empty_like(t) for t in cur]
  work_b = [torch.empty_like(t) for t in cur]

  def step(dst, src, lext, iters):
    ins = {n: s.numpy() for n, s in zip(stencil.input_names, src)}
    outs = numpy_oracle.run(stencil, ins, iterate=iters, origin=slab.origin,
                            global_extent=slab.extent)
    for d, o in zip(dst, stencil.output_names):
      d.copy_(torch.from_numpy(outs[o]))

  res = sdist.run(slab, cur, work_a, work_b, step, iterate, tdist)
  assert all(torch.equal(a, b) for a, b in zip(cur, keep)), 'inputs written'
  own = [r[slab.ghost_lo:slab.ghost_lo + slab.own_rows].numpy() for r in res]
  np.save(os.path.join(out_dir, 'rank%d.npy' % rank), own[0])
  tdist.barrier()
  tdist.destroy_process_group()


@pytest.mark.parametrize('name,extent,iterate,every,world', [
    ('jacobi2d.soda', (40, 64), 7, 3, 2),
    ('jacobi2d.soda', (40, 61), 9, 4, 3),
    ('heat3d.soda', (12, 10, 30), 4, 2, 2),
    ('blur.soda', (40, 50), 3, 2, 2),      # one-sided halo (taps 0..2)
    ('skew2d.soda', (30, 40), 1, 1, 2),    # two-stage, asymmetric
])
@pytest.mark.parametrize('border', [None, 'preserve'])
def test_slabs_match_single_process(tmp_path, name, extent, iterate, every,
                                    world, border):
  import torch.multiprocessing as mp
  from soda_amd import core, util
  from oracle import numpy_oracle
  stencil = core.from_file(soda_path(name), iterate=iterate, border=border)
  try:
    stencil.check_preserve()
  except util.SemanticError:
    pytest.skip('border: preserve does not apply to this program')
  port = _free_port()
  mp.spawn(_worker, args=(world, port, name, extent, iterate, every,
                          str(tmp_path), border), nprocs=world, join=True)
  rng = np.random.default_rng(42)
  full = {}
  for n, t in zip(stencil.input_names, stencil.input_types):
    if t.is_float:
      full[n] = rng.random(tuple(extent[::-1]), dtype=np.float32)
    else:
      full[n] = rng.integers(0, 60000, tuple(extent[::-1])).astype(t.np_name)
  want = numpy_oracle.run(stencil, full)[stencil.output_names[0]]
  got = np.concatenate([np.load(os.path.join(str(tmp_path), 'rank%d.npy' % r))
                        for r in range(world)], axis=0)
  assert got.shape == want.shape
  lo, hi = stencil.valid_box(extent)
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  assert np.array_equal(got[idx], want[idx])


def _fuzz_worker(rank, world, port, seed, every, out_dir):
  import sys
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import torch
  import torch.distributed as tdist
  import fuzz
  from soda_amd import core, dist as sdist
  from oracle import numpy_oracle
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  tdist.init_process_group('gloo', rank=rank, world_size=world)
  text, dim, iterate = fuzz.program(seed)
  stencil = core.from_text(text)
  extent = list(fuzz.extent_for(seed, dim))
  extent[-1] = max(extent[-1], 48)
  slab = sdist.Slab(stencil, extent, world, rank, every)
  full = fuzz.inputs_for(stencil, extent, seed)
  src = [torch.from_numpy(full[n][slab.begin:slab.end].copy())
         for n in stencil.input_names]
  work_a = [torch.empty_like(t) for t in src]
  work_b = [torch.empty_like(t) for t in src]

  def step(dst, cur, lext, iters):
    ins = {n: s.numpy() for n, s in zip(stencil.input_names, cur)}
    outs = numpy_oracle.run(stencil, ins, iterate=iters)
    for d, o in zip(dst, stencil.output_names):
      d.copy_(torch.from_numpy(outs[o]))

  res = sdist.run(slab, src, work_a, work_b, step, stencil.iterate, tdist)
  for i, r in enumerate(res):
    own = r[slab.ghost_lo:slab.ghost_lo + slab.own_rows].numpy()
    np.save(os.path.join(out_dir, 'rank%d_out%d.npy' % (rank, i)), own)
  tdist.barrier()
  tdist.destroy_process_group()


def _iterable_fuzz_seeds(count):
  import fuzz
  from soda_amd import core
  out = []
  seed = 0
  while len(out) < count:
    text, dim, iterate = fuzz.program(seed)
    if iterate >= 2 and dim >= 2:
      try:
        core.from_text(text)
        out.append(seed)
      except Exception:
        pass
    seed += 1
  return out


@pytest.mark.parametrize('seed', _iterable_fuzz_seeds(5))
def test_random_programs_on_two_slabs(tmp_path, seed):
  """Random iterated multi-stage / multi-input programs (asymmetric reach,
  non-zero store indices) cut into 2 slabs with an exchange every iteration."""
  import torch.multiprocessing as mp
  import fuzz
  from soda_amd import core
  from oracle import numpy_oracle
  world = 2
  port = _free_port()
  mp.spawn(_fuzz_worker, args=(world, port, seed, 1, str(tmp_path)),
           nprocs=world, join=True)
  text, dim, iterate = fuzz.program(seed)
  stencil = core.from_text(text)
  extent = list(fuzz.extent_for(seed, dim))
  extent[-1] = max(extent[-1], 48)
  full = fuzz.inputs_for(stencil, extent, seed)
  want = numpy_oracle.run(stencil, full)
  for i, o in enumerate(stencil.output_names):
    got = np.concatenate([
        np.load(os.path.join(str(tmp_path), 'rank%d_out%d.npy' % (r, i)))
        for r in range(world)], axis=0)
    lo, hi = stencil.valid_box(extent, o)
    if not all(h > l for l, h in zip(lo, hi)):
      continue
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(got[idx], want[o][idx], equal_nan=True), text


def test_slab_geometry():
  from soda_amd import core, dist as sdist
  st = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  slabs = [sdist.Slab(st, (8192, 8192), 8, r, 24) for r in range(8)]
  assert [s.own_rows for s in slabs] == [1024] * 8
  assert slabs[0].ghost_lo == 0 and slabs[0].ghost_hi == 24
  assert slabs[3].local_extent == (8192, 1024 + 48)
  assert slabs[7].ghost_hi == 0 and slabs[7].end == 8192
  # what rank 3 sends up is what rank 4 receives below, row for row
  up = [m for m in slabs[3].messages() if m[0] == 4][0]
  dn = [m for m in slabs[4].messages() if m[0] == 3][0]
  assert up[1][1] - up[1][0] == dn[2][1] - dn[2][0] == 24
  assert slabs[3].begin + up[1][0] == slabs[4].begin + dn[2][0]
  assert sdist.rounds(100, 24) == 5
  assert sdist.auto_exchange_every(st, (8192, 8192), 8, 100, 12) == 100
  assert sdist.auto_exchange_every(st, (8192, 8192), 1, 100) == 100
  j1000 = core.from_file(soda_path('jacobi2d.soda'), iterate=1000)
  assert sdist.auto_exchange_every(j1000, (8192, 8192), 8, 1000, 12) == 120
  h = core.from_file(soda_path('heat3d.soda'), iterate=50)
  assert sdist.auto_exchange_every(h, (512, 512, 512), 8, 50) == 8
  uneven = [sdist.Slab(st, (64, 10), 3, r, 1) for r in range(3)]
  assert [s.own_rows for s in uneven] == [4, 3, 3]
  from soda_amd import util
  with pytest.raises(util.InputError, match='thinner'):
    sdist.Slab(st, (64, 16), 8, 1, 4)
  b = core.from_file(soda_path('blur.soda'))
  s = sdist.Slab(b, (64, 64), 2, 0, 3)
  assert (s.reach_lo, s.reach_hi, s.ghost_hi) == (0, 2, 6)


def _full_inputs(stencil, extent, rng):
  shape = tuple(extent[::-1])
  return {n: (rng.random(shape, dtype=np.float32) if t.is_float else
              rng.integers(0, 30000, shape).astype(t.np_name))
          for n, t in zip(stencil.input_names, stencil.input_types)}


def _gpu_worker(rank, world, port, name, extent, iterate, every, fuse, out_dir,
                border=None, strategy='auto'):
  """Two ranks share the one GPU of the box; the halo exchange runs over gloo
  on host tensors, the K iterations between exchanges on the GPU kernels."""
  import sys
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import torch
  import torch.distributed as tdist
  from soda_amd import core, dist as sdist, runtime
  from soda_amd.codegen.hip import lower
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  tdist.init_process_group('gloo', rank=rank, world_size=world)
  stencil = core.from_file(soda_path(name), iterate=iterate, border=border)
  slab = sdist.Slab(stencil, extent, world, rank, every)
  rng = np.random.default_rng(7)
  full = _full_inputs(stencil, extent, rng)
  src = [torch.from_numpy(full[n][slab.begin:slab.end].copy())
         for n in stencil.input_names]
  work_a = [torch.empty_like(t) for t in src]
  work_b = [torch.empty_like(t) for t in src]
  prog = runtime.Program(stencil,
                         lower.LowerOptions(fuse=fuse, strategy=strategy),
                         device=0, extent=slab.local_extent)

  def step(dst, cur, lext, iters):
    d_in = [t.cuda() for t in cur]
    d_out = [torch.empty_like(t) for t in d_in]
    prog.run_device([t.data_ptr() for t in d_out],
                    [t.data_ptr() for t in d_in], lext, iterate=iters,
                    stream=torch.cuda.current_stream().cuda_stream,
                    origin=slab.origin, global_extent=slab.extent,
                    keep=slab.keep)
    torch.cuda.synchronize()
    for h, d in zip(dst, d_out):
      h.copy_(d.cpu())

  res = sdist.run(slab, src, work_a, work_b, step, iterate, tdist)
  own = res[0][slab.ghost_lo:slab.ghost_lo + slab.own_rows].numpy()
  np.save(os.path.join(out_dir, 'rank%d.npy' % rank), own)
  tdist.barrier()
  prog.close()
  tdist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent,iterate,every,fuse,border,strategy', [
    ('jacobi2d.soda', (1000, 400), 14, 6, (3,), None, 'auto'),  # 3 rounds
    ('heat3d.soda', (260, 24, 60), 5, 2, (), None, 'auto'),
    # border: preserve -- the kernels must tell the grid's border (kept) from
    # the seam between the slabs (computed)
    ('jacobi2d.soda', (1000, 400), 14, 6, (3,), 'preserve', 'auto'),
    ('heat3d.soda', (260, 24, 60), 6, 2, (2,), 'preserve', 'auto'),
    ('seidel2d.soda', (640, 300), 4, 2, (), 'preserve', 'direct'),
])
def test_two_ranks_on_gpu_kernels(tmp_path, name, extent, iterate, every, fuse,
                                  border, strategy):
  """The GPU engine inside the slab/exchange loop (exchanges really happen:
  K < iterate), against the single-process oracle on the global valid box."""
  import torch.multiprocessing as mp
  from soda_amd import core
  from oracle import c_oracle
  world = 2
  port = _free_port()
  mp.spawn(_gpu_worker, args=(world, port, name, extent, iterate, every, fuse,
                              str(tmp_path), border, strategy), nprocs=world,
           join=True)
  stencil = core.from_file(soda_path(name), iterate=iterate, border=border)
  rng = np.random.default_rng(7)
  full = _full_inputs(stencil, extent, rng)
  want = c_oracle.COracle(stencil).run(full)[stencil.output_names[0]]
  got = np.concatenate([np.load(os.path.join(str(tmp_path), 'rank%d.npy' % r))
                        for r in range(world)], axis=0)
  lo, hi = stencil.valid_box(extent)
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  assert np.array_equal(got[idx], want[idx])


UPWIND = """kernel: upwind
burst width: 64
unroll factor: 2
iterate: 4
input float: u(64, *)
output float: v(0, 0) = (u(0, 0) + u(0, 1) + u(-1, 0) + u(1, 1)) * 0.25f
"""


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent,iterate,every,fuse,world', [
    ('jacobi2d.soda', (1024, 1600), 40, 12, (12, 4), 4),
    ('jacobi2d.soda', (1024, 1600), 39, 13, (13, 12, 8, 4), 4),  # benched depths
    ('jacobi2d.soda', (512, 480), 23, 4, (4,), 8),     # one pass per interval
    ('heat3d.soda', (64, 48, 96), 9, 2, (2,), 4),
    # a one-sided reach: a rank fetches from the rank above and never from the
    # one below -- nothing it receives keeps it from running ahead of the rank
    # that still has to fetch its rows (the hole the slab group had)
    (UPWIND, (512, 227), 10, 1, (), 6),
    (UPWIND, (512, 454), 12, 4, (4,), 5),
])
def test_exchange_hidden_under_the_compute(name, extent, iterate, every, fuse,
                                           world):
  """dist.StreamOverlap + Program.run_device(ghosts=..., sends=..., events):
  every rank (a thread with a compute stream of its own; messages by
  tests/fabric.py, which orders the receiver's stream behind the sender's copy
  as RCCL does) runs its exchanges on a second stream while the rows that need
  no fresh ghost compute; two chained runs, equal to the oracle bit for bit --
  also with one rank held back on the GPU and another on the host."""
  import overlap_case
  from soda_amd import core, dist as sdist
  from oracle import c_oracle

  def load(iters):
    if name.endswith('.soda'):
      return core.from_file(soda_path(name), iterate=iters)
    return core.from_text(name, iterate=iters)

  stencil = load(iterate)
  rng = np.random.default_rng(5)
  fields = {n: rng.random(tuple(extent[::-1]), dtype=np.float32)
            for n in stencil.input_names}
  again = load(2 * iterate)
  want = c_oracle.COracle(again).run(fields)
  rounds = sdist.rounds(iterate, every)
  one_sided = 0 in stencil.reach_along(stencil.dim - 1)
  with overlap_case.Case(stencil, extent, every, fuse, world) as case:
    for knobs in (dict(), dict(spin={1: 3_000_000}, sleep={world - 1: 0.002})):
      got, messages = case.trial(fields, iterate, runs=2, **knobs)
      assert overlap_case.mismatches(again, extent, got, want,
                                     2 * iterate) == 0, knobs
      if one_sided:      # an end rank only sends or only receives
        assert messages[1] == 2 * rounds - 1
      else:
        assert messages[0] == 2 * rounds - 1
        assert messages[1] == 2 * (2 * rounds - 1)
    assert one_sided or case.splits > 0, 'no pass was ever split'


@pytest.mark.gpu
def test_many_short_overlapped_intervals_stay_in_order():
  """The rank-per-GPU counterpart of test_group.py's
  test_many_short_intervals_stay_in_order: 300 intervals of two iterations on
  six ranks, 30 chained runs, ranks skewed against each other on the GPU and on
  the host -- no exchange and no pass may overtake what it depends on.
  `border: preserve` keeps the whole grid defined, so every cell of the
  600-iteration result is compared."""
  import overlap_case
  from soda_amd import core
  from oracle import c_oracle
  extent, world = (256, 636), 6
  runs, per_run = 30, 20
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=per_run,
                           border='preserve')
  rng = np.random.default_rng(23)
  fields = {'t1': rng.random(tuple(extent[::-1]), dtype=np.float32)}
  total = core.from_file(soda_path('jacobi2d.soda'), iterate=runs * per_run,
                         border='preserve')
  want = c_oracle.COracle(total).run(fields)
  with overlap_case.Case(stencil, extent, 2, (2,), world) as case:
    got, messages = case.trial(fields, per_run, runs=runs,
                               spin={2: 400_000}, sleep={4: 0.0003})
    assert case.intervals == world * runs * per_run // 2
  assert messages[0] == runs * per_run // 2 - 1
  assert overlap_case.mismatches(total, extent, got, want, runs * per_run,
                                 whole_grid=True) == 0


def test_thin_slab_is_refused_on_every_rank():
  """n % world != 0: ranks differ by one row.  The halo check judges the
  thinnest slab, so either every rank raises or none does (a rank that passed
  alone would wait forever in the exchange)."""
  from soda_amd import core, dist as sdist, util
  st = core.from_file(soda_path('jacobi2d.soda'), iterate=40)
  # 43 rows over 4 ranks: slabs of 11, 11, 11, 10 rows; halo 10 -> all pass,
  # halo 11 -> ALL refuse, also the three ranks that own 11 rows
  for rank in range(4):
    sdist.Slab(st, (64, 43), 4, rank, 10)
    with pytest.raises(util.InputError, match='thinner'):
      sdist.Slab(st, (64, 43), 4, rank, 11)


def _run_bench(*flags, timeout=600):
  import subprocess
  import sys
  env = dict(os.environ)
  env.pop('WORLD_SIZE', None)
  env.pop('RANK', None)
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags],
                        capture_output=True, text=True, env=env,
                        timeout=timeout)


def test_bench_launcher_refuses_without_enough_gpus():
  """`bench.py --gpus N` started by hand is its own launcher; on a box with
  fewer than N GPUs (this one has none) it says so and exits non-zero at once,
  having launched nothing -- no hang, no traceback."""
  import torch
  if torch.cuda.device_count() >= 2:
    pytest.skip('this box could run it')
  run = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0', timeout=120)
  assert run.returncode == 2
  assert 'needs 2 GPUs' in run.stderr and 'nothing was launched' in run.stderr
  assert 'Traceback' not in run.stderr and not run.stdout.strip()


@pytest.mark.gpu
@pytest.mark.parametrize('overlap', ['auto', 'on', 'off'])
def test_bench_two_ranks_rehearsed_on_one_gpu(overlap):
  """bench.py's N-GPU path end to end under torchrun -- slabs, chained steps,
  the overlapped exchange on a side stream (dist.StreamOverlap through the real
  torch.distributed API), the overlap trial, the JSON line -- with both ranks
  on the ONE GPU of the box and gloo carrying the halos (RCCL refuses two ranks
  on a device).  A functional rehearsal, not a measurement."""
  import json
  import subprocess
  import sys
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
  env.update(SODA_BENCH_ONE_GPU='1', SODA_BENCH_BACKEND='gloo')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
         '--gpus', '2', '--steps', '3', '--warmup', '1', '--extent', '2048',
         '2048', '--iterate', '48', '--fuse', '12', '4', '--exchange-every',
         '24', '--overlap', overlap, '--no-cpu-baseline', '--no-single-iter']
  run = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=600)
  assert run.returncode == 0, run.stderr[-3000:]
  line = [l for l in run.stdout.splitlines() if l.startswith('{')][-1]
  out = json.loads(line)
  assert out['n_gpus'] == 2 and out['value'] > 0
  cfg = out['config']
  assert cfg['exchanges_per_step'] == 2 and cfg['ghost_rows_per_side'] == 24
  assert 'REHEARSAL' in cfg['transport']
  # the step bench.py checks after its timed region: both ranks' rows gathered
  # on rank 0, the whole valid box against the C oracle
  # (two chained steps: the second opens with the exchange)
  lo, hi = 96, 2048 - 96
  assert out['parity'] == dict(out['parity'], mismatches=0,
                               cells=(hi - lo) * (hi - lo))
  assert out['clock_warm_steps'] >= 10
  if overlap == 'auto':
    trial = cfg['overlap_trial']
    assert 'overlapped_failed' not in trial, trial
    assert trial['overlapped_ms_per_step'] > 0 and trial['serial_ms_per_step'] > 0
  else:
    assert cfg['overlap'] == (overlap == 'on')


@pytest.mark.gpu
def test_bench_overlap_trial_survives_a_failing_overlapped_way():
  """`--overlap auto`: when the overlapped step fails on every rank (injected:
  the first thing a never-executed path would do on real links), all ranks go
  through the same collectives, agree on the serial exchange and the run
  completes -- checked result included."""
  import json
  import subprocess
  import sys
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
  env.update(SODA_BENCH_ONE_GPU='1', SODA_BENCH_BACKEND='gloo',
             SODA_BENCH_INJECT_OVERLAP_FAILURE='1')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
         '--gpus', '2', '--steps', '2', '--warmup', '1', '--extent', '2048',
         '2048', '--iterate', '48', '--fuse', '12', '4', '--exchange-every',
         '24', '--overlap', 'auto', '--no-cpu-baseline', '--no-single-iter',
         '--clock-warm-seconds', '0.05']
  run = subprocess.run(cmd, capture_output=True, text=True, env=env,
                       timeout=600)
  assert run.returncode == 0, run.stderr[-3000:]
  out = json.loads([l for l in run.stdout.splitlines()
                    if l.startswith('{')][-1])
  trial = out['config']['overlap_trial']
  assert 'injected failure' in trial['overlapped_failed']
  assert trial['overlapped_ms_per_step'] is None
  assert trial['serial_ms_per_step'] > 0 and out['config']['overlap'] is False
  assert out['parity']['mismatches'] == 0


@pytest.mark.gpu
def test_bench_line_carries_a_result_check():
  """bench.py on one GPU (a small grid, the default depth set): the JSON line
  reports the step it checked against the C oracle after the timed region --
  every cell of the valid box, no mismatch -- and how long it span the clocks
  up; a wrong result would also have made it exit non-zero."""
  import json
  run = _run_bench('--steps', '3', '--warmup', '1', '--extent', '2048', '1536',
                   '--iterate', '38', '--no-cpu-baseline', '--no-single-iter',
                   '--no-rehearsal', '--clock-warm-seconds', '0.05')
  assert run.returncode == 0, run.stderr[-3000:]
  out = json.loads([l for l in run.stdout.splitlines()
                    if l.startswith('{')][-1])
  assert out['parity']['mismatches'] == 0
  assert out['parity']['cells'] == (2048 - 76) * (1536 - 76)
  assert out['clock_warm_steps'] >= 10 and out['value'] > 0
  assert out['roofline']['kernel'].startswith('jacobi2d_march2d_T')


@pytest.mark.gpu
def test_bench_one_rank_over_rccl():
  """The most of bench.py's RCCL path one GPU can execute: a process group of
  ONE rank on the nccl backend (SODA_BENCH_FORCE_DIST) -- RCCL initialises, and
  the barrier and the MAX all-reduces around the clock-warm windows and the
  timed region really run through it.  (Send / receive between two devices is
  what remains for the driver's multi-GPU run.)"""
  import json
  import subprocess
  import sys
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
  env.update(SODA_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1',
             MASTER_PORT=str(_free_port()))
  run = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3',
       '--warmup', '1', '--extent', '2048', '1024', '--iterate', '26',
       '--no-cpu-baseline', '--no-single-iter', '--no-rehearsal',
       '--clock-warm-seconds', '0.05'],
      capture_output=True, text=True, env=env, timeout=600)
  assert run.returncode == 0, run.stderr[-3000:]
  out = json.loads([l for l in run.stdout.splitlines()
                    if l.startswith('{')][-1])
  assert out['rccl_world'] == 1 and out['n_gpus'] == 1
  assert out['parity']['mismatches'] == 0 and out['clock_warm_steps'] >= 10


@pytest.mark.gpu
def test_bench_launcher_on_the_gpu_box():
  """On the GPU box: with two or more GPUs the self-launched 2-rank run must
  come back with one JSON line from rank 0 and `rccl_world` = 2 (the RCCL
  process group really formed); with one GPU the launcher must refuse cleanly."""
  import json
  import torch
  if torch.cuda.device_count() >= 2:
    run = _run_bench('--gpus', '2', '--steps', '2', '--warmup', '1',
                     '--no-cpu-baseline', '--no-single-iter')
    assert run.returncode == 0, run.stderr[-2000:]
    line = [l for l in run.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['rccl_world'] == 2
    assert out['config']['exchanges_per_step'] >= 1
  else:
    run = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0',
                     timeout=300)
    assert run.returncode == 2 and 'needs 2 GPUs' in run.stderr
    assert not run.stdout.strip()


def test_exchange_interval_by_the_librarys_cost_choice(built, caplog,
                                                       monkeypatch):
  """dist.planned_exchange_every: the rank-per-GPU path asks the library
  (soda_hip_group_plan, a pure function) like the one-process group does; the
  headline run keeps its single interval, the long runs exchange a little less
  often than the quarter-slab rule would, and every answer is a whole number of
  the deepest pass and leaves slabs thicker than their ghosts."""
  from soda_amd import core, dist as sdist
  from soda_amd.codegen.hip import lower
  fuse = (13, 12, 8, 4)
  st = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  for world in (2, 4, 8):
    assert sdist.planned_exchange_every(
        st, (8192, 8192), world, 100, lower.LowerOptions(fuse=fuse), 13) == 100
  assert sdist.planned_exchange_every(st, (8192, 8192), 1, 100) == 100
  long = core.from_file(soda_path('jacobi2d.soda'), iterate=1000)
  k = sdist.planned_exchange_every(long, (8192, 8192), 8, 1000,
                                   lower.LowerOptions(fuse=fuse), 13)
  assert k % 13 == 0 and 100 <= k <= 1024
  sdist.Slab(long, (8192, 8192), 8, 3, k)          # (raises if too thin)
  h = core.from_file(soda_path('heat3d.soda'), iterate=50)
  k = sdist.planned_exchange_every(h, (512, 512, 512), 8, 50,
                                   lower.LowerOptions(fuse=(2,)), 2)
  assert k % 2 == 0 and 2 <= k <= 32
  sdist.Slab(h, (512, 512, 512), 8, 3, k)
  # a description the library refuses falls back to the rule
  assert sdist.planned_exchange_every(st, (8192, 8), 8, 100) == \
      sdist.auto_exchange_every(st, (8192, 8), 8, 100)
  # ... and so does a planner that breaks -- but it says so (a bare
  # `except` used to degrade K silently for ever)
  from soda_amd import runtime

  def broken():
    raise RuntimeError('no library today')
  monkeypatch.setattr(runtime, 'library', broken)
  caplog.set_level('WARNING', logger='soda_amd.dist')
  assert sdist.planned_exchange_every(long, (8192, 8192), 8, 1000) == \
      sdist.auto_exchange_every(long, (8192, 8192), 8, 1000)
  assert any('planner failed' in r.getMessage() and 'no library today' in
             r.getMessage() for r in caplog.records)

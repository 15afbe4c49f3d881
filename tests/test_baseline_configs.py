"""BASELINE.json configs 4 and 5 exactly as written (the sizes and iteration
counts the metric is quoted on), checked against the CPU oracle:

  C5  jacobi2d fp32 8192 x 8192, iterate = 1000, temporal blocking T = 4 (as
      BASELINE words it) and T = 12 (the default schedule); valid box
      [1000, 7192)^2;
  C4  heat3d fp32 512^3, iterate = 50, T = 2; valid box [50, 462)^3; random
      input and the p+q+r known answer (coefficients are powers of two that
      sum to 1: the field is a fixed point, bit for bit);

each on one GPU, and cut into the 8 slabs of the 8-GPU run: eight ranks, every
one executing the real decomposition (soda_amd.dist.Slab), the real exchange
schedule (dist.run / dist.exchange, K picked by dist.auto_exchange_every) and
the real kernels -- as eight threads sharing the one GPU of the test box, the
halo messages carried by tests/fabric.py instead of RCCL.  The stitched result
must equal the single-GPU result and the oracle bit for bit on the valid box.
"""
import functools

import numpy as np
import pytest

import fabric
from conftest import soda_path


def _field(extent, seed, kind='random'):
  shape = tuple(extent[::-1])
  if kind == 'ramp':          # the reference harness's init: p + q (+ r)
    return np.indices(shape).sum(axis=0).astype(np.float32)
  return np.random.default_rng(seed).random(shape, dtype=np.float32)


@functools.lru_cache(maxsize=None)
def _oracle(name, extent, iterate, seed, kind):
  from soda_amd import core
  from oracle import c_oracle
  stencil = core.from_file(soda_path(name), iterate=iterate)
  inp = stencil.input_names[0]
  out = c_oracle.COracle(stencil).run({inp: _field(extent, seed, kind)})
  return out[stencil.output_names[0]]


def _box(stencil, extent):
  lo, hi = stencil.valid_box(extent)
  return tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1])), lo, hi


def _run_slabs(stencil, extent, world, fuse, field, engine):
  """`world` ranks as threads over tests/fabric.py; returns the stitched own
  rows and per-rank (exchange rounds, messages sent)."""
  import torch
  from soda_amd import dist as sdist
  iterate = stencil.iterate
  every = sdist.auto_exchange_every(stencil, extent, world, iterate,
                                    multiple_of=max(fuse) if fuse else 1)
  rounds = sdist.rounds(iterate, every)

  def rank_fn(rank, endpoint):
    slab = sdist.Slab(stencil, extent, world, rank, every)
    step, to_dev, to_host = engine(slab)
    src = [to_dev(field[slab.begin:slab.end])]
    work_a = [torch.empty_like(src[0])]
    work_b = [torch.empty_like(src[0])]
    res = sdist.run(slab, src, work_a, work_b, step, iterate, endpoint)
    own = to_host(res[0][slab.ghost_lo:slab.ghost_lo + slab.own_rows])
    return own, endpoint.messages

  results = fabric.run_ranks(world, rank_fn)
  got = np.concatenate([r[0] for r in results], axis=0)
  return got, every, rounds, [r[1] for r in results]


# ---------------------------------------------------------------------------
# CPU: the thread fabric + slab schedule themselves (no GPU)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('name,extent,iterate,world', [
    ('jacobi2d.soda', (48, 160), 23, 8),
    ('heat3d.soda', (12, 10, 96), 9, 8),
])
def test_eight_slabs_over_thread_fabric_cpu(built, name, extent, iterate,
                                            world):
  import torch
  from soda_amd import core
  from oracle import numpy_oracle
  stencil = core.from_file(soda_path(name), iterate=iterate)
  field = _field(extent, 5)

  def engine(slab):
    def step(dst, src, lext, iters):
      outs = numpy_oracle.run(stencil, {stencil.input_names[0]: src[0].numpy()},
                              iterate=iters, origin=slab.origin,
                              global_extent=slab.extent)
      dst[0].copy_(torch.from_numpy(outs[stencil.output_names[0]]))
    return step, lambda a: torch.from_numpy(a.copy()), lambda t: t.numpy()

  got, every, rounds, messages = _run_slabs(stencil, extent, world, (), field,
                                            engine)
  assert rounds > 1 and every < iterate, 'the test must exchange'
  assert messages[0] == rounds - 1 and messages[3] == 2 * (rounds - 1)
  want = numpy_oracle.run(stencil, {stencil.input_names[0]: field})[
      stencil.output_names[0]]
  idx, _, _ = _box(stencil, extent)
  assert np.array_equal(got[idx], want[idx])


# ---------------------------------------------------------------------------
# GPU: the configs as written
# ---------------------------------------------------------------------------

rows = []      # rows the launches of every interval covered (all ranks)


def _gpu_engine(stencil, fuse):
  import torch
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower

  def engine(slab):
    prog = runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                           extent=slab.local_extent)
    stream = torch.cuda.current_stream().cuda_stream

    def step(dst, src, lext, iters):
      prog.run_device([t.data_ptr() for t in dst],
                      [t.data_ptr() for t in src], lext, iterate=iters,
                      stream=stream, origin=slab.origin,
                      global_extent=slab.extent, keep=slab.keep)
      rows.append(prog.last_rows())

    def to_host(t):
      torch.cuda.synchronize()
      return t.cpu().numpy()

    return step, lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda(), \
        to_host

  return engine


C5 = ('jacobi2d.soda', (8192, 8192), 1000)
C4 = ('heat3d.soda', (512, 512, 512), 50)


@pytest.mark.gpu
@pytest.mark.parametrize('fuse', ['bench', (12, 4), (4,)])
def test_c5_jacobi2d_8192_iterate_1000(built, fuse):
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  name, extent, iterate = C5
  if fuse == 'bench':        # the depths bench.py / sodac default to (T13 ...)
    fuse = lower.DEFAULT_FUSE
  stencil = core.from_file(soda_path(name), iterate=iterate)
  field = _field(extent, 3)
  with runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                       extent=extent) as prog:
    got = prog.run({'t1': field})['t0']
    launches, fused = prog.last_launches()
  assert launches <= (250 if fuse == (4,) else 90)
  idx, lo, hi = _box(stencil, extent)
  assert (lo, hi) == ((1000, 1000), (7192, 7192))
  want = _oracle(name, extent, iterate, 3, 'random')
  assert np.array_equal(got[idx], want[idx])


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['random', 'ramp'])
def test_c4_heat3d_512_iterate_50(built, kind):
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  name, extent, iterate = C4
  stencil = core.from_file(soda_path(name), iterate=iterate)
  field = _field(extent, 2, kind)
  with runtime.Program(stencil, lower.LowerOptions(fuse=(2,)),
                       extent=extent) as prog:
    got = prog.run({stencil.input_names[0]: field})[stencil.output_names[0]]
    assert prog.last_launches() == (25, 25)       # 25 launches of T = 2
  idx, lo, hi = _box(stencil, extent)
  assert (lo, hi) == ((50,) * 3, (462,) * 3)
  if kind == 'ramp':      # closed form: p + q + r is a fixed point, exactly
    assert np.array_equal(got[idx], field[idx])
  want = _oracle(name, extent, iterate, 2, kind)
  assert np.array_equal(got[idx], want[idx])


@pytest.mark.gpu
@pytest.mark.parametrize('config,fuse,seed,every_want,rounds_want', [
    (C5, (12, 4), 3, 120, 9),     # 120 ghost rows per side of a 1024-row slab
    (C4, (2,), 2, 8, 7),          # 8 ghost planes per side of a 64-plane slab
])
def test_eight_slabs_on_one_gpu(built, config, fuse, seed, every_want,
                                rounds_want):
  """The 8-GPU decomposition of C4 / C5 with its exchanges, eight ranks on the
  one GPU: equal to the single-GPU run and to the oracle, bit for bit."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  name, extent, iterate = config
  stencil = core.from_file(soda_path(name), iterate=iterate)
  field = _field(extent, seed)
  got, every, rounds, messages = _run_slabs(
      stencil, extent, 8, fuse, field, _gpu_engine(stencil, fuse))
  assert (every, rounds) == (every_want, rounds_want)
  assert messages == [rounds - 1] + [2 * (rounds - 1)] * 6 + [rounds - 1]
  idx, _, _ = _box(stencil, extent)
  want = _oracle(name, extent, iterate, seed, 'random')
  assert np.array_equal(got[idx], want[idx])
  with runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                       extent=extent) as prog:
    single = prog.run({stencil.input_names[0]: field})[stencil.output_names[0]]
  assert np.array_equal(got[idx], single[idx])

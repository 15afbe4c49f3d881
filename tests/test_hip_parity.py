"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Integer programs must match bit for bit; floating-point programs are compiled
with -ffp-contract=off from the same expression text as the oracle, so they
are ALSO required to match bit for bit on the valid box (stricter than the
reference harness's 1e-5 rule, frt/host.py:634-657, which is checked too).
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import soda_path

pytestmark = pytest.mark.gpu


def _inputs(stencil, extent, seed=0, kind='random'):
  shape = tuple(extent[::-1])
  rng = np.random.default_rng(seed)
  out = {}
  for name, t in zip(stencil.input_names, stencil.input_types):
    dt = np.dtype(t.np_name)
    if kind == 'ramp':  # the reference harness's integer init: p + q (+ r)
      out[name] = np.indices(shape).sum(axis=0).astype(dt)
    elif t.is_float:
      out[name] = rng.random(shape, dtype=np.float64).astype(dt)
    else:
      info = np.iinfo(dt)
      out[name] = rng.integers(info.min, int(info.max) + 1, size=shape,
                               dtype=np.int64).astype(dt)
  for p in stencil.param_stmts:       # param arrays: C order, small values
    dt = np.dtype(p.haoda_type.np_name)
    size = p.size or (1,)
    out[p.name] = (rng.random(size).astype(dt) if p.haoda_type.is_float else
                   rng.integers(-9, 10, size=size).astype(dt))
  return out


def _check(stencil, extent, opts, iterate=None, seed=0, kind='random',
           oracle='numpy'):
  from soda_amd import runtime
  from oracle import numpy_oracle, c_oracle
  inputs = _inputs(stencil, extent, seed, kind)
  iterate = stencil.iterate if iterate is None else iterate
  with runtime.Program(stencil, opts, extent=extent) as prog:
    got = prog.run(inputs, iterate=iterate)
    kernels = [k.name for k in prog.module.kernels]
  if oracle == 'numpy':
    want = numpy_oracle.run(stencil, inputs, iterate=iterate)
  else:
    want = c_oracle.COracle(stencil).run(inputs, iterate=iterate)
  for name in stencil.output_names:
    lo, hi = stencil.valid_box(extent, name, iterate)
    assert all(h > l for l, h in zip(lo, hi)), 'empty valid box: bad test'
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    g, w = got[name][idx], want[name][idx]
    assert numpy_oracle.compare(got[name], want[name], lo, hi) == 0
    same = g.view(np.uint8) == w.view(np.uint8) if g.dtype.kind == 'f' \
        else g == w
    assert same.all(), '%s: %d cells not bit-identical (kernels %s)' % (
        name, (~same).sum(), kernels)
    # outside the box the caller's array is untouched (zeros here)
    mask = np.ones(got[name].shape, bool)
    mask[idx] = False
    assert not got[name][mask].any()


def test_dpp_wave_shift_direction(built):
  """soda_lane_dn/up move data the way codegen/hip/march.py assumes."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  src = lower.runtime_text() + '''
extern "C" __global__ void probe(soda_hip_kargs_t a) {
  int* out = (int*)a.buf[1];
  const int lane = threadIdx.x;
  out[lane] = soda_lane_dn(lane + 100);
  out[64 + lane] = soda_lane_up(lane + 100);
  out[128 + lane] = (int)soda_lane_dn((uint16_t)(lane + 7));
  out[192 + lane] = (int)(soda_lane_up((double)lane + 0.5) * 2.0);
}
'''
  code = runtime.compile_source(src, 'probe.hip')
  plan = runtime.Plan()
  plan.abi_version = runtime.ABI_VERSION
  plan.dim = 1
  plan.num_inputs = plan.num_outputs = 1
  plan.elem_size[0] = plan.elem_size[1] = 4
  plan.num_kernels = 1
  plan.kernels[0].name = b'probe'
  plan.kernels[0].block[0] = 64
  plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
  plan.kernels[0].tile[0] = 256
  plan.num_passes = 1
  plan.passes[0].fused_iters = 1
  plan.passes[0].num_kernels = 1
  lib = runtime.library()
  handle = ctypes.c_void_p()
  runtime.check(lib.soda_hip_program_create(code, len(code), ctypes.byref(plan),
                                            0, ctypes.byref(handle)), 'create')
  a = np.zeros(256, np.int32)
  out = np.full(256, -1, np.int32)
  ext = (ctypes.c_int32 * 1)(256)
  strd = (ctypes.c_int32 * 1)(1)
  tin = (runtime.HostTensor * 1)(runtime.HostTensor(a.ctypes.data, ext, strd,
                                                    None))
  tout = (runtime.HostTensor * 1)(runtime.HostTensor(out.ctypes.data, ext,
                                                     strd, None))
  runtime.check(lib.soda_hip_run_host(handle, tin, tout, 1), 'run')
  lib.soda_hip_program_destroy(handle)
  lane = np.arange(64)
  assert (out[:64] == np.where(lane > 0, lane + 99, 0)).all()
  assert (out[64:128] == np.where(lane < 63, lane + 101, 0)).all()
  assert (out[128:192] == np.where(lane > 0, lane + 6, 0)).all()
  assert (out[192:] == np.where(lane < 63, 2 * lane + 3, 0)).all()


def _probe(src, name, n_out):
  """Runs a one-block probe kernel `name(kargs)` that fills n_out ints."""
  from soda_amd import runtime
  code = runtime.compile_source(src, name + '.hip')
  plan = runtime.Plan()
  plan.abi_version = runtime.ABI_VERSION
  plan.dim = 1
  plan.num_inputs = plan.num_outputs = 1
  plan.elem_size[0] = plan.elem_size[1] = 4
  plan.num_kernels = 1
  plan.kernels[0].name = name.encode()
  plan.kernels[0].block[0] = 64
  plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
  plan.kernels[0].tile[0] = n_out
  plan.num_passes = 1
  plan.passes[0].fused_iters = 1
  plan.passes[0].num_kernels = 1
  lib = runtime.library()
  handle = ctypes.c_void_p()
  runtime.check(lib.soda_hip_program_create(code, len(code), ctypes.byref(plan),
                                            0, ctypes.byref(handle)), 'create')
  a = np.zeros(n_out, np.int32)
  out = np.full(n_out, -1, np.int32)
  ext = (ctypes.c_int32 * 1)(n_out)
  strd = (ctypes.c_int32 * 1)(1)
  tin = (runtime.HostTensor * 1)(runtime.HostTensor(a.ctypes.data, ext, strd,
                                                    None))
  tout = (runtime.HostTensor * 1)(runtime.HostTensor(out.ctypes.data, ext,
                                                     strd, None))
  runtime.check(lib.soda_hip_run_host(handle, tin, tout, 1), 'run')
  lib.soda_hip_program_destroy(handle)
  return out


def test_swizzle_lane_shift_direction(built):
  """The ds_swizzle rotate forms of soda_rt.h move data the way
  codegen/hip/march.py assumes: *32 rotate inside each 32-lane half, *64 shift
  the whole wave (lane 0 of `dn` / lane 63 of `up` hold a wrapped value: halo
  lanes)."""
  from soda_amd.codegen.hip import lower
  src = lower.runtime_text() + '''
extern "C" __global__ void swzprobe(soda_hip_kargs_t a) {
  int* out = (int*)a.buf[1];
  const int lane = threadIdx.x;
  out[lane] = soda_lane_dn32(lane + 100);
  out[64 + lane] = soda_lane_up32(lane + 100);
  out[128 + lane] = soda_lane_dn64(lane + 100);
  out[192 + lane] = soda_lane_up64(lane + 100);
  out[256 + lane] = (int)soda_lane_dn64((uint16_t)(lane + 7));
  out[320 + lane] = (int)(soda_lane_up64((double)lane + 0.5) * 2.0);
  out[384 + lane] = (int)soda_lane_dn32((float)lane);
}
'''
  out = _probe(src, 'swzprobe', 448)
  lane = np.arange(64)
  half = lane & 32
  msg = 'dn32 %s\nup32 %s' % (out[:64].tolist(), out[64:128].tolist())
  assert (out[:64] == 100 + (half | ((lane - 1) & 31))).all(), msg
  assert (out[64:128] == 100 + (half | ((lane + 1) & 31))).all(), msg
  assert (out[129:192] == lane[1:] + 99).all(), out[128:192].tolist()
  assert (out[192:255] == lane[:63] + 101).all(), out[192:256].tolist()
  assert (out[257:320] == lane[1:] + 6).all()
  assert (out[320:383] == 2 * lane[:63] + 3).all()
  assert (out[384:448] == (half | ((lane - 1) & 31))).all()


CORPUS_2D = ['jacobi2d.soda', 'blur.soda', 'seidel2d.soda', 'sobel2d.soda',
             'denoise2d.soda', 'skew2d.soda', 'erosion.soda', 'xcorr.soda',
             'contrast.soda']


@pytest.mark.parametrize('name', CORPUS_2D)
@pytest.mark.parametrize('strategy', ['direct', 'auto'])
def test_corpus_2d(built, name, strategy):
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name))
  # ragged on purpose: not a multiple of the strip width or the chunk height
  extent = (600, 150)
  _check(stencil, extent, lower.LowerOptions(strategy=strategy, fuse=(2,)))


@pytest.mark.parametrize('name', ['heat3d.soda', 'jacobi3d.soda',
                                  'denoise3d.soda'])
def test_corpus_3d(built, name):
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name))
  _check(stencil, (40, 36, 28), lower.LowerOptions())


@pytest.mark.parametrize('iterate,fuse', [(1, (4,)), (3, (4,)), (4, (4,)),
                                          (7, (4, 2)), (9, (8,)), (13, (3,))])
def test_jacobi2d_temporal_blocking(built, iterate, fuse):
  """Fused passes + remainder passes give the oracle's result."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=iterate)
  _check(stencil, (1000, 300), lower.LowerOptions(fuse=fuse), oracle='c')


@pytest.mark.parametrize('vec_extent', [(1000, 64), (1002, 64), (1001, 64)])
def test_row_length_picks_vector_width(built, vec_extent):
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=3)
  _check(stencil, vec_extent, lower.LowerOptions(fuse=(2,)))


@pytest.mark.parametrize('name,extent,iterate', [
    ('jacobi2d.soda', (8, 8), 1), ('jacobi2d.soda', (5, 7), 2),
    ('jacobi2d.soda', (260, 9), 3), ('jacobi2d.soda', (256, 3), 1),
    ('jacobi2d.soda', (4, 300), 1), ('blur.soda', (16, 5), 1),
    ('blur.soda', (3, 3), 1), ('skew2d.soda', (12, 6), 1),
    ('heat3d.soda', (8, 8, 8), 2), ('heat3d.soda', (260, 5, 4), 1),
    ('heat3d.soda', (516, 7, 70), 1), ('jacobi3d.soda', (12, 3, 9), 1),
])
def test_small_and_odd_extents(built, name, extent, iterate):
  """Grids smaller than one strip / chunk / tile, and not multiples of them."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  lo, hi = stencil.valid_box(extent)
  if not all(h > l for l, h in zip(lo, hi)):
    pytest.skip('valid box is empty')
  _check(stencil, extent, lower.LowerOptions(fuse=(2,)))


def test_one_dimensional_program(built):
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_text(
      'kernel: smooth1d\nburst width: 64\nunroll factor: 2\niterate: 3\n'
      'input float: a\n'
      'output float: b(0) = (a(-1) + a(0) * 2.0f + a(1)) * 0.25f')
  _check(stencil, (1000,), lower.LowerOptions())


def test_explicit_chunk_and_wave_shapes(built):
  """Non-default launch geometry gives the same bits."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=5)
  for kw in (dict(chunk_rows=7, waves_x=2, waves_y=2),
             dict(chunk_rows=300, prefetch=1),
             dict(chunk_rows=16, prefetch=4, nt_store=True, nt_load=False,
                  xcd_swizzle=False, edge_loads=False),
             dict(warm_guards=True), dict(interleave=True),
             dict(lane_shift='bperm'), dict(lane_shift='lds'), dict(vec=2),
             dict(lane_shift='swz'), dict(lane_shift='swzh'),
             dict(lane_shift='mixh'), dict(lane_shift='mixh', pipe=3),
             dict(lane_shift='mix64'), dict(lane_shift='mix64d'),
             dict(vec=1, chunk_rows=33)):
    _check(stencil, (1000, 200), lower.LowerOptions(fuse=(3,), **kw))
  h = core.from_file(soda_path('heat3d.soda'), iterate=2)
  for kw in (dict(tile_rows=1), dict(tile_rows=6, chunk_rows=5),
             dict(edge_loads=False, prefetch=2)):
    _check(h, (300, 20, 24), lower.LowerOptions(**kw))


@pytest.mark.gpu
@pytest.mark.parametrize('name,iterate,fuse,extent', [
    ('jacobi2d.soda', 24, (12,), (1000, 333)),
    ('jacobi2d.soda', 9, (4,), (520, 97)),
    ('seidel2d.soda', 8, (8,), (640, 200)),
    ('blur.soda', 4, (2,), (640, 200)),          # a tap that reaches two cells
    ('coupled2d.soda', 4, (2,), (300, 90)),      # two tensors per iteration
])
def test_lane_neighbours_through_lds(built, name, iterate, fuse, extent):
  """`lane_shift='lds'`: a lane files the end cells of every computed row in a
  wave-private LDS line and reads its neighbours' one row step later instead
  of shifting registers with DPP.  Bit-identical to the oracle."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  mod = lower.lower(stencil, lower.LowerOptions(fuse=fuse, lane_shift='lds',
                                                vec=4 if name != 'blur.soda' else 8))
  assert any('_ldsx' in k.name for k in mod.kernels)
  _check(stencil, extent, lower.LowerOptions(fuse=fuse, lane_shift='lds'),
         oracle='c')


@pytest.mark.parametrize('name,iterate,fuse', [
    ('blur.soda', 3, (2,)), ('blur.soda', 4, (4,)), ('seidel2d.soda', 7, (3,)),
    ('jacobi3d.soda', 3, ()), ('heat3d.soda', 5, ()),
    ('jacobi3d.soda', 3, (2,)), ('heat3d.soda', 5, (2,)),   # 3-D, T=2 fused
    ('heat3d.soda', 4, (4,)),                               # capped at T=2
])
def test_iterated_multi_stage_and_3d(built, name, iterate, fuse):
  """Temporal blocking of a two-stage program (4 stages fused for blur x 2)
  and iterated 3-D programs."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  extent = (640, 200) if stencil.dim == 2 else (300, 24, 40)
  _check(stencil, extent, lower.LowerOptions(fuse=fuse), oracle='c')


@pytest.mark.parametrize('iterate,extent', [(2, (40, 12, 10, 9)),
                                            (3, (68, 9, 11, 12)),
                                            (1, (4, 5, 3, 6))])
def test_four_dimensional_program(built, iterate, extent):
  """The reference's coordinate sets are four deep (ref src/soda/util.py:4-6)
  and SODA_HIP_MAX_DIM is 4: a 4-D program (tests/golden/heat4d.soda, a
  9-point star whose coefficients are powers of two that sum to 1) on the
  `direct` kernels, against both oracles and the closed form -- the field
  p + q + r + s is a fixed point bit for bit on the valid box."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import numpy_oracle
  stencil = core.from_file(soda_path('heat4d.soda'), iterate=iterate)
  assert stencil.dim == 4
  _check(stencil, extent, lower.LowerOptions(), oracle='c')
  ramp = np.indices(extent[::-1]).sum(axis=0).astype(np.float32)
  with runtime.Program(stencil, lower.LowerOptions(), extent=extent) as prog:
    assert all('direct' in k.name for k in prog.module.kernels)
    got = prog.run({'in': ramp})['out']
  lo, hi = stencil.valid_box(extent)
  assert lo == (iterate,) * 4
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  assert np.array_equal(got[idx], ramp[idx])
  want = numpy_oracle.run(stencil, {'in': ramp})['out']
  assert np.array_equal(got[idx], want[idx])


@pytest.mark.parametrize('fuse,strategy,extent', [
    ((3,), 'auto', (600, 150)), ((), 'auto', (1000, 333)),
    ((2,), 'auto', (260, 70)), ((), 'direct', (600, 150))])
def test_integer_window_forms(built, fuse, strategy, extent):
  """tests/golden/winsum2d.soda -- a 7-row and an 8-cell int16 sum, a 6-row min
  and a 6-cell max, iterate 3 -- through every window form of the backend: the
  sliding sum along the streamed dimension (an int32 accumulator that lives
  across row steps, set up at the stage's first step of a chunk), the
  power-of-two chain (min along it; both directions on `direct`), the
  dimension-0 windows reduced for all cells of a lane jointly; fused and not,
  chunks that start in the peeled warm-up and in the loop.  Bit for bit."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('winsum2d.soda'))
  opts = lower.LowerOptions(fuse=fuse, strategy=strategy)
  import os
  if strategy == 'auto' and not (os.environ.get('SODA_HIP_WINDOWS') or
                                 os.environ.get('SODA_HIP_SLIDE')):
    src = lower.lower(stencil, opts).source       # (the defaults, not an A/B run)
    assert 'xa_t0_rows_r0' in src and 'xw_t0_cols' in src and 'xm_t0_hi' in src
    assert 'in_min1_4' in src
  _check(stencil, extent, opts, oracle='c')
  for chunk in (5, 40):       # first step in the loop / deep in the warm-up
    if strategy == 'auto':
      _check(stencil, extent, lower.LowerOptions(fuse=fuse, chunk_rows=chunk,
                                                 peel=0 if chunk == 5 else -1),
             oracle='c', seed=chunk)


@pytest.mark.parametrize('name,extent,fuse', [
    ('coupled2d.soda', (300, 90), (2,)),     # 2 inputs -> 2 outputs, iterate 3
    ('coupled2d.soda', (300, 90), ()),
    ('lets2d.soda', (260, 70), ()),          # let variables, double, sqrt, max
    ('ints2d.soda', (520, 66), ()),          # uint8/int32/int64, select/abs/%/^/&
])
@pytest.mark.parametrize('strategy', ['auto', 'direct'])
def test_language_surface(built, name, extent, fuse, strategy):
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name))
  _check(stencil, extent, lower.LowerOptions(strategy=strategy, fuse=fuse),
         oracle='c')


@pytest.mark.parametrize('name,iterate', [('jacobi2d.soda', 3),
                                          ('seidel2d.soda', 2)])
def test_lds_halo_tile_variant(built, name, iterate):
  """The classic LDS-tile kernel (the measured alternative to march2d)."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  _check(stencil, (1000, 150), lower.LowerOptions(strategy='lds'), oracle='c')
  _check(stencil, (260, 33), lower.LowerOptions(strategy='lds'))


def test_in_place_is_rejected(built):
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  import torch
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=2)
  a = torch.rand((64, 256), device='cuda')
  with runtime.Program(stencil, lower.LowerOptions(), extent=(256, 64)) as prog:
    with pytest.raises(util.BackendError, match='alias'):
      prog.run_device([a.data_ptr()], [a.data_ptr()], (256, 64))


def test_sodac_hip_backend_runs(built):
  """`sodac file.soda --hip-backend`: parse, JIT, run on the GPU, JSON out."""
  import json
  import subprocess
  import sys
  from conftest import ROOT
  r = subprocess.run([sys.executable, '-m', 'soda_amd.sodac',
                      soda_path('blur.soda'), '--hip-backend', '--hip-extent',
                      '2000', '64'], cwd=ROOT, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  out = json.loads(r.stdout.strip().splitlines()[-1])
  assert out['kernel'] == 'blur' and out['extent'] == [2000, 64]
  # ramp input p+q -> blur_y = p+q+2 on [0,1998) x [0,62), zero elsewhere
  q, p = np.indices((64, 2000))
  assert out['checksum']['blur_y'] == float((p + q + 2)[:62, :1998].sum())


def test_sodac_hip_backend_on_several_gpus(built):
  """`sodac --hip-backend --hip-gpus N --hip-virtual`: the same program cut
  into N slabs (soda_hip_group_*, here all on the one GPU), same checksum as on
  one GPU -- blur (one shot) and heat3d (iterated: halo exchanges)."""
  import json
  import subprocess
  import sys
  from conftest import ROOT

  def run(*flags):
    r = subprocess.run([sys.executable, '-m', 'soda_amd.sodac', *flags],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])

  blur = [soda_path('blur.soda'), '--hip-backend', '--hip-extent', '2000', '96']
  one = run(*blur)
  four = run(*blur, '--hip-gpus', '4', '--hip-virtual')
  assert four['gpus'] == 4 and four['devices'] == [0] * 4
  assert four['checksum'] == one['checksum']
  heat = [soda_path('heat3d.soda'), '--iterate', '12', '--hip-fuse', '2',
          '--hip-backend', '--hip-extent', '64', '48', '96']
  one = run(*heat)
  three = run(*heat, '--hip-gpus', '3', '--hip-virtual',
              '--hip-exchange-every', '4')
  assert (three['exchange_every'], three['exchanges']) == (4, 2)
  assert three['split_passes'] > 0 and three['checksum'] == one['checksum']
  assert any(v != 0 for v in one['checksum'].values())


def test_blur_reference_init_closed_form(built):
  """blur on the reference harness's p+q input is p+q+2 (SURVEY 8c KAT)."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('blur.soda'))
  extent = (2000, 1024)   # BASELINE config 1's grid
  inputs = _inputs(stencil, extent, kind='ramp')
  with runtime.Program(stencil, lower.LowerOptions(), extent=extent) as prog:
    got = prog.run(inputs)['blur_y']
  q, p = np.indices(extent[::-1])
  assert (got[:1022, :1998] == (p + q + 2)[:1022, :1998]).all()


def test_heat3d_ramp_is_fixed_point(built):
  """heat3d leaves p+q+r unchanged, bit for bit, for any iterate."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('heat3d.soda'), iterate=5)
  extent = (48, 40, 32)
  inputs = _inputs(stencil, extent, kind='ramp')
  with runtime.Program(stencil, lower.LowerOptions(), extent=extent) as prog:
    got = prog.run(inputs)['out']
  lo, hi = stencil.valid_box(extent)
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  assert (got[idx] == inputs['in'][idx]).all()


def test_strided_host_arrays(built):
  """(ptr, extent, stride, min) tensors with non-dense strides."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import numpy_oracle
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=2)
  extent = (256, 96)
  big = np.random.default_rng(3).random((96, 300), dtype=np.float32)
  a = big[:, 20:276]           # row pitch 300, not 256
  out_big = np.zeros((96, 280), np.float32)
  out = out_big[:, 8:264]
  with runtime.Program(stencil, lower.LowerOptions(fuse=(2,)),
                       extent=extent) as prog:
    prog.run({'t1': a}, outputs={'t0': out})
  want = numpy_oracle.run(stencil, {'t1': np.ascontiguousarray(a)})['t0']
  assert np.array_equal(out, want)
  assert not out_big[:, :8].any() and not out_big[:, 264:].any()


@pytest.mark.gpu
@pytest.mark.parametrize('kb', [8, 48])
def test_host_entry_in_bands(built, monkeypatch, kb):
  """soda_hip_run_host_box cut into bands along the last dimension (round 5,
  soda_host.cpp run_banded): copy-in, kernels and copy-out of neighbouring
  bands overlap on three streams, every band a window run with iterate x reach
  ghost rows.  Forced onto small grids with staging chunks of a few KiB
  (SODA_HIP_HOST_CHUNK_KB): the result must be the oracle's bit for bit on the
  valid box and leave the rest of the caller's array alone -- 2-D and 3-D,
  fused iterations, two coupled outputs, a preserved border, strided arrays
  -- and equal the unbanded run (SODA_HIP_HOST_BANDS=0)."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import c_oracle
  monkeypatch.setenv('SODA_HIP_HOST_CHUNK_KB', str(kb))
  cases = [
      ('jacobi2d.soda', (512, 700), dict(iterate=7), (4,)),
      ('blur.soda', (768, 400), {}, ()),
      ('heat3d.soda', (64, 48, 160), dict(iterate=3), (2,)),
      ('denoise2d.soda', (256, 600), {}, ()),
      ('jacobi2d.soda', (256, 900), dict(iterate=5, border='preserve'), (4,)),
      ('sobel2d.soda', (512, 333), {}, ()),
      ('coupled2d.soda', (256, 500), dict(iterate=4), (2,)),   # 2 in, 2 out
      ('conv2d.soda', (256, 420), {}, ()),                     # a param array
      ('smooth1d', (40000,), {}, ()),                          # 1-D: no bands
  ]
  rng = np.random.default_rng(11)
  for name, extent, kw, fuse in cases:
    if name == 'smooth1d':
      st = core.from_text(
          'kernel: smooth1d\nburst width: 64\nunroll factor: 2\niterate: 3\n'
          'input float: a\n'
          'output float: b(0) = (a(-1) + a(0) * 2.0f + a(1)) * 0.25f')
    else:
      st = core.from_file(soda_path(name), **kw)
    shape = extent[::-1]
    ins = {}
    for n, t in zip(st.input_names, st.input_types):
      dt = np.dtype(t.np_name)
      ins[n] = (rng.random(shape).astype(dt) if dt.kind == 'f'
                else rng.integers(0, 2000, shape).astype(dt))
    for pstmt in st.param_stmts:           # param arrays: C order, as written
      ins[pstmt.name] = rng.random(pstmt.size or (1,)).astype(
          np.dtype(pstmt.haoda_type.np_name))
    want = c_oracle.COracle(st).run(ins)
    runs = {}
    for bands in ('1', '0', 'estimate'):
      if bands == 'estimate':      # the library's own choice (choose_bands)
        monkeypatch.delenv('SODA_HIP_HOST_BANDS')
      else:
        monkeypatch.setenv('SODA_HIP_HOST_BANDS', bands)
      with runtime.Program(st, lower.LowerOptions(fuse=fuse),
                           extent=extent) as prog:
        # strided caller arrays: rows with a tail
        big_in = {n: np.zeros(shape[:-1] + (shape[-1] + 24,), ins[n].dtype)
                  for n in st.input_names}
        for n in st.input_names:
          big_in[n][..., 8:8 + shape[-1]] = ins[n]
        outs_big = {n: np.full(shape[:-1] + (shape[-1] + 10,), 77,
                               np.dtype(t.np_name))
                    for n, t in zip(st.output_names, st.output_types)}
        outs = {n: a[..., 3:3 + shape[-1]] for n, a in outs_big.items()}
        given = {n: a[..., 8:8 + shape[-1]] for n, a in big_in.items()}
        for pstmt in st.param_stmts:
          given[pstmt.name] = ins[pstmt.name]
        prog.run(given, outputs=outs)
        runs[bands] = {n: a.copy() for n, a in outs_big.items()}
    for o in st.output_names:
      assert np.array_equal(runs['1'][o].view(np.uint8),
                            runs['0'][o].view(np.uint8)), (name, o)
      assert np.array_equal(runs['estimate'][o].view(np.uint8),
                            runs['0'][o].view(np.uint8)), (name, o)
      got = runs['1'][o][..., 3:3 + shape[-1]]
      if st.preserve_border:
        idx = tuple(slice(None) for _ in shape)
      else:
        lo, hi = st.valid_box(extent, o)
        idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      assert np.array_equal(got[idx].view(np.uint8),
                            want[o][idx].view(np.uint8)), (name, o, kb)
      mask = np.ones(runs['1'][o].shape, bool)
      mask[..., 3:3 + shape[-1]][idx] = False
      assert (runs['1'][o][mask] == 77).all(), (name, o, 'outside the box')


def test_device_entry_and_errors(built):
  from soda_amd import core, runtime, util
  from soda_amd.codegen.hip import lower
  import torch
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=5)
  extent = (512, 128)
  a = torch.rand(extent[::-1], device='cuda', dtype=torch.float32)
  b = torch.empty_like(a)
  keep = a.clone()
  with runtime.Program(stencil, lower.LowerOptions(fuse=(4,)),
                       extent=extent) as prog:
    stream = torch.cuda.current_stream().cuda_stream
    prog.run_device([b.data_ptr()], [a.data_ptr()], extent, stream=stream)
    torch.cuda.synchronize()
    assert prog.last_launches() == (2, 1)   # one T=4 pass + one T=1 pass
    assert torch.equal(a, keep), 'inputs must never be written'
    host = prog.run({'t1': a.cpu().numpy()})['t0']
    lo, hi = stencil.valid_box(extent)
    assert np.array_equal(b.cpu().numpy()[lo[1]:hi[1], lo[0]:hi[0]],
                          host[lo[1]:hi[1], lo[0]:hi[0]])
    with pytest.raises(util.BackendError):
      prog.run_device([b.data_ptr()], [a.data_ptr()], extent, iterate=0)
    with pytest.raises(util.BackendError):
      prog.run_device([b.data_ptr()], [0], extent)


@pytest.mark.parametrize('name,extent,iterate,fuse', [
    ('jacobi2d.soda', (8192, 8192), 12, (4,)),
    ('jacobi2d.soda', (8192, 8192), 100, (12, 4)),     # BASELINE config 1/2
    ('jacobi2d.soda', (8192, 8192), 100, 'bench'),     # ... as bench.py runs it
    ('blur.soda', (16384, 16384), 1, ()),              # BASELINE config 3
    ('heat3d.soda', (512, 512, 512), 6, ()),           # BASELINE config 4 grid
])
def test_full_size_properties(built, name, extent, iterate, fuse):
  """BASELINE-sized grids: compare against the multi-threaded C oracle on the
  whole valid box (it finishes in seconds), plus size-independent properties:
  fused and unfused schedules agree bit for bit; a constant field is a fixed
  point of jacobi2d."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import c_oracle
  stencil = core.from_file(soda_path(name), iterate=iterate)
  inputs = _inputs(stencil, extent, seed=11)
  if fuse == 'bench':       # the depth set bench.py / sodac default to
    fuse = lower.DEFAULT_FUSE
    assert max(fuse) == 13
  with runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                       extent=extent) as prog:
    got = prog.run(inputs)
  want = c_oracle.COracle(stencil).run(inputs)
  for o in stencil.output_names:
    lo, hi = stencil.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(got[o][idx], want[o][idx])
  if fuse:
    with runtime.Program(stencil, lower.LowerOptions(fuse=()),
                         extent=extent) as prog:
      unfused = prog.run(inputs)
    for o in stencil.output_names:
      assert np.array_equal(got[o], unfused[o])


def test_every_kernel_of_the_benched_schedule_matches_the_oracle(built):
  """What bench.py times is what is checked: the program it builds by default
  (jacobi2d, lower.DEFAULT_FUSE, 8192 x 8192), and for EVERY pass the
  library schedules for the 100-iteration step on that extent a run that
  consists of launches of that pass alone (as bench.py's own per-kernel timing
  does), on the full grid, against the OpenMP C oracle bit for bit.  The
  kernel names are the ones the bench line reports (`roofline.kernel`,
  `roofline.scheduled_kernels`).  The reference checks every run it times:
  frt/host.py:545-553 (run), :625-657 (compare)."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import c_oracle
  import conftest
  extent, iterate = (8192, 8192), 100
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=iterate)
  inputs = _inputs(stencil, extent, seed=29)
  lo_hi = stencil.valid_box
  with runtime.Program(stencil, lower.LowerOptions(fuse=lower.DEFAULT_FUSE),
                       extent=extent, calibrate=True) as prog:   # as bench.py
    sched = prog.schedule(extent, iterate)
    assert sum(t * n for t, n in sched.items()) == iterate
    by_depth = {p.fused_iters: prog.module.kernels[p.kernels[0]].name
                for p in prog.module.sorted_passes()}
    # the clock picks the mix (BENCH_r03: 4 x T13 + 4 x T12; the model alone:
    # 7 x T12 + 2 x T8): whatever it picked here, plus the two deepest passes
    # the bench line has named so far
    depths = set(sched) | {13, 12}
    checked = {}
    for depth in sorted(depths, reverse=True):
      n = 2
      while n > 1 and prog.schedule(extent, depth * n) != {depth: n}:
        n -= 1
      assert prog.schedule(extent, depth * n) == {depth: n}
      got = prog.run(inputs, iterate=depth * n)['t0']
      assert prog.last_launches()[0] == n
      want = c_oracle.COracle(stencil).run(inputs, iterate=depth * n)['t0']
      lo, hi = lo_hi(extent, 't0', depth * n)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      assert np.array_equal(got[idx], want[idx]), by_depth[depth]
      checked[by_depth[depth]] = depth * n
      conftest.VERIFIED_KERNELS.add(by_depth[depth])
    # and the step itself, as scheduled
    got = prog.run(inputs)['t0']
    want = c_oracle.COracle(stencil).run(inputs)['t0']
    lo, hi = lo_hi(extent)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(got[idx], want[idx])
  assert set(checked) >= {by_depth[d] for d in sched}
  assert all('_T%d_' % d in by_depth[d] for d in depths)


# (file, input decl override, extent, output decl override, mode): `dense` =
# the stream is run as the original n-D program with the marching kernels
# (tile row block a multiple of the burst), `linear` = as the 1-D program,
# `fallback` = dense requested but the tile is not burst-aligned
STREAM_CASES = [
    ('blur.soda', None, (2000, 20), None, 'dense'),
    ('blur.soda', None, (2000, 20), None, 'linear'),
    ('jacobi2d.soda', 'input dram 0 float: t1(33, *)', (33, 12), None,
     'fallback'),                                     # 33 % (64 / 32) != 0
    ('jacobi2d.soda', 'input dram 0 float: t1(33, *)', (80, 11), None,
     'fallback'),                                     # ... and padded tiles
    ('blur.soda', 'input dram 0 uint16: input(2048, *)', (2048, 20), None,
     'dense'),
    ('blur.soda', 'input dram 0 uint16: input(2048, *)', (5000, 37), None,
     'dense'),                                        # three tiles, ragged last
    ('jacobi2d.soda', None, (32, 12), None, 'dense'),  # iterate 2, as shipped
    ('jacobi2d.soda', None, (32, 12), None, 'linear'),
    ('jacobi2d.soda', None, (20, 9), None, 'dense'),   # narrower than the tile
    ('jacobi2d.soda', None, (100, 9), None, 'dense'),  # four overlapping tiles
    ('heat3d.soda', None, (32, 32, 9), None, 'dense'),  # iterate 2, 3-D tiles
    ('heat3d.soda', None, (32, 32, 9), None, 'linear'),
    ('heat3d.soda', None, (70, 40, 7), None, 'dense'),  # 3 x 2 tiles
    ('sobel2d.soda', None, (32, 8), None, 'dense'),    # three stages, int16
    ('jacobi2d.soda', 'input dram 0.1 float: t1(32, *)', (32, 12),
     'output dram 2.3 float:', 'dense'),              # two banks each side
    ('jacobi2d.soda', 'input dram 0.1 float: t1(32, *)', (32, 13),
     'output dram 2.3 float:', 'linear'),
    ('blur.soda', 'input dram 0.1.2.3 uint16: input(2048, *)', (2048, 20),
     'output dram 0.1.2.3 uint16:', 'dense'),         # four banks, 8 cells per 16 B
    ('blur.soda', 'input dram 0.1.2.3 uint16: input(2000, *)', (2000, 20),
     'output dram 0.1.2.3 uint16:', 'fallback'),      # 2000 % 64 != 0
    ('blur.soda', 'input dram 0.1.2 uint16: input(1008, *)', (1008, 9),
     'output dram 1.2.3 uint16:', 'dense'),           # three
    ('heat3d.soda', 'input dram 0.1 float: in(32, 32, *)', (32, 32, 9),
     'output dram 0.1 float:', 'dense'),
    # several inputs: the host delays each by its produce offset (f by two
    # rows behind u in denoise2d); two outputs in coupled2d
    ('denoise2d.soda', None, (32, 14), None, 'dense'),
    ('denoise2d.soda', None, (32, 14), None, 'linear'),
    ('coupled2d.soda', None, (32, 11), None, 'dense'),
]


@pytest.mark.parametrize('name,in_decl,extent,out_decl,mode', STREAM_CASES)
def test_wire_format_kernel_abi(built, name, in_decl, extent, out_decl, mode):
  """<app>_kernel(out banks, in banks, coalesced_data_num) on the
  reference's tiled / burst-aligned / bank-interleaved streams, driven by a
  restatement of the reference host's scatter and gather."""
  import re
  from soda_amd import core, stream
  from oracle import frt_layout, numpy_oracle
  text = open(soda_path(name)).read()
  if in_decl:
    text = re.sub(r'input dram [^\n]*', in_decl, text)
  if out_decl:
    text = re.sub(r'output dram [\d.]+ \w+:', out_decl, text)
  stencil = core.from_text(text)
  inputs = _inputs(stencil, extent, seed=5)
  layout = stream.WireLayout(stencil, extent)
  in_banks = frt_layout.scatter(layout, inputs)
  out_banks = frt_layout.alloc(layout, stencil.output_names)
  prog = stream.StreamProgram(stencil, dense=mode != 'linear')
  try:
    prog.run_banked_host(out_banks, in_banks, layout.cycle_count)
    assert prog.last_mode == ('dense' if mode == 'dense' else 'linear')
  finally:
    prog.close()
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(stencil.output_names, stencil.output_types)}
  frt_layout.gather(layout, out_banks, got)
  # the kernel's contract: every cell the host reads back (any tile count).
  # The host gathers EVERY output over the region of the program's window
  # (frt/host.py:357-375); with several outputs some of those cells lie
  # outside an output's own valid box, where the dense n-D form and the causal
  # 1-D form may differ (both unspecified): compare on the common box there.
  ref = {o: np.zeros_like(got[o]) for o in got}
  frt_layout.gather(layout, frt_layout.kernel_on_streams(layout, in_banks), ref)
  boxes = [stencil.valid_box(extent, o) for o in stencil.output_names]
  lo = [max(b[0][d] for b in boxes) for d in range(stencil.dim)]
  hi = [min(b[1][d] for b in boxes) for d in range(stencil.dim)]
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  for o in stencil.output_names:
    if len(stencil.output_names) == 1:
      assert np.array_equal(got[o], ref[o]), o
    else:
      assert np.array_equal(got[o][idx], ref[o][idx]), o
  want = numpy_oracle.run(stencil, inputs)
  assert all(h > l for l, h in zip(lo, hi))
  if layout.tiles > 1:
    # Known upstream defect: the reference host scatters
    # tile t from column t * (tile - kStencilDim) (frt/host.py:224-228) but
    # gathers it to column t * (tile - kStencilDim + 1) (:389-393), so from
    # the second tile on the caller gets results one column further off per
    # tile.  Where both maps agree -- tile index 0 in every tiled dimension --
    # the n-D oracle IS comparable: the cells tile 0 delivers.
    window = stencil.stencil_window_points(stencil.output_names[0])
    first = []
    for d in range(stencil.dim - 1):
      off = -min(p[d] for p in window)
      sdim = max(p[d] for p in window) + off + 1
      first.append((max(lo[d], off),
                    min(hi[d], stencil.tile_size[d] - max(0, sdim - 1 - off))))
    first.append((lo[-1], hi[-1]))
    assert all(b > a for a, b in first)
    idx0 = tuple(slice(a, b) for a, b in first[::-1])
    for o in stencil.output_names:
      assert got[o][idx0].size and np.array_equal(got[o][idx0], want[o][idx0]), o
    # ... and beyond it the defect is really there (or this comment is stale)
    o = stencil.output_names[0]
    assert not np.array_equal(got[o][idx], want[o][idx])
    return
  for o in stencil.output_names:
    assert np.array_equal(got[o][idx], want[o][idx]), o
    if len(stencil.output_names) == 1:
      assert np.array_equal(got[o], want[o]), o


@pytest.mark.gpu
@pytest.mark.parametrize('kb', [16, 4096])
def test_host_entry_on_pinned_arrays(built, monkeypatch, capfd, kb):
  """Dense caller arrays in pinned memory (runtime.PinnedBuffer: page-aligned
  memory under soda_hip_host_register, allocated once for the whole test, as a
  host would) go by DMA from / to where they are, no staging slot, no worker
  thread: whole rows for boxes that hold whole rows, one strided copy per chunk
  for a 2-D box that holds a column range or a 3-D box cut in dimension 0 / 1.
  Same bits as the pageable run, nothing outside the box touched, whole and
  in bands."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  monkeypatch.setenv('SODA_HIP_HOST_CHUNK_KB', str(kb))
  cases = [
      ('jacobi2d.soda', (512, 700), dict(iterate=7), (4,)),    # columns [7, 505)
      ('blur.soda', (768, 400), {}, ()),                       # columns [0, 766)
      ('heat3d.soda', (64, 48, 160), dict(iterate=3), (2,)),   # a 3-D box
      ('jacobi2d.soda', (256, 900), dict(iterate=5, border='preserve'), (4,)),
      ('coupled2d.soda', (256, 500), dict(iterate=4), (2,)),   # 2 in, 2 out
  ]
  rng = np.random.default_rng(12)
  room = 1 << 22            # per tensor, four tensors: the largest case fits
  with runtime.PinnedBuffer(4 * room) as pinned:
    for case in cases:
      _pinned_case(case, pinned, room, rng, monkeypatch, capfd, kb)


def _pinned_case(case, pinned, room, rng, monkeypatch, capfd, kb):
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  name, extent, kw, fuse = case
  st = core.from_file(soda_path(name), **kw)
  shape = extent[::-1]
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    dt = np.dtype(t.np_name)
    ins[n] = (rng.random(shape).astype(dt) if dt.kind == 'f'
              else rng.integers(0, 2000, shape).astype(dt))
  # the same values in the pinned buffer: tensor k at k * room
  pin_in = {}
  for k, n in enumerate(st.input_names):
    pin_in[n] = pinned.array(shape, ins[n].dtype, k * room)
    pin_in[n][...] = ins[n]
  runs = {}
  with runtime.Program(st, lower.LowerOptions(fuse=fuse),
                       extent=extent) as prog:
    for how in ('pageable', 'pinned', 'pinned in bands', 'pinned whole',
                'pinned, slots'):
      monkeypatch.delenv('SODA_HIP_HOST_BANDS', raising=False)
      monkeypatch.delenv('SODA_HIP_HOST_DIRECT', raising=False)
      if how == 'pinned in bands':
        monkeypatch.setenv('SODA_HIP_HOST_BANDS', '1')
      if how == 'pinned whole':
        monkeypatch.setenv('SODA_HIP_HOST_BANDS', '0')
      if how == 'pinned, slots':
        monkeypatch.setenv('SODA_HIP_HOST_DIRECT', '0')
      if how == 'pageable':
        outs = {n: np.full(shape, 77, np.dtype(t.np_name))
                for n, t in zip(st.output_names, st.output_types)}
      else:
        outs = {}
        for k, (n, t) in enumerate(zip(st.output_names, st.output_types)):
          outs[n] = pinned.array(shape, np.dtype(t.np_name),
                                 (len(ins) + k) * room)
          outs[n][...] = 77
      monkeypatch.setenv('SODA_HIP_HOST_TRACE', '1')
      capfd.readouterr()
      prog.run(ins if how == 'pageable' else pin_in, outputs=outs)
      said = capfd.readouterr().err
      monkeypatch.delenv('SODA_HIP_HOST_TRACE')
      n_in, n_out = len(st.input_names), len(st.output_names)
      direct = how.startswith('pinned') and how != 'pinned, slots'
      want_out = n_out if direct else 0
      assert 'in place: %d of %d inputs, %d of %d outputs' % (
          n_in if direct else 0, n_in, want_out, n_out) in said, (name, how,
                                                                  said)
      if how == 'pinned in bands' and kb == 16:    # (4 MiB: one chunk)
        assert ' bands of ' in said, (name, said)
      runs[how] = {n: a.copy() for n, a in outs.items()}
  for how, outs in runs.items():
    for o in st.output_names:
      assert np.array_equal(outs[o].view(np.uint8),
                            runs['pageable'][o].view(np.uint8)), (name, o, how)
  for o in st.output_names:        # (the pageable run itself: box vs rest)
    if st.preserve_border:
      continue
    lo, hi = st.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    mask = np.ones(shape, bool)
    mask[idx] = False
    assert (runs['pinned'][o][mask] == 77).all(), (name, o)
    assert (runs['pinned'][o][idx] != 77).any(), (name, o)


def test_what_may_be_registered(built):
  """soda_hip_host_register takes whole pages of the caller's own, nothing that
  starts inside a heap page; unregistering what it did not register is an
  error, not a call into the runtime."""
  import ctypes
  from soda_amd import runtime
  lib = runtime.library()
  with runtime.PinnedBuffer(1 << 16) as buf:
    a = buf.array((128, 64), np.float32, 4096)
    assert a.ctypes.data % 4096 == 0 and a.flags.c_contiguous
    inside = ctypes.c_void_p(a.ctypes.data + 64)
    assert lib.soda_hip_host_register(inside, 4096) != 0
    assert 'page boundary' in runtime.last_error()
    assert lib.soda_hip_host_unregister(inside) != 0
    assert lib.soda_hip_host_register(ctypes.c_void_p(a.ctypes.data), 0) != 0
    with pytest.raises(Exception):
      buf.array((1 << 20,), np.float32)


@pytest.mark.parametrize('name,tile,extent,iterate,banks', [
    ('blur.soda', None, (2000, 700), None, 1),
    ('jacobi2d.soda', (256,), (256, 1500), 6, 1),  # 6 x (0, 1) late, 12 ghost rows
    ('heat3d.soda', (64, 32), (64, 32, 400), 2, 1),
    ('sobel2d.soda', (512,), (512, 900), None, 1),
    # banked tensors: woven by the host threads in the pack / unpack step
    ('blur.soda', (2048,), (2048, 600), None, 4),
    ('jacobi2d.soda', (256,), (256, 1500), 6, 2),
    ('heat3d.soda', (96, 32), (96, 32, 300), 2, 3),
    # two inputs, one delayed two tile rows by the host: un-delayed in the
    # pack step (array exactly one tile wide, or the n-D oracle does not apply)
    ('denoise2d.soda', (512,), (512, 900), None, 1),
])
def test_wire_host_banks_in_bands(built, monkeypatch, capfd, name, tile, extent,
                                  iterate, banks):
  """<app>_kernel on host banks, every tensor in place on the dense view: the
  call goes through the host-array entry, here forced into bands of 16 KiB
  chunks (copy-in, kernels, copy-out overlapped; window runs with iterate x
  reach ghost rows on the late program's one-sided window) -- the same cells
  as the n-D oracle, and the same banks as one whole run."""
  from soda_amd import core, stream
  from oracle import frt_layout, numpy_oracle
  import re
  text = open(soda_path(name)).read()
  if banks > 1:
    text = re.sub(r'(input|output) dram [\d.]+',
                  r'\1 dram ' + '.'.join(map(str, range(banks))), text)
  stencil = core.from_text(text, tile_size=tile, iterate=iterate)
  inputs = _inputs(stencil, extent, seed=11)
  layout = stream.WireLayout(stencil, extent)
  assert layout.tiles == 1 and max(layout.bank_count.values()) == banks
  in_banks = frt_layout.scatter(layout, inputs)
  prog = stream.StreamProgram(stencil, dense=True)
  runs = {}
  try:
    assert any(t.startswith('wire_') for t in prog.specs) == (banks > 1)
    for mode in ('bands', 'whole'):
      monkeypatch.setenv('SODA_HIP_HOST_BANDS', '1' if mode == 'bands' else '0')
      monkeypatch.setenv('SODA_HIP_HOST_CHUNK_KB', '16')
      monkeypatch.setenv('SODA_HIP_HOST_TRACE', '1')
      out_banks = frt_layout.alloc(layout, stencil.output_names)
      capfd.readouterr()
      prog.run_banked_host(out_banks, in_banks, layout.cycle_count)
      said = capfd.readouterr().err
      assert prog.last_mode == 'dense'
      # (the library's own account of the call, soda_host.cpp run_banded)
      assert (' bands of ' in said) == (mode == 'bands'), said
      got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
             for o, t in zip(stencil.output_names, stencil.output_types)}
      frt_layout.gather(layout, out_banks, got)
      runs[mode] = got
  finally:
    prog.close()
  want = numpy_oracle.run(stencil, inputs) if iterate is None or iterate <= 2 \
      else None
  if want is None:
    from oracle import c_oracle
    want = c_oracle.COracle(stencil).run(inputs)
  for o in stencil.output_names:
    lo, hi = stencil.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert runs['bands'][o][idx].any()
    assert np.array_equal(runs['bands'][o][idx], want[o][idx]), o
    assert np.array_equal(runs['whole'][o], runs['bands'][o]), o


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['blur', 'jacobi2d_4tiles', 'heat3d_2x2tiles',
                                 'denoise2d', 'blur_4banks', 'jacobi2d_2banks'])
@pytest.mark.parametrize('dense', [True, False])
def test_committed_wire_vectors_on_the_gpu(built, tag, dense):
  """tests/golden/wire_*.npz: <app>_kernel on the COMMITTED input banks (host
  banks, as the generated host passes them); what the host gathers from the
  banks it leaves equals the committed outputs -- several tiles, 3-D tiles, a
  delayed input, two and four banks, dense view and linear form."""
  import importlib.util
  from conftest import GOLDEN_DIR
  from soda_amd import stream
  from oracle import frt_layout
  spec = importlib.util.spec_from_file_location(
      'make_wire_golden', os.path.join(GOLDEN_DIR, 'make_wire_golden.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  case = [c for c in mod.CASES if c[0] == tag][0]
  gold = np.load(os.path.join(GOLDEN_DIR, 'wire_%s.npz' % tag))
  st = mod.program(case)
  extent = tuple(int(x) for x in gold['extent'])
  lay = stream.WireLayout(st, extent)
  in_banks = {n: [np.ascontiguousarray(gold['inbank%d_%s' % (b, n)])
                  for b in range(lay.bank_count[n])] for n in st.input_names}
  out_banks = frt_layout.alloc(lay, st.output_names)
  prog = stream.StreamProgram(st, dense=dense)
  try:
    prog.run_banked_host(out_banks, in_banks, int(gold['cycle_count']))
  finally:
    prog.close()
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, out_banks, got)
  for o in st.output_names:
    assert np.array_equal(got[o], gold['out_' + o]), o


@pytest.mark.gpu
def test_narrow_tiles_take_the_dense_view_on_host_banks_only(built):
  """Tiles under 256 cells leave most of a marching strip idle, so device-
  resident banks run the linear form there; host banks take the dense view
  anyway -- the copies dominate and the dense view is what lets them overlap in
  bands (soda_hip_stream_set_device_dense_min_tile).  Same cells either way."""
  import torch
  from soda_amd import core, stream
  from oracle import frt_layout, numpy_oracle
  stencil = core.from_file(soda_path('heat3d.soda'))         # 32 x 32 tiles
  extent = (32, 32, 40)
  inputs = _inputs(stencil, extent, seed=21)
  layout = stream.WireLayout(stencil, extent)
  in_banks = frt_layout.scatter(layout, inputs)
  prog = stream.StreamProgram(stencil)                       # dense=None
  try:
    assert 'dense' in prog.specs
    out_host = frt_layout.alloc(layout, stencil.output_names)
    prog.run_banked_host(out_host, in_banks, layout.cycle_count)
    assert prog.last_mode == 'dense'
    dev_in = {n: [torch.from_numpy(b).cuda() for b in bs]
              for n, bs in in_banks.items()}
    dev_out = {n: [torch.zeros_like(torch.from_numpy(b)).cuda() for b in bs]
               for n, bs in out_host.items()}
    prog.run_banked_device({n: [t.data_ptr() for t in ts]
                            for n, ts in dev_out.items()},
                           {n: [t.data_ptr() for t in ts]
                            for n, ts in dev_in.items()}, layout.cycle_count,
                           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert prog.last_mode == 'linear'
    out_dev = {n: [t.cpu().numpy() for t in ts] for n, ts in dev_out.items()}
  finally:
    prog.close()
  want = numpy_oracle.run(stencil, inputs)
  for banks in (out_host, out_dev):
    got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
           for o, t in zip(stencil.output_names, stencil.output_types)}
    frt_layout.gather(layout, banks, got)
    for o in stencil.output_names:
      lo, hi = stencil.valid_box(extent, o)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      assert got[o][idx].any() and np.array_equal(got[o][idx], want[o][idx])


@pytest.mark.gpu
def test_wire_banks_at_any_address(built):
  """The bank copy kernels move 16 bytes per bank per thread when every bank
  is 16-byte aligned, element by element otherwise: device banks that start 4
  bytes into an allocation give the same streams."""
  import re
  import torch
  from soda_amd import core, stream
  text = open(soda_path('jacobi2d.soda')).read()
  text = re.sub(r'input dram [^\n]*', 'input dram 0.1 float: t1(32, *)', text)
  text = re.sub(r'output dram [\d.]+ \w+:', 'output dram 2.3 float:', text)
  stencil = core.from_text(text)
  extent = (32, 45)
  layout = stream.WireLayout(stencil, extent)
  per_bank = layout.buf_elems['t1'] // 2
  rng = np.random.default_rng(3)
  host = [rng.random(per_bank, dtype=np.float32) for _ in range(2)]
  prog = stream.StreamProgram(stencil, dense=True)
  try:
    results = []
    for lead in (0, 1):
      ins = [torch.zeros(per_bank + 4, device='cuda') for _ in range(2)]
      outs = [torch.full((per_bank + 4,), -7.0, device='cuda') for _ in range(2)]
      for t, h in zip(ins, host):
        t[lead:lead + per_bank] = torch.from_numpy(h).cuda()
      prog.run_banked_device(
          {'t0': [t.data_ptr() + 4 * lead for t in outs]},
          {'t1': [t.data_ptr() + 4 * lead for t in ins]}, layout.cycle_count,
          stream=torch.cuda.current_stream().cuda_stream)
      torch.cuda.synchronize()
      for t in outs:   # nothing outside the bank
        assert (t[:lead] == -7).all() and (t[lead + per_bank:] == -7).all()
      results.append([t[lead:lead + per_bank].cpu().numpy() for t in outs])
  finally:
    prog.close()
  for a, b in zip(*results):
    assert np.array_equal(a, b)
  assert any(a.any() for a in results[0])


@pytest.mark.parametrize('name,iterate,fuse,pipe,extent', [
    ('jacobi2d.soda', 12, 12, 4, (1000, 333)),
    ('jacobi2d.soda', 12, 12, 2, (520, 97)),
    ('jacobi2d.soda', 12, 12, 3, (256, 64)),
    ('jacobi2d.soda', 14, 12, 6, (777 * 4, 41)),     # 12 pipelined + 2 single
    ('seidel2d.soda', 8, 8, 4, (640, 200)),
    ('blur.soda', 4, 4, 2, (640, 200)),              # two stages per iteration
    ('blur.soda', 6, 6, 3, (1024, 50)),
    ('coupled2d.soda', 4, 4, 2, (300, 90)),          # 2 tensors cross per wave
    ('coupled2d.soda', 6, 6, 3, (300, 90)),
    ('heat3d.soda', 6, 2, 2, (300, 24, 40)),         # 3-D: tile planes handed on
    ('jacobi3d.soda', 4, 2, 2, (64, 21, 33)),
])
def test_stage_pipelined_blocks(built, name, iterate, fuse, pipe, extent):
  """The fused iterations split over the waves of a block (rows handed from
  wave to wave through an LDS ring, one barrier per row step): bit-identical
  to the oracle, like the one-wave kernels."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  opts = lower.LowerOptions(fuse=(fuse,), pipe=pipe)
  mod = lower.lower(stencil, lower.LowerOptions(fuse=(fuse,), pipe=pipe, vec=4))
  assert any('_pipe%d' % pipe in k.name for k in mod.kernels)
  _check(stencil, extent, opts, oracle='c')


TWO_STAGE_3D = """kernel: smooth3d
burst width: 64
unroll factor: 2
iterate: 4
input float: a(32, 32, *)
local float: m(0, 0, 0) = (a(-1, 0, 0) + a(0, 0, 0) + a(1, 0, 0) + a(0, -1, 0) + a(0, 1, 0) + a(0, 0, 1)) * 0.125f
output float: b(0, 0, 0) = m(0, 0, -1) * 0.25f + (m(-1, 0, 0) + m(1, 0, 0)) * 0.125f + m(0, 0, 0) * 0.25f + m(0, 0, 1) * 0.25f
"""


DOUBLE_2D = """kernel: jacobi2d_f64
burst width: 64
unroll factor: 2
iterate: 4
input double: a(32, *)
output double: b(0, 0) = (a(0, 1) + a(1, 0) + a(0, 0) + a(-1, 0) + a(0, -1)) * 0.2
"""


@pytest.mark.gpu
@pytest.mark.parametrize('name,iterate,opts,extent,waves,border', [
    ('heat3d.soda', 4, dict(fuse=(2,)), (512, 16, 20), 2, None),
    ('heat3d.soda', 6, dict(fuse=(2,)), (300, 21, 24), 2, None),   # ragged strip
    ('heat3d.soda', 5, dict(fuse=(2,)), (256, 12, 18), 1, None),   # one strip: no neighbour
    ('heat3d.soda', 4, dict(fuse=(2,)), (64, 9, 40), 1, None),
    ('heat3d.soda', 2, dict(fuse=(2,)), (1024, 9, 12), 4, None),
    ('jacobi3d.soda', 4, dict(fuse=(2,)), (508, 14, 16), 2, None),
    ('heat3d.soda', 5, dict(fuse=(2,)), (300, 24, 40), 2, 'preserve'),
    (TWO_STAGE_3D, 4, dict(fuse=(2,)), (384, 13, 17), 2, None),    # a local crosses strips too
    # 2-D on request: 4, 8 and 12 fused iterations, up to 3 strips side by side
    ('jacobi2d.soda', 9, dict(fuse=(4,), xshare=True), (512, 130), 2, None),
    ('jacobi2d.soda', 16, dict(fuse=(8,), xshare=True), (700, 90), 3, None),
    ('jacobi2d.soda', 12, dict(fuse=(12,), xshare=True), (256, 200), 1, None),
    (DOUBLE_2D, 4, dict(fuse=(2,), xshare=True), (200, 60), 2, None),   # 8-byte cells
    # taps that reach two cells, or touch the newest plane: the generator
    # falls back to overlapping strips (waves = 0)
    ('blur.soda', 4, dict(fuse=(2,), xshare=True), (1024, 50), 0, None),
    ('coupled2d.soda', 4, dict(fuse=(2,), xshare=True), (300, 90), 0, None),
])
def test_rows_shared_by_the_waves_of_a_block(built, name, iterate, opts, extent,
                                              waves, border):
  """Fused kernels whose block covers the whole row (MarchConfig.xshare): every
  wave a strip of 64 valid lanes, the strips' end cells -- of the inputs and of
  every fused iteration's intermediate result -- handed to the neighbours
  through LDS, one barrier per row step.  Bit-identical to the oracle."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  kw = dict(iterate=iterate)
  if border:
    kw['border'] = border
  if name.endswith('.soda'):
    stencil = core.from_file(soda_path(name), **kw)
  else:
    stencil = core.from_text(name, **kw)
  lo = lower.LowerOptions(**opts)
  mod = lower.lower(stencil, runtime.resolve_options(stencil, lo, extent))
  fused = [k for k in mod.kernels if k.tune and k.tune.get('fused', 1) > 1]
  assert fused
  if waves:
    assert all(k.name.endswith('_xs%d' % waves) for k in fused), \
        [k.name for k in mod.kernels]
    assert all(k.tune['max_extent0'] >= extent[0] for k in fused)
  else:
    assert not any('_xs' in k.name or k.tune['max_extent0'] for k in fused)
  _check(stencil, extent, lo, oracle='c')


@pytest.mark.gpu
@pytest.mark.parametrize('name,extent,iterate,fuse,keep', [
    ('blur.soda', (640, 300), 5, 'direct', (60, 180)),          # locals in HBM scratch
    ('denoise2d.soda', (256, 200), 1, (), (50, 120)),           # two inputs, one pass
    ('jacobi2d.soda', (520, 400), 40, (12, 4), (100, 300)),     # both sides trimmed
    ('jacobi2d.soda', (520, 400), 40, (12, 4), (0, 250)),       # a global border below
    ('jacobi2d.soda', (520, 400), 17, (4,), (150, 400)),        # ... above
    ('heat3d.soda', (256, 24, 90), 10, (2,), (30, 60)),
    ('blur.soda', (640, 300), 6, (3,), (40, 200)),              # reaches 0 below, 2 above
    # the slabs of the 8-GPU bench run (middle rank, end rank), benched depths
    ('jacobi2d.soda', (8192, 1224), 100, 'bench', (100, 1124)),
    ('jacobi2d.soda', (8192, 1124), 100, 'bench', (0, 1024)),
    ('jacobi2d.soda', (8192, 1224), 26, (13,), (100, 1124)),
])
def test_runs_that_keep_a_row_range_skip_the_rest(built, name, extent, iterate,
                                                  fuse, keep):
  """soda_hip_run_device_cone: a run that only has to deliver rows [lo, hi)
  (a slab's own rows) launches every pass on the rows that can still reach
  them.  The kept rows equal the untrimmed run bit for bit, and the launches
  covered fewer rows."""
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  ins = _inputs(stencil, extent, 7)
  if fuse == 'bench':
    fuse = lower.DEFAULT_FUSE
  opts = lower.LowerOptions(strategy='direct') if fuse == 'direct' else \
      lower.LowerOptions(fuse=fuse)
  with runtime.Program(stencil, opts, extent=extent) as prog:
    src = [torch.from_numpy(ins[n]).cuda() for n in stencil.input_names]
    full = [torch.zeros_like(src[0]) for _ in stencil.output_names]
    part = [torch.full_like(src[0], 77) for _ in stencil.output_names]
    s = torch.cuda.current_stream().cuda_stream
    prog.run_device([t.data_ptr() for t in full], [t.data_ptr() for t in src],
                    extent, stream=s)
    rows_full = prog.last_rows()
    launches = prog.last_launches()[0]
    prog.run_device([t.data_ptr() for t in part], [t.data_ptr() for t in src],
                    extent, stream=s, keep=keep)
    rows_part = prog.last_rows()
    assert prog.last_launches()[0] == launches
    torch.cuda.synchronize()
  passes = rows_full // extent[-1]       # rows are counted once per pass
  assert rows_full == passes * extent[-1] and launches % passes == 0
  assert rows_part < rows_full - 2 * max(1, passes - 1)   # passes narrow
  lo, hi = stencil.valid_box(extent)
  box = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  a = full[0].cpu().numpy()[box]
  b = part[0].cpu().numpy()[box]
  k0, k1 = max(keep[0], lo[-1]) - lo[-1], min(keep[1], hi[-1]) - lo[-1]
  assert k1 > k0
  assert np.array_equal(a[k0:k1], b[k0:k1])
  if extent[0] >= 8192:      # the bench's slabs: against the oracle as well
    from oracle import c_oracle
    want = c_oracle.COracle(stencil).run(ins)[stencil.output_names[0]][box]
    assert np.array_equal(b[k0:k1], want[k0:k1])


@pytest.mark.gpu
def test_row_covering_kernels_refuse_longer_rows(built):
  """A kernel built for rows of <= 512 cells has no strip for cell 512: the
  library returns SODA_HIP_ERR_INVALID instead of computing a seam wrongly."""
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('heat3d.soda'), iterate=2)
  with runtime.Program(stencil, lower.LowerOptions(fuse=(2,)),
                       extent=(512, 16, 12)) as prog:
    a = torch.rand((12, 16, 768), device='cuda')
    b = torch.full_like(a, -1.0)
    lib = runtime.library()
    outs = (ctypes.c_void_p * 1)(b.data_ptr())
    ins = (ctypes.c_void_p * 1)(a.data_ptr())
    ext = (ctypes.c_int32 * 3)(768, 16, 12)
    rc = lib.soda_hip_run_device(prog._handle, outs, ins, ext, 2, None)
    assert rc == 1, rc                      # SODA_HIP_ERR_INVALID
    assert 'at most 512 cells' in runtime.last_error()
    torch.cuda.synchronize()
    assert bool((b == -1.0).all())


@pytest.mark.parametrize('name,iterate,opts,extent', [
    ('jacobi2d.soda', 1, dict(), (520, 61)),
    ('jacobi2d.soda', 9, dict(fuse=(4,)), (520, 61)),          # 4 + 4 + 1
    ('jacobi2d.soda', 12, dict(fuse=(12,)), (1000, 130)),
    ('jacobi2d.soda', 8, dict(fuse=(8,), pipe=4), (776, 90)),
    ('jacobi2d.soda', 5, dict(strategy='direct'), (260, 33)),
    ('seidel2d.soda', 6, dict(fuse=(3,)), (300, 77)),
    ('blur.soda', 3, dict(fuse=(3,)), (640, 50)),              # box through a local
    ('blur.soda', 2, dict(strategy='direct'), (640, 50)),
    ('coupled2d.soda', 4, dict(fuse=(2,)), (300, 90)),         # two paired tensors
    ('heat3d.soda', 5, dict(fuse=(2,)), (300, 24, 40)),
    ('jacobi3d.soda', 3, dict(strategy='direct'), (64, 20, 18)),
])
def test_border_preserve(built, name, iterate, opts, extent):
  """`border: preserve` (defined by this build, core.Stencil.check_preserve):
  the whole grid is defined after any number of iterations and equals the
  oracle bit for bit, border cells included."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate, border='preserve')
  assert stencil.valid_box(extent) == ((0,) * stencil.dim, tuple(extent))
  _check(stencil, extent, lower.LowerOptions(**opts), oracle='c')


PARAM3D = """kernel: wsum3d
burst width: 64
unroll factor: 2
iterate: 3
input float: a(32, 32, *)
param float: c[2][3]
param int32: shift
output float: b(0, 0, 0) = a(0, 0, -1) * c(0, 0) + a(0, -1, 0) * c(0, 1) + a(-1, 0, 0) * c(0, 2) + a(1, 0, 0) * c(1, 0) + a(0, 1, 0) * c(1, 1) + a(0, 0, 1) * c(1, 2) + shift
"""

PARAM_INT = """kernel: lut2d
burst width: 64
unroll factor: 2
iterate: 1
input int16: x(32, *)
param int16: k[4]
local int32: t(0, 0) = x(0, 0) * k(0) + x(1, 0) * k(1)
output int16: y(0, 0) = int16((t(0, 0) + t(0, 1) * k(2)) / 4) + k(3)
"""


@pytest.mark.parametrize('text,extent,opts', [
    (None, (520, 61), dict()),                       # conv2d.soda, iterate 2
    (None, (520, 61), dict(fuse=(2,))),
    (None, (260, 33), dict(strategy='direct')),
    (None, (259, 33), dict(strategy='direct')),      # one cell per thread
    (PARAM3D, (64, 20, 18), dict()),
    (PARAM3D, (64, 20, 18), dict(fuse=(2,))),
    (PARAM3D, (40, 12, 10), dict(strategy='direct')),
    (PARAM_INT, (512, 40), dict()),
    (PARAM_INT, (512, 40), dict(strategy='direct')),
])
def test_param_arrays(built, text, extent, opts):
  """`param` arrays (reference grammar.py:41-45; its kernel emitter never
  delivered them): small read-only arrays addressed absolutely, C order,
  passed after the inputs."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = (core.from_text(text) if text else
             core.from_file(soda_path('conv2d.soda')))
  assert stencil.param_names
  _check(stencil, extent, lower.LowerOptions(**opts), oracle='c')
  bordered = (core.from_text(text, border='preserve') if text else
              core.from_file(soda_path('conv2d.soda'), border='preserve'))
  _check(bordered, extent, lower.LowerOptions(**opts))


def test_scheduler_picks_the_cheapest_pass_mix(built):
  """100 iterations with kernels of 12 / 8 / 4 / 1 fused iterations: the
  library picks the pass mix of least modelled time for the extent (queried
  through Program.schedule, the C ABI's soda_hip_program_schedule), runs
  exactly that, and the result is the same bits whatever the mix."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  import torch
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  extent = (1024, 600)
  a = torch.rand((600, 1024), device='cuda')
  outs = []
  for fuse in ((12, 8, 4), (12, 4)):
    b = torch.empty_like(a)
    with runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                         extent=extent) as prog:
      prog.run_device([b.data_ptr()], [a.data_ptr()], extent)
      torch.cuda.synchronize()
      launches, deepest = prog.last_launches()
      sched = prog.schedule(extent, 100)
      assert sum(t * c for t, c in sched.items()) == 100
      assert launches == sum(sched.values()) <= 25
      assert deepest == sched.get(12, 0) and sched.get(1, 0) <= 3
      times = prog.pass_times(extent)[0]
      cost = lambda mix: sum(times[t] * c for t, c in mix.items())
      assert cost(sched) <= cost({12: 8, 4: 1}) + 1e-6
    outs.append(b[100:500, 100:924].clone())
  assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('tag,soda,border', [
    ('jacobi2d', 'jacobi2d.soda', None),       # hand-written C kernels' outputs
    ('blur', 'blur.soda', None),
    ('heat3d', 'heat3d.soda', None),
    ('skew2d', 'skew2d.soda', None),
    ('sobel2d', 'sobel2d.soda', None),
    ('denoise2d', 'denoise2d.soda', None),
    ('jacobi3d', 'jacobi3d.soda', None),       # hand-written nests, round 3
    ('denoise3d', 'denoise3d.soda', None),
    ('jacobi2d_preserve', 'jacobi2d.soda', 'preserve'),
    ('heat3d_preserve', 'heat3d.soda', 'preserve'),
    ('conv2d', 'conv2d.soda', None),
    ('conv2d_preserve', 'conv2d.soda', 'preserve')])
@pytest.mark.parametrize('strategy', ['auto', 'direct'])
def test_committed_golden_vectors_on_gpu(built, tag, soda, border, strategy):
  """The committed vectors (tests/golden/make_golden.py) through the kernels."""
  import os
  from conftest import GOLDEN_DIR
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  data = np.load(os.path.join(GOLDEN_DIR, '%s.npz' % tag))
  st = core.from_file(soda_path(soda), iterate=int(data['iterate']),
                      border=border)
  ins = {n: data['in_' + n] for n in st.input_names + st.param_names}
  extent = tuple(ins[st.input_names[0]].shape[::-1])
  with runtime.Program(st, lower.LowerOptions(strategy=strategy, fuse=(2,)),
                       extent=extent) as prog:
    got = prog.run(ins)
  for o in st.output_names:
    lo, hi = st.valid_box(extent, o)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(got[o][idx], data['out_' + o][idx])


WIDE_INT = """kernel: wideint
burst width: 64
unroll factor: 2
iterate: 1
input int32: a(64, *)
local int32: s(0, 0) = a(-6, -3) * 3 + a(-2, -1) * 5 - a(0, 0) + a(3, 1) * 7 + a(7, 2)
output int32: b(0, 0) = s(0, 0) / 3 + a(-5, 2) * a(6, -3) - a(1, 0)
"""

WIDE_FLOAT = """kernel: widefloat
burst width: 64
unroll factor: 2
iterate: 1
input float: a(64, *)
output float: b(1, -1) = (a(-9, -2) + a(12, 0) * 0.25f) * (a(0, 1) - a(3, -4) / (1.5f + a(2, 2) * a(2, 2))) + sqrt(a(10, 3) + 1.0f) + a(-1, -1) * a(11, -4)
"""


@pytest.mark.parametrize('text,extent', [
    ('contrast.soda', (1024, 200)),
    ('contrast.soda', (520, 77)),           # a ragged second strip, odd chunks
    ('contrast.soda', (2052, 131)),
    ('contrast.soda', (20, 40)),            # narrower than one lane's reach
    (WIDE_INT, (1028, 150)),                # taps on both sides of the cell
    (WIDE_INT, (36, 70)),
    (WIDE_FLOAT, (1540, 99)),               # off-centre store
])
def test_wide_windows_through_lds(built, text, extent):
  """`ldswin` (soda_amd/codegen/hip/ldswin.py): window rows in an LDS ring,
  8 cells per lane, all lanes valid -- contrast with the reference's rebalanced
  association (its six groups folded into one stage as cast sub-expressions),
  windows on both sides of the cell, integer cells, an off-centre store; the
  family `auto` picks for contrast.  Bit for bit against the C oracle, nothing
  written outside the valid box."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  st = core.from_file(soda_path(text)) if text.endswith('.soda') else \
      core.from_text(text)
  with runtime.Program(st, lower.LowerOptions(strategy='ldswin'),
                       extent=extent) as prog:
    assert [p.kind for p in prog.module.passes] == ['ldswin']
  _check(st, extent, lower.LowerOptions(strategy='ldswin'), oracle='c')
  # eight rows per step (eight waves per block, a 32-row ring)
  _check(st, extent, lower.LowerOptions(strategy='ldswin', waves_y=8,
                                        chunk_rows=40), oracle='c')
  if text == 'contrast.soda':
    with runtime.Program(st, lower.LowerOptions(), extent=extent) as prog:
      assert [p.kind for p in prog.module.passes] == ['ldswin']
    # ... and it can be told not to (the 7-stage marching form)
    with runtime.Program(st, lower.LowerOptions(strategy='march'),
                         extent=extent) as prog:
      assert {p.kind for p in prog.module.passes} == {'march2d'}
    _check(st, extent, lower.LowerOptions(strategy='march'), oracle='c')


@pytest.mark.parametrize('name,iterate,opts,extent,border', [
    ('jacobi2d.soda', 24, dict(fuse=(12,)), (2050, 1031), None),   # V=2 rows
    ('jacobi2d.soda', 24, dict(fuse=(12,), pipe=4), (2052, 1031), None),
    ('jacobi2d.soda', 13, dict(fuse=(12, 4)), (1027, 517), None),  # V=1 rows
    # the bench's deepest pass (4 halo lanes per side of a half strip, not 3)
    ('jacobi2d.soda', 13, dict(fuse=(13,)), (2050, 1031), None),
    ('jacobi2d.soda', 26, dict(fuse=(13,)), (1027, 517), None),
    ('jacobi2d.soda', 26, dict(fuse=(13,)), (2304, 1100), None),   # V=4, ragged
    ('jacobi2d.soda', 38, dict(fuse=(13, 12)), (4100, 700), None),
    ('jacobi2d.soda', 9, dict(fuse=(8,)), (1027, 517), 'preserve'),
    ('blur.soda', 1, dict(), (4100, 2057), None),                  # u16, V=4
    ('heat3d.soda', 6, dict(fuse=(2,)), (258, 131, 70), None),
    ('heat3d.soda', 5, dict(fuse=(2,)), (130, 70, 33), 'preserve'),
    ('denoise2d.soda', 1, dict(), (1026, 519), None),
])
def test_awkward_mid_size_grids(built, name, iterate, opts, extent, border):
  """Row lengths that force narrower vectors, ragged last strips / chunks /
  tiles, at sizes where every strip-chunk combination occurs; against the
  OpenMP C oracle."""
  from soda_amd import core
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate, border=border)
  _check(stencil, extent, lower.LowerOptions(**opts), oracle='c')


@pytest.mark.parametrize('name,extent,iterate,fuse,strategy', [
    ('jacobi2d.soda', (32768, 32768), 4, (4,), 'auto'),
    ('jacobi2d.soda', (32768, 32768), 2, (), 'direct'),
    ('heat3d.soda', (1024, 1024, 1024), 2, (2,), 'auto'),
])
def test_beyond_4gib_arrays(built, name, extent, iterate, fuse, strategy):
  """4 GiB per array: byte offsets past 2^32, where the 32-bit buffer offsets
  of the marching kernels rely on the per-wave window rebasing and the direct
  kernels on 64-bit indices.  Device-resident; the fused and the
  one-iteration-per-launch schedules must agree bit for bit over the whole
  array, and three slabs along the last dimension (start, across 2 GiB, end)
  must equal the CPU oracle run on just those slabs."""
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import numpy_oracle
  stencil = core.from_file(soda_path(name), iterate=iterate)
  shape = tuple(extent[::-1])
  a = torch.rand(shape, device='cuda')
  outs = []
  for f in ((fuse, ()) if fuse else ((),)):
    b = torch.empty_like(a)
    with runtime.Program(stencil, lower.LowerOptions(fuse=f, strategy=strategy),
                         extent=extent) as prog:
      prog.run_device([b.data_ptr()], [a.data_ptr()], extent)
      torch.cuda.synchronize()
    outs.append(b)
  r = iterate                       # cells the border eats per side
  inner = tuple(slice(r, -r) for _ in extent)
  if len(outs) == 2:
    assert torch.equal(outs[0][inner], outs[1][inner])
  n_last = extent[-1]
  thick = 48 if len(extent) == 2 else 12
  for start in (0, n_last // 2 - thick // 2, n_last - thick):
    sl = slice(start, start + thick)
    part = {stencil.input_names[0]: a[sl].cpu().numpy()}
    want = numpy_oracle.run(stencil, part)[stencil.output_names[0]]
    got = outs[0][sl].cpu().numpy()
    assert np.array_equal(got[inner], want[inner]), start
  del outs, a
  torch.cuda.empty_cache()


def test_c_abi_refuses_extents_the_kernels_cannot_run(built):
  """The vector-width constraint is enforced behind the C ABI itself (a V-wide
  kernel on a ragged row would write past the row end): soda_hip_run_device
  returns SODA_HIP_ERR_INVALID, nothing is launched."""
  import torch
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=1)
  for strategy in ('auto', 'direct'):
    with runtime.Program(stencil, lower.LowerOptions(strategy=strategy, vec=4,
                                                     fuse=()),
                         extent=(512, 64)) as prog:
      a = torch.rand((64, 510), device='cuda')
      b = torch.full_like(a, -1.0)
      lib = runtime.library()
      outs = (ctypes.c_void_p * 1)(b.data_ptr())
      ins = (ctypes.c_void_p * 1)(a.data_ptr())
      ext = (ctypes.c_int32 * 2)(510, 64)
      rc = lib.soda_hip_run_device(prog._handle, outs, ins, ext, 1, None)
      assert rc == 1, rc                      # SODA_HIP_ERR_INVALID
      assert 'multiple of 4 cells' in runtime.last_error()
      torch.cuda.synchronize()
      assert bool((b == -1.0).all())


def test_calibrated_schedule(built):
  """soda_hip_program_calibrate: passes timed on the GPU, the schedule follows
  the clock, results do not change by a bit."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  from oracle import c_oracle
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=37)
  extent = (1024, 600)
  inputs = _inputs(stencil, extent, seed=4)
  want = c_oracle.COracle(stencil).run(inputs)['t0']
  with runtime.Program(stencil, lower.LowerOptions(fuse=(12, 8, 4)),
                       extent=extent, calibrate=False) as prog:
    model, measured = prog.pass_times(extent)
    assert not measured and set(model) == {12, 8, 4, 1}
    before = prog.run(inputs)['t0']
    assert not prog.pass_times(extent)[1]     # told not to: still the model
    times = prog.calibrate(extent)
    assert prog.pass_times(extent)[1]
    assert all(0.5 < v < 5000 for v in times.values()), times
    sched = prog.schedule(extent, 37)
    assert sum(t * c for t, c in sched.items()) == 37
    after = prog.run(inputs)['t0']
    assert prog.last_launches()[0] == sum(sched.values())
  lo, hi = stencil.valid_box(extent)
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  assert np.array_equal(before[idx], want[idx])
  assert np.array_equal(after, before)
  # by default a program calibrates by itself on the first run of an extent
  with runtime.Program(stencil, lower.LowerOptions(fuse=(12, 8, 4)),
                       extent=extent) as prog:
    assert not prog.pass_times(extent)[1]
    auto = prog.run(inputs)['t0']
    assert prog.pass_times(extent)[1]
    assert not prog.pass_times((1024, 608))[1]      # that extent only
  assert np.array_equal(auto, before)


@pytest.mark.parametrize('name,extent,fuse,iterate', [
    ('jacobi2d.soda', (8192, 8192), (12, 8, 4), 100),
    ('jacobi2d.soda', (8192, 1224), (12, 8, 4), 100),   # slab of an 8-GPU run
    ('jacobi2d.soda', (8192, 2248), (12, 8, 4), 100),
    ('heat3d.soda', (512, 512, 80), (2,), 8),
])
def test_model_schedule_is_close_to_the_calibrated_one(built, name, extent,
                                                       fuse, iterate):
  """The constants of the library's time model are fits (soda_hip.cpp,
  tools/fit_model.py); they decide the schedule of every caller that turns
  calibration off and the exchange interval of a slab group.  Priced with the
  MEASURED pass times, the schedule the model picks must cost at most 10 %
  more than the one the clock picks."""
  from soda_amd import core, runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name), iterate=iterate)
  with runtime.Program(stencil, lower.LowerOptions(fuse=fuse), extent=extent,
                       calibrate=False) as prog:
    by_model = prog.schedule(extent, iterate)
    model_us, _ = prog.pass_times(extent)
    clock_us = prog.calibrate(extent, launches=8)
    by_clock = prog.schedule(extent, iterate)
  cost = lambda sched: sum(clock_us[t] * c for t, c in sched.items())
  assert cost(by_model) <= 1.10 * cost(by_clock), (by_model, by_clock,
                                                  model_us, clock_us)
  # and no pass is modelled off by more than a third
  for t, us in clock_us.items():
    assert 0.67 < model_us[t] / us < 1.5, (t, model_us, clock_us)

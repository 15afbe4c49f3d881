"""`sodac --hip-host`: the generated C++ host (soda_amd/codegen/hip/host.py)
has the reference's operator signature (frt/host.py:62-88) and runs without
Python, on the C ABI alone."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, soda_path


def _generate(tmp_path, soda, *flags):
  out = os.path.join(str(tmp_path), 'host.cpp')
  subprocess.run([sys.executable, '-m', 'soda_amd.sodac', soda_path(soda),
                  '--hip-host', out, *flags], cwd=ROOT, check=True)
  return out


@pytest.mark.parametrize('soda,flags,needle', [
    ('blur.soda', (), 'int blur(const uint16_t* var_input_ptr,'),
    ('jacobi2d.soda', ('--iterate', '24', '--hip-fuse', '12', '4'),
     'soda_hip_run_host_box(program, inputs, outputs, 24,'),
    ('heat3d.soda', (), 'const int tile_size_1 = 32'),
    ('conv2d.soda', (), 'const float* var_w_ptr'),      # params after outputs
])
def test_generated_host_compiles(built, tmp_path, soda, flags, needle):
  src = _generate(tmp_path, soda, *flags)
  text = open(src).read()
  assert needle in text
  assert 'const char* bitstream' in text and 'namespace soda' in text
  subprocess.run(['g++', '-std=c++17', '-Wall', '-Werror', '-c', src,
                  '-I', os.path.join(ROOT, 'include'), '-o',
                  os.path.join(str(tmp_path), 'host.o')], check=True)


@pytest.mark.gpu
def test_generated_host_runs_without_python(built, tmp_path):
  src = _generate(tmp_path, 'blur.soda')
  exe = os.path.join(str(tmp_path), 'blur_host')
  libdir = os.path.join(ROOT, 'soda_amd')
  subprocess.run(['g++', '-std=c++17', '-O1', src,
                  os.path.join(ROOT, 'tests', 'host', 'blur_main.cpp'),
                  '-I', os.path.join(ROOT, 'include'), '-L', libdir,
                  '-lsoda_hip', '-Wl,-rpath,' + libdir, '-o', exe], check=True)
  env = dict(os.environ)
  env['LD_LIBRARY_PATH'] = '/opt/rocm/lib:' + env.get('LD_LIBRARY_PATH', '')
  run = subprocess.run([exe], capture_output=True, text=True, env=env,
                       timeout=300)
  assert run.returncode == 0 and run.stdout.startswith('OK'), (
      run.stdout + run.stderr)


def test_generated_multi_gpu_host_compiles(built, tmp_path):
  """--hip-gpus N: the same operator signature, soda_hip_group_* behind it."""
  src = _generate(tmp_path, 'heat3d.soda', '--iterate', '9', '--hip-gpus', '4',
                  '--hip-fuse', '2')
  text = open(src).read()
  assert 'desc.num_slabs = 4;' in text and 'desc.reach_lo = 1;' in text
  assert 'soda_hip_group_run_host(group, inputs, outputs, 9,' in text
  assert 'soda_hip_run_host_box' not in text
  subprocess.run(['g++', '-std=c++17', '-Wall', '-Werror', '-c', src,
                  '-I', os.path.join(ROOT, 'include'), '-o',
                  os.path.join(str(tmp_path), 'host.o')], check=True)


@pytest.mark.gpu
def test_generated_multi_gpu_host_runs_without_python(built, tmp_path):
  """One blocking C++ call, four slabs with halo exchanges behind it (virtual
  devices on the one GPU): the p + q + r field is a fixed point of heat3d bit
  for bit on the valid box, the rest of the caller's array untouched."""
  src = _generate(tmp_path, 'heat3d.soda', '--iterate', '9', '--hip-gpus', '4',
                  '--hip-fuse', '2', '--hip-extent', '64', '48', '96')
  exe = os.path.join(str(tmp_path), 'heat3d_group')
  libdir = os.path.join(ROOT, 'soda_amd')
  subprocess.run(['g++', '-std=c++17', '-O1', src,
                  os.path.join(ROOT, 'tests', 'host', 'heat3d_group_main.cpp'),
                  '-I', os.path.join(ROOT, 'include'), '-L', libdir,
                  '-lsoda_hip', '-Wl,-rpath,' + libdir, '-o', exe], check=True)
  env = dict(os.environ)
  env['LD_LIBRARY_PATH'] = '/opt/rocm/lib:' + env.get('LD_LIBRARY_PATH', '')
  env['SODA_HIP_VIRTUAL_GPUS'] = '1'
  run = subprocess.run([exe], capture_output=True, text=True, env=env,
                       timeout=300)
  assert run.returncode == 0 and run.stdout.startswith('OK'), (
      run.stdout + run.stderr)


def _generate_wire(tmp_path, soda):
  out = os.path.join(str(tmp_path), 'wire.cpp')
  subprocess.run([sys.executable, '-m', 'soda_amd.sodac', soda_path(soda),
                  '--hip-wire-kernel', out], cwd=ROOT, check=True)
  return out


@pytest.mark.parametrize('soda,needle', [
    # outputs first, then inputs, then the burst count (ref frt/host.py:44-59)
    ('blur.soda', 'extern "C" void blur_kernel(void* bank_0_blur_y, '
     'void* bank_0_input, uint64_t coalesced_data_num)'),
    ('denoise2d.soda', 'extern "C" void denoise2d_kernel(void* bank_0_output, '
     'void* bank_0_f, void* bank_0_u, uint64_t coalesced_data_num)'),
    ('heat3d.soda', 'extern "C" void heat3d_kernel(void* bank_0_out, '
     'void* bank_0_in, uint64_t coalesced_data_num)'),
])
def test_generated_wire_kernel_compiles(built, tmp_path, soda, needle):
  """`sodac --hip-wire-kernel`: the reference kernel's C ABI, defined on
  libsoda_hip.so's stream object."""
  src = _generate_wire(tmp_path, soda)
  text = open(src).read()
  assert needle in text and 'soda_hip_stream_run_host' in text
  subprocess.run(['g++', '-std=c++17', '-Wall', '-Werror', '-c', src,
                  '-I', os.path.join(ROOT, 'include'), '-o',
                  os.path.join(str(tmp_path), 'wire.o')], check=True)


@pytest.mark.gpu
def test_wire_kernel_links_under_a_reference_style_host(built, tmp_path):
  """A C++ caller written from the reference host's text (sizes, scatter, the
  <app>_kernel call with its port order, gather) links against the generated
  definition and gets the closed-form answer -- no Python in the process."""
  src = _generate_wire(tmp_path, 'blur.soda')
  exe = os.path.join(str(tmp_path), 'blur_wire')
  libdir = os.path.join(ROOT, 'soda_amd')
  subprocess.run(['g++', '-std=c++17', '-O1', '-DSODA_CPP_BINDING', src,
                  os.path.join(ROOT, 'tests', 'host', 'blur_wire_main.cpp'),
                  '-I', os.path.join(ROOT, 'include'), '-L', libdir,
                  '-lsoda_hip', '-Wl,-rpath,' + libdir, '-o', exe], check=True)
  env = dict(os.environ)
  env['LD_LIBRARY_PATH'] = '/opt/rocm/lib:' + env.get('LD_LIBRARY_PATH', '')
  run = subprocess.run([exe], capture_output=True, text=True, env=env,
                       timeout=300)
  assert run.returncode == 0 and run.stdout.startswith('OK'), (
      run.stdout + run.stderr)


def _copy_box(lib, strided, dense, extent, lo, hi, to_dense, row0, threads):
  import ctypes
  import numpy as np
  dim = len(extent)
  item = strided.dtype.itemsize
  i32 = lambda v: (ctypes.c_int32 * dim)(*v)            # noqa: E731
  strides = [s // item for s in strided.strides[::-1]]
  base = strided.ctypes.data
  rc = lib.soda_hip_host_copy_box(
      ctypes.c_void_p(base), i32(strides), ctypes.c_void_p(dense.ctypes.data),
      i32(extent), i32(lo), i32(hi), dim, item, int(to_dense), row0, threads)
  assert rc == 0


@pytest.mark.parametrize('threads', [1, 0])
def test_pack_and_unpack_of_the_host_entry(built, threads):
  """soda_hip_host_copy_box (soda_host.cpp): the pack / unpack step of
  soda_hip_run_host_box -- the reference's scatter and gather loops,
  frt/host.py:181-249,340-427 -- on the calling thread and on the worker pool,
  against numpy slicing: dense and strided arrays, 1 to 4 dimensions, whole
  arrays and inner boxes, staging arrays that start at a row > 0; cells outside
  the box keep what they held."""
  import numpy as np
  from soda_amd import runtime
  lib = runtime.library()
  rng = np.random.default_rng(7)
  cases = [
      # extent (dim 0 first), dtype, lo, hi, row0
      ((5000,), np.float32, (0,), (5000,), 0),
      ((3000000,), np.uint8, (17,), (2999990,), 5),         # one long line
      ((700, 900), np.float32, (0, 0), (700, 900), 0),       # > 2 MiB: the pool
      ((700, 900), np.float32, (3, 100), (690, 870), 100),
      ((64, 48, 96), np.uint16, (1, 2, 10), (60, 40, 90), 7),
      ((1024, 32, 40), np.float64, (0, 0, 0), (1024, 32, 40), 0),
      ((16, 8, 6, 5), np.int16, (1, 1, 1, 1), (15, 7, 5, 4), 1),
  ]
  for extent, dt, lo, hi, row0 in cases:
    shape = extent[::-1]
    dim = len(extent)
    for layout in ('dense', 'padded', 'every-other'):
      if layout == 'dense':
        backing = rng.integers(0, 250, shape).astype(dt)
        arr = backing
      elif layout == 'padded':            # rows with a tail, stride[0] = 1
        backing = rng.integers(0, 250, shape[:-1] + (shape[-1] + 13,)).astype(dt)
        arr = backing[..., :shape[-1]]
      else:                               # stride[0] = 2
        backing = rng.integers(0, 250, shape[:-1] + (2 * shape[-1],)).astype(dt)
        arr = backing[..., ::2]
      rows = extent[-1] - row0
      box = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      # pack: strided -> dense (rows from row0 on)
      stage = np.full((rows,) + shape[1:], 251, dtype=dt)
      _copy_box(lib, arr, stage, extent, lo, hi, True, row0, threads)
      want = np.full_like(stage, 251)
      sbox = (slice(lo[-1] - row0, hi[-1] - row0),) + box[1:]
      want[sbox] = arr[box]
      assert np.array_equal(stage, want), (extent, layout, 'pack')
      # unpack: dense -> strided, nothing outside the box written
      src = rng.integers(0, 250, stage.shape).astype(dt)
      before = backing.copy()
      _copy_box(lib, arr, src, extent, lo, hi, False, row0, threads)
      expect = before.copy()
      view = expect if layout == 'dense' else (
          expect[..., :shape[-1]] if layout == 'padded' else expect[..., ::2])
      view[box] = src[sbox]
      assert np.array_equal(backing, expect), (extent, layout, 'unpack')


WIRE_HOSTS = [
    # main, program (None: blur with two banks per tensor), first word of stdout
    ('jacobi2d_wire_main.cpp', 'jacobi2d.soda'),
    ('heat3d_wire_main.cpp', 'heat3d.soda'),
    ('blur_banks_wire_main.cpp', None),
]


def _two_bank_blur(tmp_path):
  text = open(soda_path('blur.soda')).read()
  text = text.replace('input dram 0 uint16', 'input dram 0.1 uint16')
  text = text.replace('output dram 1 uint16', 'output dram 2.3 uint16')
  assert 'dram 0.1' in text and 'dram 2.3' in text
  path = os.path.join(str(tmp_path), 'blur_banks.soda')
  with open(path, 'w') as f:
    f.write(text)
  return path


@pytest.mark.parametrize('main,soda', WIRE_HOSTS)
def test_independent_wire_hosts_against_the_kernel_contract(tmp_path, main,
                                                            soda):
  """The three reference-style callers (tests/host/frt_host.h: the reference
  host's sizes, scatter, call and gather transcribed from its text, no import
  of anything of this repository) against a plain CPU statement of the kernel
  contract (tests/host/cpu_stream_kernels.cpp), under ASan + UBSan: their
  layout arithmetic and closed-form expectations hold before a GPU is asked
  -- four tiles of jacobi2d as shipped, 2 x 2 tiles of heat3d, two banks per
  tensor."""
  exe = os.path.join(str(tmp_path), 'host_cpu')
  host = os.path.join(ROOT, 'tests', 'host')
  subprocess.run(['g++', '-std=c++17', '-O1', '-Wall', '-Werror',
                  '-fsanitize=address,undefined', os.path.join(host, main),
                  os.path.join(host, 'cpu_stream_kernels.cpp'), '-o', exe],
                 check=True)
  run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert run.returncode == 0 and run.stdout.startswith('OK'), (
      run.stdout + run.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize('main,soda', WIRE_HOSTS)
def test_wire_kernels_under_independent_hosts(built, tmp_path, main, soda):
  """`sodac --hip-wire-kernel` definitions of
  jacobi2d_kernel (iterate 2, tile 32, four tiles), heat3d_kernel (3-D tiles)
  and a two-bank blur_kernel, each linked under a caller written from the
  reference host's text alone and checked against a closed form -- no Python
  in the process, no layout constant shared with soda_amd.stream."""
  path = _two_bank_blur(tmp_path) if soda is None else soda_path(soda)
  src = os.path.join(str(tmp_path), 'wire.cpp')
  subprocess.run([sys.executable, '-m', 'soda_amd.sodac', path,
                  '--hip-wire-kernel', src], cwd=ROOT, check=True)
  if soda is None:
    assert ('void* bank_0_blur_y, void* bank_1_blur_y, void* bank_0_input, '
            'void* bank_1_input, uint64_t coalesced_data_num') in \
        open(src).read()
  exe = os.path.join(str(tmp_path), 'wire_host')
  libdir = os.path.join(ROOT, 'soda_amd')
  subprocess.run(['g++', '-std=c++17', '-O1', '-DSODA_CPP_BINDING', src,
                  os.path.join(ROOT, 'tests', 'host', main),
                  '-I', os.path.join(ROOT, 'include'), '-L', libdir,
                  '-lsoda_hip', '-Wl,-rpath,' + libdir, '-o', exe], check=True)
  env = dict(os.environ)
  env['LD_LIBRARY_PATH'] = '/opt/rocm/lib:' + env.get('LD_LIBRARY_PATH', '')
  # the banks travel through the pinned staging ring (soda_host.cpp ring_send /
  # ring_fetch): once whole (one slot per bank, re-used bank after bank -- the
  # stress run of round 5 caught the second bank overwriting the first one's
  # slot while its DMA was in flight), once in 16 KiB chunks (all four slots)
  for chunk_kb in (None, '16'):
    env.pop('SODA_HIP_HOST_CHUNK_KB', None)
    if chunk_kb:
      env['SODA_HIP_HOST_CHUNK_KB'] = chunk_kb
    run = subprocess.run([exe], capture_output=True, text=True, env=env,
                         timeout=300)
    assert run.returncode == 0 and run.stdout.startswith('OK'), (
        chunk_kb, run.stdout + run.stderr)


@pytest.mark.parametrize('threads', [1, 0])
def test_pack_and_unpack_of_banked_streams(built, threads):
  """soda_hip_host_weave_banks: stream element k lives in bank k % nb at index
  k / nb (reference docs/data-layout.md "Multi-Bank", frt/host.py:241-246,
  422-424); a run of the stream <-> a dense staging run, both ways, on the
  calling thread and on the pool, against numpy's own strided views; what lies
  outside the run keeps its value."""
  import ctypes
  import numpy as np
  from soda_amd import runtime
  lib = runtime.library()
  rng = np.random.default_rng(17)
  for dt, nb, groups, first_g, count_g in (
      (np.uint16, 4, 1 << 20, 0, 1 << 20),       # 8 MiB: the pool
      (np.float32, 2, 700001, 100, 600000),
      (np.uint8, 3, 50000, 7, 40000),
      (np.float64, 8, 3000, 0, 3000),
      (np.int16, 1, 4096, 16, 4000)):
    stream = rng.integers(0, 250, groups * nb).astype(dt)
    banks = [np.ascontiguousarray(stream[b::nb]) for b in range(nb)]
    ptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in banks])
    first, count = first_g * nb, count_g * nb
    dense = np.full(count + 8, 251, dt)
    assert lib.soda_hip_host_weave_banks(
        ptrs, nb, ctypes.c_void_p(dense.ctypes.data), first, count,
        dense.itemsize, 1, threads) == 0
    assert np.array_equal(dense[:count], stream[first:first + count])
    assert (dense[count:] == 251).all()
    # and back into fresh banks
    back = [np.full_like(b, 252) for b in banks]
    bptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in back])
    assert lib.soda_hip_host_weave_banks(
        bptrs, nb, ctypes.c_void_p(dense.ctypes.data), first, count,
        dense.itemsize, 0, threads) == 0
    for b in range(nb):
      want = np.full_like(banks[b], 252)
      want[first_g:first_g + count_g] = banks[b][first_g:first_g + count_g]
      assert np.array_equal(back[b], want), (dt, nb, b)
  # runs that do not start or end on a whole group are refused
  one = (ctypes.c_void_p * 2)(banks[0].ctypes.data, banks[0].ctypes.data)
  assert lib.soda_hip_host_weave_banks(one, 2, ctypes.c_void_p(
      dense.ctypes.data), 1, 4, 2, 1, 1) != 0


def test_pack_and_unpack_from_several_caller_threads(built):
  """The worker pool serves one job at a time; callers from several threads
  take turns (CopyPool::run's `turn_` mutex).  Four Python threads pack big
  arrays at once (ctypes drops the GIL inside the call): every result right,
  no deadlock.  tools/sanitize.sh runs this under TSan."""
  import threading
  import numpy as np
  from soda_amd import runtime
  lib = runtime.library()
  rng = np.random.default_rng(3)
  extent = (1500, 1100)
  arrays = [rng.integers(0, 250, extent[::-1]).astype(np.float32)
            for _ in range(4)]
  stages = [np.zeros_like(a) for a in arrays]
  errors = []

  def work(i):
    try:
      for _ in range(5):
        stages[i][:] = 0
        _copy_box(lib, arrays[i], stages[i], extent, (0, 0), extent, True, 0, 0)
        if not np.array_equal(stages[i], arrays[i]):
          errors.append(i)
    except Exception as e:     # noqa: BLE001
      errors.append(repr(e))

  threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=120)
  assert not any(t.is_alive() for t in threads), 'a caller never returned'
  assert not errors, errors


def test_numpy_host_against_the_transcribed_host(tmp_path):
  """oracle/frt_layout.py (scatter, gather: what every wire case on the GPU is
  driven by) against tests/host/frt_host.h (the reference host transcribed to
  C++ from the same text, written separately) on random layouts: one to
  several tiles per dimension, ragged last tiles, one to four banks a side,
  2-D and 3-D, windows that are not centred.  The kernel in between only
  records -- input banks as laid out, output banks filled with their own
  stream positions -- so every element of every bank and every gathered cell
  is compared (cells the host never writes included).  Under ASan + UBSan."""
  import numpy as np
  from oracle import frt_layout
  from soda_amd import core, stream, util
  exe = os.path.join(str(tmp_path), 'frt_dump')
  host = os.path.join(ROOT, 'tests', 'host')
  subprocess.run(['g++', '-std=c++17', '-O1', '-Wall', '-Werror',
                  '-fsanitize=address,undefined',
                  os.path.join(host, 'frt_dump_main.cpp'), '-o', exe],
                 check=True)
  rng = np.random.default_rng(77)
  ran = multi_tile = banked = gathered = 0
  for trial in range(40):
    dim = 2 if trial % 3 else 3
    tile = [int(rng.choice([8, 16, 32])) for _ in range(dim - 1)]
    taps = {tuple(int(rng.integers(-2, 3)) for _ in range(dim))
            for _ in range(int(rng.integers(1, 5)))}
    taps.add((0,) * dim)
    nb_in = int(rng.integers(1, 5))
    # (banks a side differ in a quarter of the trials: the reference host sizes
    # an output tile in cycles of the INPUT's elements per cycle,
    # frt/host.py:140-145, and then gathers past its own buffer -- the product
    # refuses such programs; only the scatter is compared)
    nb_out = nb_in if trial % 4 else int(rng.integers(1, 5))
    text = ('kernel: lay%d\nburst width: 64\nunroll factor: 2\niterate: 1\n'
            'input dram %s int32: i(%s, *)\n'
            'output dram %s int32: o(%s) = %s\n' % (
                trial, '.'.join(map(str, range(nb_in))),
                ', '.join(map(str, tile)),
                '.'.join(map(str, range(nb_out))), ', '.join(['0'] * dim),
                ' + '.join('i(%s)' % ', '.join(map(str, t))
                           for t in sorted(taps))))
    st = core.from_text(text)
    extent = [int(rng.integers(t // 2 + 3, 3 * t)) for t in tile] + \
        [int(rng.integers(5, 12))]
    try:
      lay = stream.WireLayout(st, extent)
    except util.SodaError:
      continue
    window = st.stencil_window
    sdim = core.get_stencil_dim(window)
    woff = core.get_stencil_window_offset(window)
    argv = [exe, str(dim), '64', str(nb_in), str(nb_out),
            str(st.stencil_distance)]
    for d in range(dim):
      argv += [str(extent[d]), str(tile[d] if d < dim - 1 else 0),
               str(sdim[d]), str(woff[d])]
    run = subprocess.run(argv, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (text, extent, run.stdout[-300:], run.stderr)
    lines = run.stdout.split('\n')
    assert int(lines[0].split()[1]) == lay.cycle_count, (text, extent)
    cells = int(np.prod(extent))
    field = (np.arange(cells, dtype=np.int64) * 7 + 3).astype(np.int32)
    banks = frt_layout.scatter(lay, {'i': field.reshape(extent[::-1])})
    for b in range(nb_in):
      got = np.array(lines[1 + b].split()[2:], dtype=np.int64)
      assert np.array_equal(got, banks['i'][b].astype(np.int64)), (
          text, extent, 'input bank %d' % b)
    ran += 1
    multi_tile += lay.tiles > 1
    banked += nb_in > 1
    if nb_out != nb_in:
      continue
    out_banks = frt_layout.alloc(lay, ['o'])
    for b in range(nb_out):
      n = len(out_banks['o'][b])
      out_banks['o'][b][:] = 1000000 + np.arange(n) * nb_out + b
    mine = {'o': np.full(extent[::-1], -1, np.int32)}
    frt_layout.gather(lay, out_banks, mine)
    theirs = np.array(lines[1 + nb_in].split()[1:], dtype=np.int64)
    assert np.array_equal(theirs, mine['o'].reshape(-1).astype(np.int64)), (
        text, extent, 'gathered array')
    gathered += 1
  assert gathered >= 18 and ran >= 25 and multi_tile >= 8 and banked >= 15, (ran, multi_tile, banked)

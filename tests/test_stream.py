"""Wire-format logic on the CPU: layout numbers, linearisation,
and the whole scatter -> 1-D stencil -> shift -> gather chain with the numpy
oracle standing in for the GPU kernel (the GPU run is in test_hip_parity.py)."""
import numpy as np
import pytest

from conftest import soda_path
from soda_amd import core, stream


def test_layout_numbers_match_the_reference_formulas():
  blur = core.from_file(soda_path('blur.soda'))
  lay = stream.WireLayout(blur, (2000, 1024))
  # burst 256 bit / uint16 x 1 bank = 16 elements per cycle (frt/host.py:120)
  assert lay.epc == {'input': 16, 'blur_y': 16}
  assert lay.tile_count == [1] and lay.stencil_distance == 4002
  assert lay.elem_count_per_tile == 2000 * 1024
  assert lay.cycle_count == -(-(2000 * 1024 + 4002) // 16)
  assert lay.stencil_offset == {'blur_y': 4002}
  assert lay.buf_elems['input'] == 2000 * 1024 + 4016      # round_up(4002, 16)
  j = core.from_file(soda_path('jacobi2d.soda'))            # iterate 2, tile 32
  lay = stream.WireLayout(j, (32, 6))
  assert lay.stencil_distance == 130 and lay.stencil_offset == {'t0': 64}
  wide = stream.WireLayout(j, (100, 6))                     # > one tile wide
  assert wide.tile_count == [(100 - 5) // (32 - 5 + 1) + 1]


def test_linearize():
  j = core.from_file(soda_path('jacobi2d.soda'))
  flat = stream.linearize(j)
  assert flat.dim == 1 and flat.iterate == 2
  assert str(flat.output_stmts[0]) == (
      'output dram 1 float: t0(0) = (t1(32) + t1(1) + t1(0) + t1(-32) + '
      't1(-1)) * 0.2f')
  h = stream.linearize(core.from_file(soda_path('heat3d.soda')))
  assert 'in(1024)' in str(h.output_stmts[0]) and 'in(-32)' in str(
      h.output_stmts[0])


@pytest.mark.parametrize('name,extent', [('blur.soda', (2000, 12)),
                                         ('jacobi2d.soda', (32, 12)),
                                         ('jacobi2d.soda', (20, 9)),
                                         ('heat3d.soda', (32, 32, 9)),
                                         ('sobel2d.soda', (32, 8))])
def test_wire_chain_with_cpu_kernel(name, extent):
  from oracle import frt_layout, numpy_oracle
  st = core.from_file(soda_path(name))
  lay = stream.WireLayout(st, extent)
  rng = np.random.default_rng(5)
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    shape = tuple(extent[::-1])
    ins[n] = (rng.random(shape).astype(t.np_name) if t.is_float else
              rng.integers(-100, 100, shape).astype(t.np_name))
  banks = frt_layout.scatter(lay, ins)
  n = lay.cycle_count * lay.epc[st.input_names[0]]
  streams = {}
  for nme in st.input_names:
    nb = lay.bank_count[nme]
    s = np.zeros(n, banks[nme][0].dtype)
    for b in range(nb):
      s[b::nb] = banks[nme][b][:len(s[b::nb])]
    streams[nme] = s
  out1d = numpy_oracle.run(stream.linearize(st), streams)
  out_banks = frt_layout.alloc(lay, st.output_names)
  for o in st.output_names:
    off, nb = lay.stencil_offset[o], lay.bank_count[o]
    wire = np.zeros(n, out1d[o].dtype)
    wire[off:] = out1d[o][:n - off]
    for b in range(nb):
      out_banks[o][b][:len(wire[b::nb])] = wire[b::nb]
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, out_banks, got)
  want = numpy_oracle.run(st, ins)
  for o in st.output_names:
    assert np.array_equal(got[o], want[o])


@pytest.mark.parametrize('name,extent', [('blur.soda', (2000, 12)),
                                         ('blur.soda', (4500, 9)),
                                         ('jacobi2d.soda', (32, 12)),
                                         ('jacobi2d.soda', (100, 9)),
                                         ('heat3d.soda', (70, 40, 7)),
                                         ('sobel2d.soda', (32, 8))])
def test_wire_chain_as_dense_view(name, extent):
  """The property StreamProgram's `dense` mode rests on: when a tile's row block
  is a whole number of bursts, the stream IS a dense (tile..., rows) array and
  the original n-D program run on that view yields the same valid cells as the
  causal 1-D form."""
  from oracle import frt_layout, numpy_oracle
  st = core.from_file(soda_path(name))
  lay = stream.WireLayout(st, extent)
  rng = np.random.default_rng(6)
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    shape = tuple(extent[::-1])
    ins[n] = (rng.random(shape).astype(t.np_name) if t.is_float else
              rng.integers(-100, 100, shape).astype(t.np_name))
  banks = frt_layout.scatter(lay, ins)
  n = lay.cycle_count * lay.epc[st.input_names[0]]
  block = int(np.prod(st.tile_size[:-1]))
  assert block % lay.epc[st.input_names[0]] == 0
  assert st.stencil_distance >= block
  rows = n // block
  view = tuple(st.tile_size[:-1]) + (rows,)
  dense_in = {}
  for nme in st.input_names:
    nb = lay.bank_count[nme]
    s = np.zeros(n, banks[nme][0].dtype)
    for b in range(nb):
      s[b::nb] = banks[nme][b][:len(s[b::nb])]
    dense_in[nme] = s[:rows * block].reshape(view[::-1])
  out_nd = numpy_oracle.run(st, dense_in)
  out_banks = frt_layout.alloc(lay, st.output_names)
  for o in st.output_names:
    off, nb = lay.stencil_offset[o], lay.bank_count[o]
    flat = np.zeros(n, out_nd[o].dtype)
    flat[:rows * block] = out_nd[o].reshape(-1)
    wire = np.zeros(n, flat.dtype)
    wire[off:] = flat[:n - off]
    for b in range(nb):
      out_banks[o][b][:len(wire[b::nb])] = wire[b::nb]
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, out_banks, got)
  # the kernel's own contract: every cell the host reads back equals the
  # causal 1-D form's (several tiles included)
  ref = {o: np.zeros_like(got[o]) for o in got}
  frt_layout.gather(lay, frt_layout.kernel_on_streams(lay, banks), ref)
  for o in st.output_names:
    assert np.array_equal(got[o], ref[o])
  if lay.tiles == 1:
    want = numpy_oracle.run(st, ins)
    for o in st.output_names:
      assert np.array_equal(got[o], want[o])


@pytest.mark.parametrize('name,extent,iterate', [
    ('blur.soda', (2000, 12), None), ('blur.soda', (4500, 9), None),
    ('blur.soda', (2000, 14), 3),                   # the field moves 3 x (2, 2)
    ('jacobi2d.soda', (32, 12), None), ('jacobi2d.soda', (100, 9), None),
    ('jacobi2d.soda', (32, 20), 5),
    ('heat3d.soda', (70, 40, 7), None), ('sobel2d.soda', (32, 8), None),
    ('denoise2d.soda', (32, 14), None),             # two inputs, delayed
    ('coupled2d.soda', (32, 11), 1),                # two outputs, own offsets
    ('coupled2d.soda', (32, 14), None),             # both circulate, same (0, 1)
])
def test_outputs_born_at_their_wire_positions(name, extent, iterate):
  """`emit_late`: the program with every output's store index moved by the
  window point of largest linear offset leaves, run PLAINLY (no shift, no
  copy) as the causal 1-D form and as the n-D program on the dense view,
  every cell the host reads back where the kernel contract wants it."""
  from oracle import frt_layout, numpy_oracle
  st = core.from_file(soda_path(name), iterate=iterate)
  late = stream.emit_late(st)
  assert late is not None
  tile = st.tile_size
  for s0, s1 in zip(st.output_stmts, late.output_stmts):
    c = tuple(b - a for a, b in zip(s0.ref.idx, s1.ref.idx))
    assert c in st.stencil_window_points(s0.name, iterate=1)
    from soda_amd import util
    assert util.serialize(c, tile) * st.iterate == stream.stencil_offsets(st)[
        s0.name]
  lay = stream.WireLayout(st, extent)
  rng = np.random.default_rng(16)
  ins = {}
  for n, t in zip(st.input_names, st.input_types):
    shape = tuple(extent[::-1])
    ins[n] = (rng.random(shape).astype(t.np_name) if t.is_float else
              rng.integers(-100, 100, shape).astype(t.np_name))
  banks = frt_layout.scatter(lay, ins)
  n = lay.cycle_count * lay.epc[st.input_names[0]]
  streams = {}
  for nme in st.input_names:
    nb = lay.bank_count[nme]
    s = np.zeros(n, banks[nme][0].dtype)
    for b in range(nb):
      s[b::nb] = banks[nme][b][:len(s[b::nb])]
    po = st.produce_offsets()[nme] if len(st.input_names) > 1 else 0
    if po:
      s = np.concatenate([s[po:], np.zeros(po, s.dtype)])
    streams[nme] = s
  ref = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, frt_layout.kernel_on_streams(lay, banks), ref)
  boxes = [st.valid_box(extent, o) for o in st.output_names]
  lo = [max(b[0][d] for b in boxes) for d in range(st.dim)]
  hi = [min(b[1][d] for b in boxes) for d in range(st.dim)]
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))

  def check(out_streams):
    out_banks = frt_layout.alloc(lay, st.output_names)
    for o in st.output_names:
      nb = lay.bank_count[o]
      for b in range(nb):
        out_banks[o][b][:len(out_streams[o][b::nb])] = out_streams[o][b::nb]
    got = {o: np.zeros_like(ref[o]) for o in ref}
    frt_layout.gather(lay, out_banks, got)
    for o in st.output_names:
      assert ref[o][idx].any()
      if len(st.output_names) == 1:
        assert np.array_equal(got[o], ref[o]), o
      else:
        assert np.array_equal(got[o][idx], ref[o][idx]), o

  # the causal 1-D form
  check(numpy_oracle.run(stream.linearize(late), streams))
  # the n-D program on the dense view
  block = int(np.prod(tile[:-1]))
  if block % lay.epc[st.input_names[0]] == 0 and st.stencil_distance >= block:
    rows = n // block
    view = tuple(tile[:-1]) + (rows,)
    out_nd = numpy_oracle.run(
        late, {k: v[:rows * block].reshape(view[::-1])
               for k, v in streams.items()})
    flat = {}
    for o in st.output_names:
      flat[o] = np.zeros(n, out_nd[o].dtype)
      flat[o][:rows * block] = out_nd[o].reshape(-1)
    check(flat)
  else:
    assert name == 'never', 'every case here has a dense view'


def test_circulating_tensors_that_move_apart_keep_the_copy_pass():
  """Two tensors fed back with different late vectors: after one iteration
  the next one's inputs would sit at different displacements."""
  st = core.from_text(
      'kernel: k\nburst width: 64\nunroll factor: 2\niterate: 2\n'
      'input float: a(32, *)\ninput float: b(32, *)\n'
      'output float: a2(0, 0) = a(0, 1) + b(0, 0)\n'
      'output float: b2(0, 0) = b(1, 0) + a(0, 0)\n')
  assert stream.emit_late(st) is None
  same = core.from_file(soda_path('coupled2d.soda'))
  assert same.iterate > 1 and len(same.output_names) == 2
  assert stream.emit_late(same) is not None


@pytest.mark.parametrize('name,extent', [('denoise2d.soda', (32, 14)),
                                         ('coupled2d.soda', (32, 11))])
def test_wire_chain_with_several_inputs(name, extent):
  """Programs with more than one input: the reference host delays every input
  stream by its `produce_offset` (frt/host.py:241-246; the offsets solve the
  integer program of core.py:371-426, restated in Stencil.produce_offsets).
  The kernel contract on such streams -- undo the delay, run the causal 1-D
  program, emit each output late by its stencil offset -- gives the caller the
  n-D result on the valid box."""
  from oracle import frt_layout, numpy_oracle
  st = core.from_file(soda_path(name))
  delays = st.produce_offsets()
  assert min(delays.values()) == 0 and set(delays) == set(st.input_names)
  if name == 'denoise2d.soda':
    # f is read at (0, 0) only, together with g(0, +-1) = func(u(., +-2)): it
    # is needed two 32-cell rows later than u
    assert delays == {'f': 64, 'u': 0}
  lay = stream.WireLayout(st, extent)
  rng = np.random.default_rng(8)
  ins = {n: rng.random(tuple(extent[::-1])).astype(t.np_name)
         for n, t in zip(st.input_names, st.input_types)}
  banks = frt_layout.scatter(lay, ins)
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, frt_layout.kernel_on_streams(lay, banks), got)
  want = numpy_oracle.run(st, ins)
  # the host gathers every output over ONE region, that of the program's
  # stencil window (frt/host.py:357-375): compare there
  boxes = [st.valid_box(extent, o) for o in st.output_names]
  lo = [max(b[0][d] for b in boxes) for d in range(st.dim)]
  hi = [min(b[1][d] for b in boxes) for d in range(st.dim)]
  idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
  for o in st.output_names:
    assert got[o][idx].any()
    assert np.array_equal(got[o][idx], want[o][idx])


def test_outputs_without_a_window_are_refused():
  """An output that reads no input (a constant) has no stencil window: the
  reference host would take max() of an empty set; here a SemanticError says
  so (found by tools/fuzz_scan.py wire as an IndexError / ValueError)."""
  from soda_amd import core, stream, util
  head = 'kernel: k\nburst width: 64\nunroll factor: 2\niterate: 1\n' \
         'input float: a(32, *)\n'
  for body in ('output float: b(0, 0) = 0.5f\n',
               'output float: b(0, 0) = a(0, 1)\noutput float: c(0, 0) = 2.0f\n'):
    st = core.from_text(head + body)
    with pytest.raises(util.SemanticError, match='no input'):
      stream.WireLayout(st, (28, 20))


def test_windows_wider_than_a_tile_are_refused():
  from soda_amd import core, stream, util
  taps = ' + '.join('a(%d, 0)' % i for i in range(-16, 17))
  st = core.from_text('kernel: k\nburst width: 64\nunroll factor: 2\n'
                      'iterate: 1\ninput float: a(32, *)\n'
                      'output float: b(0, 0) = %s\n' % taps)
  with pytest.raises(util.SemanticError, match='more than a tile'):
    stream.WireLayout(st, (100, 20))


def test_outputs_that_only_look_back_are_refused():
  """Every tap behind the cell: the stencil offset (largest linear offset of
  the window) is negative and the reference host would gather the output from
  in front of its buffer (frt/host.py:401-424).  Refused by name (the scan met
  it as a shape error inside the host restatement)."""
  from soda_amd import core, stream, util
  st = core.from_text('kernel: k\nburst width: 64\nunroll factor: 2\n'
                      'iterate: 1\ninput float: a(32, *)\n'
                      'output float: b(0, 0) = a(-1, 0) + a(0, -1)\n')
  with pytest.raises(util.SemanticError, match='behind'):
    stream.WireLayout(st, (30, 20))
  with pytest.raises(util.SemanticError, match='behind'):
    stream.stream_specs(st)


def test_cells_no_kernel_can_be_held_to():
  """Two outputs with different windows on several tiles: the host gathers
  both over the region of the program's window (frt/host.py:357-375), so the
  output with the wider window of its own is read at tile-edge cells where the
  1-D form wraps into the next row and the dense form reads outside the array.
  The two restatements of the contract differ exactly there
  (tools/fuzz_scan.py wire, seed 333)."""
  from oracle import frt_layout
  st = core.from_text(
      'kernel: k\nburst width: 64\nunroll factor: 2\niterate: 1\n'
      'input int32: a(32, *)\n'
      'output int32: p(0, 0) = a(-2, -2) + a(1, 2)\n'
      'output int32: q(0, 0) = a(1, -1) + a(2, 2)\n')
  extent = (122, 27)
  lay = stream.WireLayout(st, extent)
  assert lay.tiles == 5
  rng = np.random.default_rng(2)
  banks = frt_layout.scatter(
      lay, {'a': rng.integers(-99, 99, extent[::-1]).astype(np.int32)})
  refs = []
  for banks_out in (frt_layout.kernel_on_streams(lay, banks),
                    frt_layout.kernel_on_dense_view(lay, banks)):
    got = {o: np.zeros(extent[::-1], np.int32) for o in st.output_names}
    frt_layout.gather(lay, banks_out, got)
    refs.append(got)
  assert np.array_equal(refs[0]['p'], refs[1]['p'])
  cols = sorted(set(np.argwhere(refs[0]['q'] != refs[1]['q'])[:, 1].tolist()))
  # in-tile column 30 of every tile (tiles step by 32 - 4 + 1 = 29 columns)
  assert cols == [30, 59, 88, 117]


def test_delays_are_right_only_on_arrays_one_tile_wide():
  """The reference host delays an input in the CALLER's coordinates
  (data[max(0, original_offset - produce_offset)], frt/host.py:241-246) by an
  offset that counts stream elements, i.e. TILE rows (core.py:371-426).  On an
  array exactly one tile wide the kernel contract gives the caller the n-D
  result (test_wire_chain_with_several_inputs); on a narrower one the delayed
  input arrives misplaced and it does not -- an upstream defect this backend
  reproduces rather than repairs (tools/fuzz_scan.py wire,
  seeds 1159 / 1165 / 1419)."""
  from oracle import frt_layout, numpy_oracle
  st = core.from_file(soda_path('denoise2d.soda'))
  assert st.produce_offsets() == {'f': 64, 'u': 0} and st.tile_size[0] == 32
  agree = {}
  for extent in ((32, 14), (28, 14)):
    lay = stream.WireLayout(st, extent)
    assert lay.tiles == 1
    rng = np.random.default_rng(8)
    ins = {n: rng.random(tuple(extent[::-1])).astype(t.np_name)
           for n, t in zip(st.input_names, st.input_types)}
    got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
           for o, t in zip(st.output_names, st.output_types)}
    frt_layout.gather(lay, frt_layout.kernel_on_streams(
        lay, frt_layout.scatter(lay, ins)), got)
    want = numpy_oracle.run(st, ins)
    lo, hi = st.valid_box(extent)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    o = st.output_names[0]
    agree[extent[0]] = np.array_equal(got[o][idx], want[o][idx])
  assert agree == {32: True, 28: False}


def test_late_programs_of_random_programs():
  """emit_late over the random programs of tests/fuzz.py (several inputs,
  outputs, locals, rich expressions, iterations): wherever it yields a
  program, that program run plainly -- as the causal 1-D form and, where the
  stream has one, on the dense view -- leaves every cell the host can hold the
  kernel to where the contract wants it.  The numpy oracle is the kernel; the
  GPU runs are in tests/test_hip_parity.py and tools/fuzz_scan.py wire."""
  import fuzz
  from oracle import frt_layout, numpy_oracle
  from soda_amd import util
  ran = dense_ran = 0
  for seed in range(3000, 3120):
    rng = np.random.default_rng(seed + 91000)
    text, dim, _ = fuzz.program(seed, rich=bool(seed % 3 == 0))
    if dim != 2:
      continue
    try:
      st = core.from_text(text)
      late = stream.emit_late(st)
    except util.SodaError:
      continue
    if late is None:
      continue
    extent = (int(rng.integers(20, 33)) if seed % 4 else 32,
              int(rng.integers(12, 30)))
    boxes = [st.valid_box(extent, o) for o in st.output_names]
    lo = [max(b[0][d] for b in boxes) for d in range(2)]
    hi = [min(b[1][d] for b in boxes) for d in range(2)]
    if not all(h > l for l, h in zip(lo, hi)):
      continue
    try:
      lay = stream.WireLayout(st, extent)
      ins = fuzz.inputs_for(st, extent, seed)
      banks = frt_layout.scatter(lay, ins)
      ref_banks = frt_layout.kernel_on_streams(lay, banks)
    except (util.SodaError, ValueError, IndexError):
      continue                 # (layouts the reference host itself cannot hold)
    n = lay.cycle_count * lay.epc[st.input_names[0]]
    streams = {}
    for name in st.input_names:
      s = np.zeros(n, banks[name][0].dtype)
      s[:] = banks[name][0][:n]
      po = st.produce_offsets()[name] if len(st.input_names) > 1 else 0
      if po:
        s = np.concatenate([s[po:], np.zeros(po, s.dtype)])
      streams[name] = s
    ref = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
           for o, t in zip(st.output_names, st.output_types)}
    frt_layout.gather(lay, ref_banks, ref)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    # cells the two readings of the contract disagree on are nobody's
    held = {o: np.ones(ref[o][idx].shape, bool) for o in ref}
    other = frt_layout.kernel_on_dense_view(lay, banks)
    if other is not None and len(st.output_names) > 1:
      ref2 = {o: np.zeros_like(ref[o]) for o in ref}
      frt_layout.gather(lay, other, ref2)
      held = {o: ref[o][idx] == ref2[o][idx] for o in ref}

    def check(out_streams, what):
      out_banks = frt_layout.alloc(lay, st.output_names)
      for o in st.output_names:
        out_banks[o][0][:n] = out_streams[o][:n]
      got = {o: np.zeros_like(ref[o]) for o in ref}
      frt_layout.gather(lay, out_banks, got)
      for o in st.output_names:
        assert np.array_equal(got[o][idx][held[o]], ref[o][idx][held[o]],
                              equal_nan=True), (seed, what, o)

    check(numpy_oracle.run(stream.linearize(late), streams), 'linear')
    ran += 1
    block = st.tile_size[0]
    if other is not None:
      rows = n // block
      out_nd = numpy_oracle.run(
          late, {k: v[:rows * block].reshape((rows, block))
                 for k, v in streams.items()})
      flat = {}
      for o in st.output_names:
        flat[o] = np.zeros(n, out_nd[o].dtype)
        flat[o][:rows * block] = out_nd[o].reshape(-1)
      check(flat, 'dense')
      dense_ran += 1
  assert ran >= 25 and dense_ran >= 15, (ran, dense_ran)


def _wire_cases():
  import importlib.util
  import os
  from conftest import GOLDEN_DIR
  spec = importlib.util.spec_from_file_location(
      'make_wire_golden', os.path.join(GOLDEN_DIR, 'make_wire_golden.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


WIRE_TAGS = ['blur', 'jacobi2d_4tiles', 'heat3d_2x2tiles', 'denoise2d',
             'blur_4banks', 'jacobi2d_2banks']


@pytest.mark.parametrize('tag', WIRE_TAGS)
def test_committed_wire_vectors(tag):
  """tests/golden/wire_*.npz (tests/golden/make_wire_golden.py): the banks the
  reference host's layout puts the caller's arrays in, the banks the kernel
  contract leaves, what the host gathers -- as committed.  Today's layout code
  and oracle reproduce every array bit for bit."""
  import os
  from oracle import frt_layout
  mod = _wire_cases()
  case = [c for c in mod.CASES if c[0] == tag][0]
  from conftest import GOLDEN_DIR
  gold = np.load(os.path.join(GOLDEN_DIR, 'wire_%s.npz' % tag))
  st = mod.program(case)
  extent = tuple(int(x) for x in gold['extent'])
  assert extent == tuple(case[3])
  lay = stream.WireLayout(st, extent)
  assert lay.cycle_count == int(gold['cycle_count'])
  ins = {n: gold['in_' + n] for n in st.input_names}
  in_banks = frt_layout.scatter(lay, ins)
  for n, bs in in_banks.items():
    for b, a in enumerate(bs):
      assert np.array_equal(a, gold['inbank%d_%s' % (b, n)]), (n, b)
  out_banks = frt_layout.kernel_on_streams(lay, in_banks)
  for n, bs in out_banks.items():
    for b, a in enumerate(bs):
      assert np.array_equal(a, gold['outbank%d_%s' % (b, n)]), (n, b)
  got = {o: np.zeros(tuple(extent[::-1]), np.dtype(t.np_name))
         for o, t in zip(st.output_names, st.output_types)}
  frt_layout.gather(lay, out_banks, got)
  for o in st.output_names:
    assert got[o].any() and np.array_equal(got[o], gold['out_' + o]), o


def _same_sizes(lay, want):
  diffs = []
  for key, val in want.items():
    got = getattr(lay, key)
    if isinstance(val, dict):
      got = dict(got)
    elif isinstance(val, list):
      got = list(got)
    if got != val:
      diffs.append((key, got, val))
  return diffs


def test_wire_constants_agree_with_a_second_derivation():
  """oracle/frt_layout.py drives the wire cases with sizes
  taken from the product's WireLayout -- a constant wrong in both would pass.
  frt_layout.sizes() derives every one of them a second time from the parsed
  taps alone (its own window search over the chained iterations, its own
  serialisation), formula by formula from the reference text
  (frt/host.py:105-162, 272-276, 395-403; core.py:616-625, 858-870, 922-926).
  The corpus (the reference's KAT distances 4002 / 130 / 4162 among them),
  banked variants and random programs, several extents each."""
  import glob
  import os
  import re
  import fuzz
  from conftest import SODA_DIR
  from oracle import frt_layout
  from soda_amd import util
  programs = []
  for path in sorted(glob.glob(os.path.join(SODA_DIR, '*.soda'))):
    text = open(path).read()
    programs.append(text)
    # two and four banks per tensor, the reference's `dram 0.1` spelling
    programs.append(re.sub(r'dram (\d+)\b(?!\.)', r'dram \1.\1', text))
  for seed in range(60):
    programs.append(fuzz.program(seed)[0])
    programs.append(fuzz.program(seed, rich=True)[0])
  for seed in range(30):
    programs.append(fuzz.window_program(seed)[0])
  checked = kat = boxes_equal = 0
  for text in programs:
    try:
      st = core.from_text(text)
    except util.SodaError:
      continue
    for scale in (1, 3):
      extent = [t * scale + 5 for t in st.tile_size[:-1]] + [23 * scale]
      try:
        lay = stream.WireLayout(st, extent)
      except util.SodaError:
        continue          # (refusals are tested where they are made)
      want = frt_layout.sizes(st, extent)
      assert not _same_sizes(lay, want), (st.app_name, extent)
      # ... and, from the same second window search, the valid boxes every
      # parity test compares inside (reference frt/host.py:565-577)
      # (the product also keeps every local inside its own box: inside the
      # reference's bounds always, equal where outputs read inputs only)
      direct = not st.local_stmts and st.iterate == 1
      for o, (lo, hi) in frt_layout.valid_boxes(st, extent).items():
        got_lo, got_hi = st.valid_box(extent, o)
        assert all(g >= l for g, l in zip(got_lo, lo)), (st.app_name, o)
        assert all(g <= h for g, h in zip(got_hi, hi)), (st.app_name, o)
        if direct:
          assert (tuple(got_lo), tuple(got_hi)) == (lo, hi), (st.app_name, o)
          boxes_equal += 1
      checked += 1
      if st.app_name in ('blur', 'jacobi2d', 'heat3d') and \
          want['stencil_distance'] in (4002, 130, 4162):
        kat += 1
  assert checked > 150 and kat >= 6 and boxes_equal >= 30, (
      checked, kat, boxes_equal)


def test_distance_of_windows_that_lie_ahead_of_the_cell():
  """reference core.py:616-625: kStencilDistance is max(distance, stencil
  offset).  When every tap lies ahead of the cell in streaming order the bare
  distance is smaller (here 0 against 63, 32 against 63) and a host that sized
  its void tail and cycle count by it would cut the last outputs off.  Found by
  test_wire_constants_agree_with_a_second_derivation; the whole chain --
  scatter, the kernel contract, gather -- then returns the n-D result."""
  from oracle import frt_layout, numpy_oracle
  for expr, want in (('i(-1, 2)', 63), ('i(-1, 2) + i(0, 1)', 63),
                     ('i(1, 0) + i(0, 0)', 1)):
    st = core.from_text('kernel: ahead\nburst width: 64\nunroll factor: 2\n'
                        'iterate: 1\ninput int32: i(32, *)\n'
                        'output int32: o(0, 0) = %s\n' % expr)
    assert st.stencil_distance == want, expr
    bare = core.get_stencil_distance(st.stencil_window, st.tile_size)
    assert bare <= want
    extent = (32, 11)
    lay = stream.WireLayout(st, extent)
    assert not _same_sizes(lay, frt_layout.sizes(st, extent))
    rng = np.random.default_rng(4)
    ins = {'i': rng.integers(-99, 99, extent[::-1]).astype(np.int32)}
    banks = frt_layout.scatter(lay, ins)
    got = {'o': np.zeros(extent[::-1], np.int32)}
    frt_layout.gather(lay, frt_layout.kernel_on_streams(lay, banks), got)
    ref = numpy_oracle.run(st, ins)
    lo, hi = st.valid_box(extent)
    idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
    assert np.array_equal(got['o'][idx], ref['o'][idx]), expr

"""The reference kernel's WIRE format on the GPU: `<app>_kernel` on banked streams.

SURVEY.md section 8(f2).  The reference's generated host does not hand the
kernel an image; it hands it a linear STREAM per tensor: tiles of
`tile_size[0..dim-2] x extent[dim-1]` cells laid end to end (dimension 0
fastest), each tile padded to whole bursts, the elements dealt cyclically over
the tensor's DRAM banks, `kStencilDistance` void elements appended
(reference src/soda/codegen/frt/host.py:124-178 sizes, :181-249 scatter,
docs/data-layout.md).  The kernel is called as

    <app>_kernel(out banks..., in banks..., coalesced_data_num)      (host.py:44-59, 282-289)

and is oblivious of tiles: it is a causal 1-D stencil over the stream.  An
output cell with in-tile linear index L appears at stream position
L + stencil_offset, stencil_offset = the largest linear offset in the overall
stencil window (host.py:400-424), because that is when the line buffer has seen
everything the cell needs.

`StreamProgram` reproduces exactly that contract so the backend can sit behind
the UNMODIFIED generated host: the program is linearised (every tap (i, j, ..)
becomes the 1-D tap serialize((i, j, ..), tile_size), reference
src/soda/util.py:9-12), run by the ordinary kernels on the de-interleaved
stream, and the outputs are written back shifted and re-interleaved.  Cells whose
taps wrap around a tile row are garbage here exactly as they are on the FPGA;
the host never reads them.
"""
import ctypes
from typing import Dict, List, Optional, Sequence

from soda_amd import core, grammar, ir, runtime, util
from soda_amd.codegen.hip import lower


def linearize(stencil: core.Stencil) -> core.Stencil:
  """The 1-D program the FPGA kernel really runs on its input stream."""
  tile = stencil.tile_size
  if any(t <= 0 for t in tile[:-1]):
    raise util.SemanticError('stream mode needs a tile size in every '
                             'dimension but the last')

  def lin(idx):
    return util.serialize(idx, tile)

  def flat(node):
    if isinstance(node, ir.Ref) and node.name not in stencil.param_names:
      return ir.Ref(node.name, (lin(node.idx),), node.lat, node.haoda_type)
    return node

  inputs = [grammar.InputStmt(s.haoda_type, s.name, (), s.dram)
            for s in stencil.input_stmts]

  def conv(stmt, cls):
    return cls(stmt.haoda_type, ir.Ref(stmt.ref.name, (lin(stmt.ref.idx),)),
               stmt.expr.transform(flat),
               [l.transform(flat) for l in stmt.let], getattr(stmt, 'dram', ()))

  return core.Stencil(
      burst_width=stencil.burst_width, border=stencil.border,
      iterate=stencil.iterate, cluster=stencil.cluster,
      app_name=stencil.app_name, input_stmts=inputs, param_stmts=[],
      local_stmts=[conv(s, grammar.LocalStmt) for s in stencil.local_stmts],
      output_stmts=[conv(s, grammar.OutputStmt) for s in stencil.output_stmts],
      dim=1, tile_size=(0,), unroll_factor=stencil.unroll_factor,
      replication_factor=stencil.replication_factor)


def stencil_offsets(stencil: core.Stencil) -> Dict[str, int]:
  """Where an output cell sits in its stream relative to its own in-tile
  linear index: the largest linear offset of the output's overall stencil
  window (reference frt/host.py:401-408)."""
  tile = stencil.tile_size
  out = {}
  for s in stencil.output_stmts:
    pts = stencil.stencil_window_points(s.name)
    if not pts:
      # (a constant: the reference's host would take max() of an empty window)
      raise util.SemanticError(
          'wire format: output `%s` depends on no input, it has no stencil '
          'window' % s.name)
    out[s.name] = core.get_stencil_distance(pts, tile) - util.serialize(
        core.get_stencil_window_offset(pts), tile)
    if out[s.name] < 0:
      # (every tap behind the cell: the reference host would gather such an
      # output from in front of its buffer, host.py:401-424)
      raise util.SemanticError(
          'wire format: output `%s` reads only cells behind it (stencil '
          'offset %d < 0)' % (s.name, out[s.name]))
  return out


def emit_late(stencil: core.Stencil) -> Optional[core.Stencil]:
  """The program whose outputs are BORN at their wire positions, or None.

  The kernel owes output cell L at stream position L + stencil_offset
  (frt/host.py:401-408).  serialize() is linear, so storing the cell c further
  on in n-D -- `out(s + c) = expr` instead of `out(s) = expr`, c the point of
  the output's one-iteration window with the largest linear offset -- puts it
  exactly there, in the linear form and on the dense (tile..., rows) view
  alike, and the shift + copy pass over every output goes away.  A cell the
  host reads has its whole window inside its tile, c is a point of that window,
  so its new position is inside the tile too.

  Iterated programs feed every output back as an input: each iteration then
  moves the field by c, `iterate` x c in all -- which must be the stencil offset
  of the iterated program (it is while one tensor circulates: the largest
  linear offset of a Minkowski sum is the sum of the largest offsets) and must
  be the SAME c for every circulating tensor, or the inputs of the next
  iteration would sit at different displacements; programs where it is not
  keep the copy pass."""
  st = stencil
  tile = st.tile_size
  total = stencil_offsets(st)
  late: Dict[str, tuple] = {}
  for s in st.output_stmts:
    pts = st.stencil_window_points(s.name, iterate=1)
    c = max(pts, key=lambda p: util.serialize(p, tile))
    if util.serialize(c, tile) * st.iterate != total[s.name]:
      return None
    late[s.name] = tuple(c)
  if st.iterate > 1 and len(set(late.values())) != 1:
    return None

  def moved(node):
    # a later statement that reads an output reads it where it now lives
    if isinstance(node, ir.Ref) and node.name in late:
      return ir.Ref(node.name,
                    tuple(i + c for i, c in zip(node.idx, late[node.name])),
                    node.lat, node.haoda_type)
    return node

  def conv(stmt, cls):
    idx = stmt.ref.idx
    if stmt.ref.name in late:
      idx = tuple(i + c for i, c in zip(idx, late[stmt.ref.name]))
    return cls(stmt.haoda_type, ir.Ref(stmt.ref.name, idx),
               stmt.expr.transform(moved),
               [l.transform(moved) for l in stmt.let], getattr(stmt, 'dram', ()))

  return core.Stencil(
      burst_width=st.burst_width, border=st.border, iterate=st.iterate,
      cluster=st.cluster, app_name=st.app_name, input_stmts=st.input_stmts,
      param_stmts=[],
      local_stmts=[conv(s, grammar.LocalStmt) for s in st.local_stmts],
      output_stmts=[conv(s, grammar.OutputStmt) for s in st.output_stmts],
      dim=st.dim, tile_size=st.tile_size, unroll_factor=st.unroll_factor,
      replication_factor=st.replication_factor)


class WireLayout:
  """Sizes and offsets of the banked streams, formula for formula as the
  reference host computes them (frt/host.py line numbers in comments)."""

  def __init__(self, stencil: core.Stencil, extent: Sequence[int]):
    st = stencil
    self.stencil = st
    self.extent = tuple(extent)
    dim = st.dim
    table = st.symbol_table
    stmts = st.input_stmts + st.output_stmts
    self.bank_count = {s.name: len(s.dram) for s in stmts}            # :99
    self.epc = {s.name: st.burst_width // table[s.name].width_in_bits *
                self.bank_count[s.name] for s in stmts}                # :120-122
    window = st.stencil_window
    if not window:
      # (the first output reads no input at all -- a constant: the reference's
      # host would take max() of an empty window here)
      raise util.SemanticError(
          'wire format: output `%s` depends on no input, the stencil has no '
          'window' % st.output_names[0])
    self.stencil_dim = core.get_stencil_dim(window)
    self.stencil_distance = st.stencil_distance
    tile = st.tile_size
    for d in range(dim - 1):
      if tile[d] - self.stencil_dim[d] + 1 < 1:
        # (the reference host would divide by zero / a negative step here)
        raise util.SemanticError(
            'wire format: the stencil window spans %d cells of dimension %d, '
            'more than a tile of %d holds' % (self.stencil_dim[d], d, tile[d]))
    self.tile_count = [
        (self.extent[d] - self.stencil_dim[d]) //
        (tile[d] - self.stencil_dim[d] + 1) + 1 for d in range(dim - 1)
    ]                                                                   # :124-128
    self.tiles = 1
    for c in self.tile_count:
      self.tiles *= c
    self.elem_count_per_tile = self.extent[dim - 1]
    for d in range(dim - 1):
      self.elem_count_per_tile *= tile[d]                               # :137-139
    in0, out0 = st.input_names[0], st.output_names[0]
    self.cycle_count_per_tile = -(-self.elem_count_per_tile // self.epc[in0])
    self.aligned_per_tile_i = self.cycle_count_per_tile * self.epc[in0]  # :142
    self.aligned_per_tile_o = self.cycle_count_per_tile * self.epc[out0]  # :144

    def round_up(a, b):
      return -(-a // b) * b

    self.buf_elems = {}
    for s in st.input_stmts:
      self.buf_elems[s.name] = (self.tiles * self.aligned_per_tile_i + round_up(
          self.stencil_distance, self.epc[s.name]))                    # :151-156
    for s in st.output_stmts:
      self.buf_elems[s.name] = (self.tiles * self.aligned_per_tile_o + round_up(
          self.stencil_distance, self.epc[s.name]))                    # :157-162
    self.cycle_count = -(-(self.elem_count_per_tile * self.tiles +
                           self.stencil_distance) // self.epc[in0])    # :272-276
    self.stencil_offset = stencil_offsets(st)                          # :401-408


_WIRE_SRC = """
// bank k %% NB, index k / NB  <->  stream position k   (reference
// docs/data-layout.md "Multi-Bank"; frt/host.py:241-246, 422-424)
extern "C" __global__ void __launch_bounds__(256) %(name)s(soda_hip_kargs_t a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= a.extent[0]) return;
%(body)s
}
"""

# The same with 16 bytes per bank per thread: one vector load (store) per bank,
# the interleave done in registers, NB vector stores (loads) on the dense side.
# Only for whole groups of V x NB elements inside the stream; the last, partial
# group goes element by element.  `lead` = elements the bank pointers are
# advanced by (an input the host delayed by a multiple of V x NB elements).
_WIRE_VEC_SRC = """
extern "C" __global__ void __launch_bounds__(256) %(name)s(soda_hip_kargs_t a) {
  typedef %(ct)s soda_v __attribute__((ext_vector_type(%(V)d)));
  const int64_t n = a.extent[0];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k0 = t * %(group)d;
  if (k0 >= n) return;
  %(dense_decl)s
  uint64_t soda_low = 0;      // the caller's banks: 16-byte aligned, or slow
  for (int b = 0; b < %(NB)d + 1; ++b) soda_low |= (uint64_t)a.buf[b];
  if (k0 + %(group)d + %(lead)d * %(NB)d <= n && (soda_low & 15) == 0) {
%(vec_body)s
  } else {
    const int64_t end = k0 + %(group)d < n ? k0 + %(group)d : n;
    for (int64_t k = k0; k < end; ++k) {
%(tail_body)s
    }
  }
}
"""


def _vec_copy_source(name: str, ct: str, elem: int, nb: int, lead: int,
                     to_dense: bool) -> Optional[str]:
  """Text of the vector form of unwire_<in> (to_dense) / wire_<out>, or None
  where 16 bytes per bank per thread do not divide evenly."""
  if elem not in (1, 2, 4, 8) or nb > 8:
    return None
  V = 16 // elem
  if lead % V:
    return None
  L = []
  if to_dense:
    decl = '%s* __restrict__ dense = (%s*)a.buf[%d];' % (ct, ct, nb)
    for b in range(nb):
      L.append('    const soda_v r%d = *(const soda_v*)((const %s*)a.buf[%d] + '
               't * %d + %d);' % (b, ct, b, V, lead))
    for o in range(nb):
      elems = ['r%d[%d]' % ((o * V + e) % nb, (o * V + e) // nb)
               for e in range(V)]
      L.append('    { soda_v w; %s' % ' '.join(
          'w[%d] = %s;' % (e, x) for e, x in enumerate(elems)))
      L.append('      *(soda_v*)(dense + k0 + %d) = w; }' % (o * V))
    tail = ['      const int64_t src = k + %d;' % (lead * nb),
            '      dense[k] = src < n ? ((const %s*)a.buf[src %% %d])[src / %d]'
            ' : (%s)0;' % (ct, nb, nb, ct)]
  else:
    decl = 'const %s* __restrict__ dense = (const %s*)a.buf[0];' % (ct, ct)
    for o in range(nb):
      L.append('    const soda_v r%d = *(const soda_v*)(dense + k0 + %d);' %
               (o, o * V))
    for b in range(nb):
      elems = ['r%d[%d]' % ((j * nb + b) // V, (j * nb + b) % V)
               for j in range(V)]
      L.append('    { soda_v w; %s' % ' '.join(
          'w[%d] = %s;' % (e, x) for e, x in enumerate(elems)))
      L.append('      *(soda_v*)((%s*)a.buf[%d] + t * %d) = w; }' %
               (ct, 1 + b, V))
    tail = ['      ((%s*)a.buf[1 + k %% %d])[k / %d] = dense[k];' % (ct, nb, nb)]
  return _WIRE_VEC_SRC % dict(name=name, ct=ct, V=V, NB=nb, group=V * nb,
                              lead=lead, dense_decl=decl,
                              vec_body='\n'.join(L),
                              tail_body='\n'.join(tail))


StreamDesc = runtime.StreamDesc


class ProgramSpec:
  """One code object + plan of the wire-format kernel (what the C++ the
  `--hip-wire-kernel` generator prints embeds, and what StreamProgram loads)."""

  def __init__(self, tag: str, source: str, plan: 'runtime.Plan',
               kernel_names: Sequence[str]):
    self.tag, self.source, self.plan = tag, source, plan
    self.kernel_names = list(kernel_names)


def _copy_plan(name: str, n_in: int, n_out: int, elem: int,
               per_block: int = 256) -> 'runtime.Plan':
  plan = runtime.Plan()
  plan.abi_version = runtime.ABI_VERSION
  plan.dim = 1
  plan.num_inputs, plan.num_outputs = n_in, n_out
  for i in range(n_in + n_out):
    plan.elem_size[i] = elem
  plan.num_kernels = 1
  plan.kernels[0].name = name.encode()
  plan.kernels[0].block[0] = 256
  plan.kernels[0].block[1] = plan.kernels[0].block[2] = 1
  plan.kernels[0].tile[0] = per_block       # stream elements per block
  for d in range(1, runtime.MAX_DIM):
    plan.kernels[0].tile[d] = 1
  plan.kernels[0].window_extra = -1
  plan.num_passes = 1
  plan.passes[0].fused_iters = 1
  plan.passes[0].num_kernels = 1
  return plan


def input_shifts(stencil: core.Stencil) -> Dict[str, int]:
  """Elements the reference host delays every input stream by (its
  `produce_offset`, frt/host.py:241-246).  Zero for a single input (the
  reference pins input 0 at 0, core.py:374); for several inputs the offsets
  come out of the reference's ILP (core.py:371-426), restated in
  core.produce_offsets."""
  if len(stencil.input_names) == 1:
    return {stencil.input_names[0]: 0}
  return stencil.produce_offsets()


def stream_specs(stencil: core.Stencil, dense: Optional[bool] = None,
                 direct: bool = True):
  """(StreamDesc, {tag: ProgramSpec}) of `<app>_kernel` for `stencil`.  Tags:
  `dense` (optional), `linear<V>`, `unwire_<input>`, `wire_<output>` (absent
  for an output the program writes in place).  `direct=False`: every output
  through the shift + copy pass, as before round 5."""
  if stencil.param_stmts:
    raise util.SemanticError('stream mode does not support param tensors')
  if stencil.preserve_border:
    raise util.SemanticError(
        'stream mode does not support border: preserve (tile edges are not '
        'grid borders)')
  st = stencil
  table = st.symbol_table
  # outputs stored at their wire positions by the program itself where that is
  # possible (emit_late): no shift left for the copy pass, and no copy pass at
  # all for an output on one bank
  late = emit_late(st) if direct else None
  run = late or st
  flat = linearize(run)
  banks = {s.name: len(s.dram) for s in st.input_stmts + st.output_stmts}
  offsets = stencil_offsets(st)
  shifts = input_shifts(st)
  names = list(st.input_names) + list(st.output_names)
  desc = StreamDesc()
  desc.dim = st.dim
  desc.num_inputs, desc.num_outputs = len(st.input_names), len(st.output_names)
  desc.iterate = st.iterate
  for d in range(st.dim - 1):
    desc.tile[d] = st.tile_size[d]
  desc.stencil_distance = st.stencil_distance if st.dim >= 2 else 0
  for t, n in enumerate(names):
    desc.banks[t] = banks[n]
    desc.elem_size[t] = table[n].size_in_bytes
    desc.elems_per_cycle[t] = st.burst_width // table[n].width_in_bits * banks[n]
    desc.shift[t] = shifts[n] if n in shifts else (0 if late else offsets[n])
  specs: Dict[str, ProgramSpec] = {}

  def program_spec(tag, sten, opts, extent):
    opts = runtime.resolve_options(sten, opts, extent)
    mod = lower.lower(sten, opts)
    res = runtime.kernel_resources(
        runtime.compile_source(mod.source, '%s.hip' % sten.app_name))
    specs[tag] = ProgramSpec(tag, mod.source, runtime.make_plan(mod, res),
                             [k.name for k in mod.kernels])

  # the linearised 1-D program, widest vector first, always ending in 1
  vecs = []
  v = lower.default_vec(flat)
  while v >= 1:
    vecs.append(v)
    v //= 2
  vecs = vecs[:4] if vecs[:4][-1] == 1 else vecs[:3] + [1]
  desc.num_linear = len(vecs)
  for k, v in enumerate(vecs):
    desc.linear_vec[k] = v
    program_spec('linear%d' % v, flat,
                 lower.LowerOptions(strategy='direct', vec=v), None)
  # the original n-D program on the dense view of the stream; narrow tiles
  # leave most of a 64-lane x V-wide marching strip idle and the linear form
  # wins (heat3d 32 x 32 tiles: 0.50 vs 0.79 ms)
  # (None: offered whenever the program is 2-D / 3-D; the library takes it for
  # device-resident banks only from DENSE_MIN_TILE0 cells per tile row on, for
  # host banks always -- soda_hip_stream_set_device_dense_min_tile)
  if dense is None:
    dense = st.dim >= 2
  if dense and st.dim >= 2:
    try:
      # (iterated programs: the depths the n-D entry offers; lower() clips
      # them to `iterate`, the library mixes them per stream length)
      program_spec('dense', run, lower.LowerOptions(fuse=lower.DEFAULT_FUSE),
                   tuple(st.tile_size[:-1]) + (1 << 20,))
    except util.SodaError:
      pass
  chunks = [lower.runtime_text()]
  copies = []
  for s_ in st.input_stmts:
    nb, ct = banks[s_.name], table[s_.name].c_type
    shift = shifts[s_.name]
    if nb == 1 and shift == 0:
      continue
    name = 'soda_unwire_%s' % s_.name
    body = ['  %s* __restrict__ dense = (%s*)a.buf[%d];' % (ct, ct, nb),
            '  const int64_t src = k + %d;   // the host delayed this tensor' %
            shift,
            '  const %s* bank = (const %s*)a.buf[src %% %d];' % (ct, ct, nb),
            '  dense[k] = src < a.extent[0] ? bank[src / %d] : (%s)0;' %
            (nb, ct)]
    elem = table[s_.name].size_in_bytes
    vec = _vec_copy_source(name, ct, elem, nb, shift // nb, True) \
        if shift % nb == 0 else None
    chunks.append(vec or _WIRE_SRC % dict(name=name, body='\n'.join(body)))
    copies.append(('unwire_%s' % s_.name, name, nb, 1, elem,
                   256 * (16 // elem) * nb if vec else 256))
  for s_ in st.output_stmts:
    nb, ct = banks[s_.name], table[s_.name].c_type
    off = 0 if late else offsets[s_.name]
    if nb == 1 and off == 0:
      continue                 # the program writes the bank itself
    name = 'soda_wire_%s' % s_.name
    body = ['  const %s* __restrict__ dense = (const %s*)a.buf[0];' % (ct, ct),
            '  %s* bank = (%s*)a.buf[1 + k %% %d];' % (ct, ct, nb),
            '  bank[k / %d] = k >= %d ? dense[k - %d] : (%s)0;' %
            (nb, off, off, ct)]
    elem = table[s_.name].size_in_bytes
    vec = _vec_copy_source(name, ct, elem, nb, 0, False) if off == 0 else None
    chunks.append(vec or _WIRE_SRC % dict(name=name, body='\n'.join(body)))
    copies.append(('wire_%s' % s_.name, name, 1, nb, elem,
                   256 * (16 // elem) * nb if vec else 256))
  source = '\n'.join(chunks)
  for tag, name, n_in, n_out, elem, per_block in copies:
    specs[tag] = ProgramSpec(tag, source,
                             _copy_plan(name, n_in, n_out, elem, per_block),
                             [name])
  return desc, specs


class StreamProgram:
  """`<app>_kernel` for one program: banked wire streams in, banked out.  A
  thin Python handle on the library's stream object (soda_hip_stream_*): the
  per-call sequence -- un-interleave, program, shift + re-interleave -- runs
  behind the C ABI."""

  # narrower tiles leave most of a 64-lane x V-wide marching strip idle and the
  # linear form (`direct` kernels) wins (heat3d 32x32 tiles: 0.50 vs 0.79 ms)
  DENSE_MIN_TILE0 = 256

  def __init__(self, stencil: core.Stencil, device: int = 0,
               dense: Optional[bool] = None, direct: bool = True):
    """`dense`: None = use the n-D marching kernels when the stream allows it
    and -- for device-resident banks -- the tile is wide enough to fill them
    (host banks: whenever the stream allows it), True = whenever the stream
    allows it, False = always the linear form.  `direct`: see stream_specs."""
    self.stencil = stencil
    self.device = device
    self.desc, self.specs = stream_specs(stencil, dense, direct)
    self.banks = {s.name: len(s.dram)
                  for s in stencil.input_stmts + stencil.output_stmts}
    self.stencil_offset = stencil_offsets(stencil)   # a program constant
    self._lib = lib = runtime.library()
    self._programs: Dict[str, ctypes.c_void_p] = {}
    codes = {}
    for tag, spec in self.specs.items():
      if spec.source not in codes:
        codes[spec.source] = runtime.compile_source(
            spec.source, '%s_%s.hip' % (stencil.app_name, tag))
      code = codes[spec.source]
      h = ctypes.c_void_p()
      runtime.check(
          lib.soda_hip_program_create(code, len(code), ctypes.byref(spec.plan),
                                      device, ctypes.byref(h)),
          'loading %s of `%s`' % (tag, stencil.app_name))
      self._programs[tag] = h
    d = self.desc
    lin = (ctypes.c_void_p * d.num_linear)(*[
        self._programs['linear%d' % d.linear_vec[k]]
        for k in range(d.num_linear)])
    unw = (ctypes.c_void_p * d.num_inputs)(*[
        self._programs.get('unwire_%s' % n) for n in stencil.input_names])
    wir = (ctypes.c_void_p * d.num_outputs)(*[
        self._programs.get('wire_%s' % n) for n in stencil.output_names])
    self._handle = ctypes.c_void_p()
    runtime.check(
        lib.soda_hip_stream_create(ctypes.byref(d), self._programs.get('dense'),
                                   lin, unw, wir, ctypes.byref(self._handle)),
        'stream object of `%s`' % stencil.app_name)
    runtime.check(
        lib.soda_hip_stream_set_device_dense_min_tile(
            self._handle, 0 if dense else self.DENSE_MIN_TILE0),
        'dense policy of `%s`' % stencil.app_name)

  @property
  def last_mode(self) -> Optional[str]:
    return {1: 'dense', 2: 'linear'}.get(
        self._lib.soda_hip_stream_last_mode(self._handle))

  def _flat(self, banks_by_name, names) -> 'ctypes.Array':
    ptrs = []
    for n in names:
      if len(banks_by_name[n]) != self.banks[n]:
        raise util.InputError('%s has %d banks' % (n, self.banks[n]))
      ptrs.extend(banks_by_name[n])
    return (ctypes.c_void_p * len(ptrs))(*ptrs)

  # -- <app>_kernel on device-resident banks -------------------------------
  def run_banked_device(self, out_banks: Dict[str, List[int]],
                        in_banks: Dict[str, List[int]],
                        coalesced_data_num: int, stream: int = 0) -> None:
    st = self.stencil
    runtime.check(
        self._lib.soda_hip_stream_run_device(
            self._handle, self._flat(out_banks, st.output_names),
            self._flat(in_banks, st.input_names), coalesced_data_num,
            ctypes.c_void_p(stream)), '%s_kernel' % st.app_name)

  # -- <app>_kernel on host banks (what SODA_CPP_BINDING links against) ------
  def run_banked_host(self, out_banks: Dict[str, list], in_banks: Dict[str, list],
                      coalesced_data_num: int) -> None:
    """numpy arrays per bank, sized as the reference host allocates them."""
    st = self.stencil
    outs = {n: [a.ctypes.data for a in arrs] for n, arrs in out_banks.items()}
    ins = {n: [a.ctypes.data for a in arrs] for n, arrs in in_banks.items()}
    runtime.check(
        self._lib.soda_hip_stream_run_host(
            self._handle, self._flat(outs, st.output_names),
            self._flat(ins, st.input_names), coalesced_data_num),
        '%s_kernel' % st.app_name)

  def close(self) -> None:
    if getattr(self, '_handle', None):
      self._lib.soda_hip_stream_destroy(self._handle)
      self._handle = None
    for h in getattr(self, '_programs', {}).values():
      self._lib.soda_hip_program_destroy(h)
    self._programs = {}

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

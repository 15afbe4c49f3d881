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
    out[s.name] = core.get_stencil_distance(pts, tile) - util.serialize(
        core.get_stencil_window_offset(pts), tile)
  return out


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
    self.stencil_dim = core.get_stencil_dim(window)
    self.stencil_distance = st.stencil_distance
    tile = st.tile_size
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


_WIRE_SRC = '''
// bank k %% NB, index k / NB  <->  stream position k   (reference
// docs/data-layout.md "Multi-Bank"; frt/host.py:241-246, 422-424)
extern "C" __global__ void __launch_bounds__(256) %(name)s(soda_hip_kargs_t a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= a.extent[0]) return;
%(body)s
}
'''


class StreamProgram:
  """`<app>_kernel` for one program: banked wire streams in, banked out."""

  # narrower tiles leave most of a 64-lane x V-wide marching strip idle and the
  # linear form (`direct` kernels) wins (heat3d 32x32 tiles: 0.50 vs 0.79 ms)
  DENSE_MIN_TILE0 = 256

  def __init__(self, stencil: core.Stencil, device: int = 0,
               dense: Optional[bool] = None):
    """`dense`: None = use the n-D marching kernels when the stream allows it
    and the tile is wide enough to fill them, True = whenever the stream
    allows it, False = always the linear form."""
    if stencil.param_stmts:
      raise util.SemanticError('stream mode does not support param tensors')
    if stencil.preserve_border:
      raise util.SemanticError(
          'stream mode does not support border: preserve (tile edges are not '
          'grid borders)')
    self.stencil = stencil
    self.device = device
    self.flat = linearize(stencil)
    table = stencil.symbol_table
    self.banks = {s.name: len(s.dram)
                  for s in stencil.input_stmts + stencil.output_stmts}
    self.stencil_offset = stencil_offsets(stencil)   # a program constant
    self._lib = runtime.library()
    # two ways of running the program on the de-interleaved stream:
    #  * as the ORIGINAL n-D program on the stream viewed as a dense array of
    #    extent (tile_size..., rows) -- tiles are whole rows laid end to end --
    #    with the fast marching kernels (built on first use);
    #  * as the linearised 1-D program (always valid, `direct` kernels).
    self._linear = {}      # cells per thread -> Program of the 1-D form
    self._linear_program(1)
    self._dense = None
    if dense is None:
      dense = stencil.dim >= 2 and stencil.tile_size[0] >= self.DENSE_MIN_TILE0
    self._dense_failed = not dense
    self.last_mode = None
    # wire <-> dense copy kernels, one per tensor
    chunks = [lower.runtime_text()]
    self._copy = {}
    for s in stencil.input_stmts:
      nb, ct = self.banks[s.name], table[s.name].c_type
      name = 'soda_unwire_%s' % s.name
      body = ['  %s* __restrict__ dense = (%s*)a.buf[%d];' % (ct, ct, nb),
              '  const %s* bank = (const %s*)a.buf[k %% %d];' % (ct, ct, nb),
              '  dense[k] = bank[k / %d];' % nb]
      chunks.append(_WIRE_SRC % dict(name=name, body='\n'.join(body)))
      self._copy[s.name] = (name, nb, 1, table[s.name].size_in_bytes)
    for s in stencil.output_stmts:
      nb, ct = self.banks[s.name], table[s.name].c_type
      off = self.stencil_offset[s.name]
      name = 'soda_wire_%s' % s.name
      body = ['  const %s* __restrict__ dense = (const %s*)a.buf[0];' % (ct, ct),
              '  %s* bank = (%s*)a.buf[1 + k %% %d];' % (ct, ct, nb),
              '  bank[k / %d] = k >= %d ? dense[k - %d] : (%s)0;' %
              (nb, off, off, ct)]
      chunks.append(_WIRE_SRC % dict(name=name, body='\n'.join(body)))
      self._copy[s.name] = (name, 1, nb, table[s.name].size_in_bytes)
    code = runtime.compile_source('\n'.join(chunks),
                                  '%s_wire.hip' % stencil.app_name)
    self._handles = {}
    for tensor, (name, n_in, n_out, elem) in self._copy.items():
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
      plan.kernels[0].tile[0] = 256
      for d in range(1, runtime.MAX_DIM):
        plan.kernels[0].tile[d] = 1
      plan.num_passes = 1
      plan.passes[0].fused_iters = 1
      plan.passes[0].num_kernels = 1
      h = ctypes.c_void_p()
      runtime.check(
          self._lib.soda_hip_program_create(code, len(code), ctypes.byref(plan),
                                            device, ctypes.byref(h)),
          'loading %s' % name)
      self._handles[tensor] = (h, plan)
    self._scratch = {}

  def _linear_program(self, vec: int) -> 'runtime.Program':
    if vec not in self._linear:
      self._linear[vec] = runtime.Program(
          self.flat, lower.LowerOptions(strategy='direct', vec=vec),
          device=self.device)
    return self._linear[vec]

  # -- device memory helpers -----------------------------------------------
  def _dev(self, key, nbytes):
    cur = self._scratch.get(key)
    if cur and cur[1] >= nbytes:
      return cur[0]
    if cur:
      self._lib.soda_hip_free(self.device, ctypes.c_void_p(cur[0]))
    p = ctypes.c_void_p()
    runtime.check(self._lib.soda_hip_malloc(self.device, nbytes,
                                            ctypes.byref(p)), 'malloc')
    self._scratch[key] = (p.value, nbytes)
    return p.value

  def _launch(self, tensor, outs, ins, n, stream):
    h, _ = self._handles[tensor]
    o = (ctypes.c_void_p * len(outs))(*outs)
    i = (ctypes.c_void_p * len(ins))(*ins)
    ext = (ctypes.c_int32 * 1)(n)
    runtime.check(self._lib.soda_hip_run_device(h, o, i, ext, 1,
                                                ctypes.c_void_p(stream)),
                  'wire copy of %s' % tensor)

  # -- <app>_kernel on device-resident banks -------------------------------
  def run_banked_device(self, out_banks: Dict[str, List[int]],
                        in_banks: Dict[str, List[int]],
                        coalesced_data_num: int, stream: int = 0) -> None:
    st = self.stencil
    table = st.symbol_table
    epc = {n: st.burst_width // table[n].width_in_bits * self.banks[n]
           for n in self.banks}
    n_elems = {n: coalesced_data_num * epc[n] for n in self.banks}
    n = n_elems[st.input_names[0]]
    if any(v != n for v in n_elems.values()):
      raise util.InputError(
          'stream mode needs every tensor to move the same number of elements '
          'per cycle (burst width / element width x banks)')
    if n >= 2**31:
      raise util.InputError('stream longer than 2^31 elements')
    dense_in, dense_out = [], []
    for name in st.input_names:
      if len(in_banks[name]) != self.banks[name]:
        raise util.InputError('%s has %d banks' % (name, self.banks[name]))
      if self.banks[name] == 1:
        # one bank: the bank IS the dense stream, and inputs are never written
        dense_in.append(in_banks[name][0])
        continue
      d = self._dev(('in', name), n * table[name].size_in_bytes)
      self._launch(name, [d], in_banks[name], n, stream)
      dense_in.append(d)
    for name in st.output_names:
      dense_out.append(self._dev(('out', name), n * table[name].size_in_bytes))
    if self._run_dense(dense_out, dense_in, n, epc[st.input_names[0]], stream):
      self.last_mode = 'dense'
    else:
      vec = lower.default_vec(self.flat)
      while vec > 1 and n % vec:
        vec //= 2
      self._linear_program(vec).run_device(dense_out, dense_in, (n,),
                                           st.iterate, stream)
      self.last_mode = 'linear'
    for name, d in zip(st.output_names, dense_out):
      if len(out_banks[name]) != self.banks[name]:
        raise util.InputError('%s has %d banks' % (name, self.banks[name]))
      self._launch(name, out_banks[name], [d], n, stream)

  def _run_dense(self, dense_out, dense_in, n: int, epc: int,
                 stream: int) -> bool:
    """Runs the original program on the stream seen as (tile..., rows).  Valid
    when the stream really is such an array:
      * every tile starts on a row-block boundary.  A tile occupies
        round_up(block * extent_last, epc) elements (frt/host.py:137-142), so
        this holds for any extent iff block % epc == 0 (the kernel is not told
        the extent, only the cycle count);
      * the void tail the host appends (kStencilDistance elements,
        frt/host.py:151-162) is at least one row block, so that the partial
        last row the view drops holds no cell of any tile.
    Cells whose taps cross a tile edge read zeros here and wrapped neighbours
    in the linear form: both are outside the valid region."""
    st = self.stencil
    if st.dim < 2 or self._dense_failed:
      return False
    block = 1
    for t in st.tile_size[:-1]:
      block *= t
    rows = n // block
    if rows < 1 or st.stencil_distance < block or block % epc:
      return False
    extent = tuple(st.tile_size[:-1]) + (rows,)
    if self._dense is None:
      try:
        self._dense = runtime.Program(st, lower.LowerOptions(),
                                      device=self.device, extent=extent)
      except util.SodaError:
        self._dense_failed = True
        return False
    self._dense.run_device(dense_out, dense_in, extent, st.iterate, stream)
    return True

  # -- <app>_kernel on host banks (what SODA_CPP_BINDING links against) ------
  def run_banked_host(self, out_banks: Dict[str, list], in_banks: Dict[str, list],
                      coalesced_data_num: int) -> None:
    """numpy arrays per bank, sized as the reference host allocates them."""
    lib = self._lib
    dev_in, dev_out = {}, {}
    for name, arrs in in_banks.items():
      dev_in[name] = []
      for b, a in enumerate(arrs):
        p = self._dev(('hin', name, b), a.nbytes)
        runtime.check(lib.soda_hip_memcpy_h2d(ctypes.c_void_p(p),
                                              ctypes.c_void_p(a.ctypes.data),
                                              a.nbytes, None), 'h2d')
        dev_in[name].append(p)
    for name, arrs in out_banks.items():
      dev_out[name] = [self._dev(('hout', name, b), a.nbytes)
                       for b, a in enumerate(arrs)]
    self.run_banked_device(dev_out, dev_in, coalesced_data_num)
    runtime.synchronize()
    for name, arrs in out_banks.items():
      for p, a in zip(dev_out[name], arrs):
        runtime.check(lib.soda_hip_memcpy_d2h(ctypes.c_void_p(a.ctypes.data),
                                              ctypes.c_void_p(p), a.nbytes,
                                              None), 'd2h')

  def close(self) -> None:
    for h, _ in getattr(self, '_handles', {}).values():
      self._lib.soda_hip_program_destroy(h)
    self._handles = {}
    for p, _ in getattr(self, '_scratch', {}).values():
      self._lib.soda_hip_free(self.device, ctypes.c_void_p(p))
    self._scratch = {}
    for prog in getattr(self, '_linear', {}).values():
      prog.close()
    self._linear = {}
    if getattr(self, '_dense', None):
      self._dense.close()
      self._dense = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

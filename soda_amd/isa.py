"""What the COMPILED kernels are: per-kernel identity and instruction counts
read from the gfx950 code object, no GPU needed.

Two uses:

* `isa_key` -- a content key per KERNEL, over its own machine code and kernel
  descriptor.  profiles/traffic.json ties every counter measurement to the key
  of the kernel it was taken on; until round 4 that key covered the whole
  module SOURCE, so an edit to a helper no benched kernel even calls orphaned
  every number of the round.
* `static_profile` / `march_valu_per_launch` -- vector-ALU wave-instructions a
  launch of a marching kernel issues, counted from the disassembly (straight-
  line prologue + loop body x trips per wave, over the live waves of the
  launch geometry): the VALU-issue roof of a temporally blocked stencil in the
  bench line even when no PMC pass was run, and a cross-check of SQ_INSTS_VALU
  when one was.

The reference has no counterpart (its kernel is an HLS dataflow design whose
resource report comes from Vivado: reference src/soda/model/xilinx.py).
"""
import hashlib
import os
import re
import struct
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

OBJDUMP_CANDIDATES = ('/opt/rocm/lib/llvm/bin/llvm-objdump',
                      '/opt/rocm/llvm/bin/llvm-objdump')

# a wave64 VALU instruction holds a SIMD for 2 cycles at best (16 lanes x 2
# passes x dual issue; tools/valubench.py measured 2.0-2.15 with >= 2 waves
# resident), 1024 SIMDs, 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
SIMDS = 1024
PEAK_CLOCK_HZ = 2.4e9
CYCLES_PER_VALU = 2.0
VALU_PEAK_PER_S = SIMDS * PEAK_CLOCK_HZ / CYCLES_PER_VALU


def _sections(code: bytes):
  shoff, = struct.unpack_from('<Q', code, 0x28)
  shentsize, shnum = struct.unpack_from('<HH', code, 0x3A)
  return [struct.unpack_from('<IIQQQQIIQQ', code, shoff + i * shentsize)
          for i in range(shnum)]


def _symbols(code: bytes) -> Dict[str, Tuple[int, int, int, int]]:
  """{symbol: (type, section index, value, size)} of the ELF symbol table."""
  out = {}
  heads = _sections(code)
  for h in heads:
    if h[1] != 2:                     # SHT_SYMTAB
      continue
    str_off = heads[h[6]][4]
    for pos in range(h[4], h[4] + h[5], 24):
      st_name, st_info, _, shndx, value, size = struct.unpack_from(
          '<IBBHQQ', code, pos)
      end = code.index(b'\0', str_off + st_name)
      name = code[str_off + st_name:end].decode()
      if name:
        out[name] = (st_info & 0xF, shndx, value, size)
  return out


def _symbol_bytes(code: bytes, sym: Tuple[int, int, int, int]) -> bytes:
  _, shndx, value, size = sym
  heads = _sections(code)
  if not 0 < shndx < len(heads):
    return b''
  h = heads[shndx]
  off = h[4] + (value - h[3])
  return code[off:off + size]


def isa_key(code: bytes, kernel: str) -> Optional[str]:
  """Content key of ONE kernel of a code object: sha256 over its machine code
  and its kernel descriptor (`<kernel>.kd`: register counts, LDS, flags).
  None if the symbols cannot be read."""
  try:
    if code[:4] != b'\x7fELF':
      return None
    syms = _symbols(code)
    if kernel not in syms:
      return None
    h = hashlib.sha256()
    h.update(_symbol_bytes(code, syms[kernel]))
    kd = syms.get(kernel + '.kd')
    if kd:
      h.update(_symbol_bytes(code, kd))
    return h.hexdigest()[:24]
  except Exception:      # noqa: BLE001 -- a malformed object has no key
    return None


def isa_keys(code: bytes, kernels: Sequence[str]) -> Dict[str, Optional[str]]:
  return {k: isa_key(code, k) for k in kernels}


def objdump() -> Optional[str]:
  for p in OBJDUMP_CANDIDATES:
    if os.path.exists(p):
      return p
  return None


_LINE = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):')
_TARGET = re.compile(r'<([^>+]+)\+0x([0-9a-fA-F]+)>\s*$')
_FUNC = re.compile(r'^([0-9a-f]+) <([^>]+)>:')


def disassemble(code: bytes) -> Dict[str, List[Tuple[int, str, str, Optional[int]]]]:
  """{kernel: [(address, mnemonic, operands, branch target address or None)]}
  through llvm-objdump (part of the ROCm image, here and on the GPU box).
  Raises OSError if it is not installed."""
  tool = objdump()
  if tool is None:
    raise OSError('llvm-objdump not found under /opt/rocm')
  import tempfile
  with tempfile.NamedTemporaryFile(suffix='.hsaco') as f:
    f.write(code)
    f.flush()
    text = subprocess.run([tool, '-d', '--mcpu=gfx950', f.name],
                          capture_output=True, text=True, check=True).stdout
  out: Dict[str, List[Tuple[int, str, str, Optional[int]]]] = {}
  cur = None
  base = 0
  for line in text.splitlines():
    m = _FUNC.match(line)
    if m:
      base = int(m.group(1), 16)
      cur = out.setdefault(m.group(2), [])
      continue
    if cur is None:
      continue
    m = _LINE.match(line)
    if not m:
      continue
    mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
    target = None
    if mnem.startswith('s_cbranch') or mnem == 's_branch':
      t = _TARGET.search(line)
      if t:
        target = base + int(t.group(2), 16)
    cur.append((addr, mnem, ops, target))
  return out


def _classify(mnem: str, ops: str) -> List[str]:
  kinds = []
  if mnem.startswith('v_'):
    kinds.append('valu')
    if '_dpp' in mnem or 'row_sh' in ops or 'wave_sh' in ops or \
        'quad_perm' in ops or 'row_bcast' in ops or 'row_mirror' in ops:
      kinds.append('dpp')
  elif mnem.startswith('ds_'):
    kinds.append('lds')
    if mnem.startswith('ds_swizzle') or mnem.startswith('ds_bpermute') or \
        mnem.startswith('ds_permute'):
      kinds.append('lds_crossbar')
  elif mnem.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
    kinds.append('vmem')
    kinds.append('vmem_store' if '_store' in mnem else 'vmem_load')
  elif mnem.startswith('s_waitcnt'):
    kinds.append('waitcnt')
  elif mnem.startswith('s_'):
    kinds.append('salu')
  return kinds


def _count(instrs) -> Dict[str, int]:
  out: Dict[str, int] = {'total': 0}
  for _, mnem, ops, _ in instrs:
    out['total'] += 1
    for k in _classify(mnem, ops):
      out[k] = out.get(k, 0) + 1
  return out


def static_profile(instrs) -> Dict[str, Dict[str, int]]:
  """Instruction counts of one kernel by region: `pre` (entry to the head of
  its innermost-outermost loop, executed once by a live wave), `loop` (one
  trip of the body), `post`.  The marching kernels have exactly one loop (the
  unrolled row-step loop); a kernel without a backward branch is all `pre`."""
  loops = [(t, a) for a, m, _, t in instrs if t is not None and t <= a]
  if not loops:
    return {'pre': _count(instrs), 'loop': {'total': 0}, 'post': {'total': 0},
            'loops': 0}
  head = min(t for t, _ in loops)
  tail = max(a for _, a in loops)
  pre = [i for i in instrs if i[0] < head]
  body = [i for i in instrs if head <= i[0] <= tail]
  post = [i for i in instrs if i[0] > tail]
  return {'pre': _count(pre), 'loop': _count(body), 'post': _count(post),
          'loops': len(loops)}


def march_waves(extent: Sequence[int], tile: Sequence[int], dim: int,
                tile_rows: int = 1) -> List[Tuple[int, int]]:
  """[(waves, rows each marches)] of a one-wave-per-block marching launch on
  `extent` with block tile `tile` (the library's geometry): strips along
  dimension 0 (x row tiles in 3-D) x chunks along the last dimension, the last
  chunk shorter."""
  ax = dim - 1
  others = -(-extent[0] // tile[0])
  if dim == 3:
    others *= -(-extent[1] // max(1, tile[1]))
  n, c = extent[ax], tile[ax]
  full, rest = divmod(n, c)
  out = [(others * full, c)] if full else []
  if rest:
    out.append((others, rest))
  return out


def march_valu_per_launch(profile: Dict[str, Dict[str, int]],
                          extent: Sequence[int], tile: Sequence[int], dim: int,
                          warm: int, peeled_steps: int, unroll: int,
                          kind: str = 'valu') -> float:
  """Wave-instructions of class `kind` one launch issues: every live wave runs
  the prologue (peeled warm-up included) once and ceil((chunk + warm - peeled)
  / unroll) trips of the loop (march.py `_emit_wave`: tau from m_begin + m_lo +
  lead + peeled to m_end + max_delay, `warm` = max_delay - m_lo - lead)."""
  pre = profile['pre'].get(kind, 0) + profile['post'].get(kind, 0)
  loop = profile['loop'].get(kind, 0)
  total = 0.0
  for waves, rows in march_waves(extent, tile, dim):
    trips = max(0, -(-(rows + warm - peeled_steps) // unroll))
    total += waves * (pre + trips * loop)
  return total


def module_static(module, code: bytes, tiles: Dict[str, Sequence[int]],
                  extent: Sequence[int]) -> Dict[str, dict]:
  """{kernel name: static counts} for the marching kernels of a lowered module
  (`module`: codegen.hip.module.Module, `code`: its code object, `tiles`:
  {kernel: block tile} the library uses on `extent` -- Program.geometry):
  wave-instructions per launch by class, per row step of the loop, the key of
  the kernel's machine code, and the VALU-issue floor of a launch."""
  dis = disassemble(code)
  dim = module.stencil.dim
  out = {}
  for k in module.kernels:
    tune = k.tune or {}
    if 'axis' not in tune or k.name not in dis or k.name not in tiles:
      continue
    if tune.get('waves_per_block', 1) != 1 or tune.get('pipe', 1) != 1:
      continue          # (blocks of several waves: not modelled here)
    prof = static_profile(dis[k.name])
    unroll = int(tune.get('unroll') or 1)
    peeled = int(tune.get('peel_trips') or 0) * unroll
    warm = int(tune.get('warm') or 0)
    entry = {'isa_key': isa_key(code, k.name), 'loop_row_steps': unroll,
             'loops': prof['loops']}
    for kind in ('valu', 'dpp', 'lds_crossbar', 'vmem_load', 'vmem_store',
                 'waitcnt', 'salu'):
      entry['%s_per_launch' % kind] = march_valu_per_launch(
          prof, extent, tiles[k.name], dim, warm, peeled, unroll, kind)
      entry['%s_per_row_step' % kind] = prof['loop'].get(kind, 0) / float(unroll)
    entry['min_issue_ms'] = entry['valu_per_launch'] / VALU_PEAK_PER_S * 1e3
    out[k.name] = entry
  return out

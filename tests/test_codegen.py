"""CPU-side checks of the HIP backend: the C-ABI library loads and exports
every symbol include/soda_hip.h declares, struct mirrors match, every corpus
program lowers and JIT-compiles for gfx950 (hiprtc needs no GPU), the CLI
behaves like the reference driver, and the product path fails loudly -- never
falls back -- when it cannot run on a GPU."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT, SODA_DIR, soda_path
from soda_amd import core, util


def _declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'soda_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(soda_hip_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(built):
  from soda_amd import runtime
  lib = ctypes.CDLL(runtime.LIB_PATH)
  declared = _declared_symbols()
  assert len(declared) >= 20
  for name in declared:
    assert hasattr(lib, name), 'libsoda_hip.so lacks %s' % name
  assert sorted(runtime.API) == declared, 'runtime.API and the header differ'


def test_struct_mirrors_match(built):
  from soda_amd import runtime
  lib = runtime.library()   # also runs the built-in layout check
  assert lib.soda_hip_abi_version() == runtime.ABI_VERSION
  # buf, stride, extent, ntile, tile, origin, gextent; skip_from, skip_count,
  # reserved[2]
  assert lib.soda_hip_sizeof(0) == 16 * 8 + 4 * 8 + 5 * 4 * 4 + 4 * 4
  assert lib.soda_hip_sizeof(3) == ctypes.sizeof(runtime.Plan)
  assert lib.soda_hip_sizeof(99) == 0
  assert lib.soda_hip_status_string(5) == b'no usable GPU'


def test_kargs_struct_in_device_runtime_matches_header():
  from soda_amd.codegen.hip import lower
  rt = lower.runtime_text()
  assert 'void* buf[16];' in rt and 'int64_t stride[4];' in rt
  assert 'int32_t extent[4];' in rt and 'int32_t ntile[4];' in rt
  assert 'int32_t tile[4];' in rt
  assert rt.index('gextent[4]') < rt.index('skip_from;') < rt.index(
      'skip_count;') < rt.index('reserved[2];')


ALL = sorted(f for f in os.listdir(SODA_DIR) if f.endswith('.soda')) + [
    'skew2d.soda']


@pytest.mark.parametrize('name', ALL)
def test_corpus_lowers_and_jit_compiles(built, name, tmp_path):
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path(name))
  fuse = (3,) if (len(stencil.input_names) == len(stencil.output_names) and
                  stencil.input_types == stencil.output_types) else ()
  if fuse:    # fused kernels are only built up to the program's iterate
    stencil = core.from_file(soda_path(name), iterate=3)
  opts = lower.LowerOptions(fuse=fuse)
  opts.vec = runtime.pick_vec(stencil, None)
  mod = lower.lower(stencil, opts)
  plan = runtime.make_plan(mod)
  assert plan.passes[plan.num_passes - 1].fused_iters == 1
  code = runtime.compile_source(mod.source, name, cache_dir=str(tmp_path))
  assert code[:4] == b'\x7fELF'
  for k in mod.kernels:
    assert k.name.encode() in code
  # (contrast: 17 x 17 taps; as the reference really evaluates it -- seven
  # stages of <= 32 taps after `inline.rebalance` -- it fits register windows
  # at one cell per lane)
  if stencil.dim == 2:
    assert all(p.kind == 'march2d' for p in mod.passes)
    # (three fused iterations of contrast do not fit; nor of erosion and xcorr with the
    # partial-window tensors optimization/windows.py gives them -- the library
    # then runs such a program one iteration per launch)
    if name not in ('contrast.soda', 'erosion.soda', 'xcorr.soda'):
      assert sorted(p.fused_iters for p in mod.passes) == sorted({1} | set(fuse))
  else:
    assert all(p.kind == 'march3d' for p in mod.passes)
    assert max(p.fused_iters for p in mod.passes) <= lower.MAX_FUSE_3D


def test_march2d_geometry_for_jacobi2d():
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=4)
  opts = lower.LowerOptions(fuse=(4,), vec=4, chunk_rows=64)
  mod = lower.lower(stencil, opts)
  by_t = {p.fused_iters: p for p in mod.passes}
  k4 = mod.kernels[by_t[4].kernels[0]]
  k1 = mod.kernels[by_t[1].kernels[0]]
  # 4 fused iterations: 4 halo cells per side = 1 lane -> 62 lanes x 4 cells
  assert k4.tile[:2] == (248, 64) and k4.block == (64, 1, 1)
  # 1 iteration: the halo cell per side comes from the edge lanes' own loads,
  # all 64 lanes are valid and rows start on 1 KiB boundaries
  assert k1.tile[:2] == (256, 64)
  assert by_t[1].traffic_model['edge'] == (1, 1)
  assert by_t[4].traffic_model['edge'] == (0, 0)
  # T rows of reach below + T above; the prefetch-fill ticks are load-only
  assert by_t[4].traffic_model["warm_rows"] == 4 + 4
  assert by_t[4].traffic_model['bytes_per_cell_min'] == 8
  src = mod.source
  assert 'soda_lane_dn' in src and 'soda_lane_up' in src
  assert '__shared__' not in src            # registers + DPP only


def test_march3d_geometry_for_heat3d():
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('heat3d.soda'))
  mod = lower.lower(stencil, lower.LowerOptions(vec=4))
  by_t = {q.fused_iters: q for q in mod.passes}
  # heat3d ships with iterate 2: the default fusion depth is capped at 2 in 3-D
  assert sorted(by_t) == [1, lower.MAX_FUSE_3D]
  p2 = by_t[2]
  assert p2.kind == 'march3d' and p2.traffic_model['rows_in'] == 4 + 2 * 2
  p = by_t[1]
  k = mod.kernels[p.kernels[0]]
  # 512-wide grids: two aligned 256-cell strips; 4 output rows + 2 halo rows
  # in registers; 64 planes marched per wave
  assert p.kind == 'march3d' and k.tile[:3] == (256, 4, 64)
  assert p.traffic_model['rows_in'] == 6 and p.traffic_model['edge'] == (1, 1)


def test_chunk_is_sized_from_the_code_objects_registers(built, tmp_path):
  """Launch geometry is the LIBRARY's (soda_hip_plan_geometry, no GPU needed):
  chunk lengths follow the extent and the compiled kernels' register counts so
  that the grid is whole rounds of waves on the 1024 SIMDs."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'))
  opts = lower.LowerOptions(fuse=(12,), vec=4, peel=0)
  mod = lower.lower(stencil, opts)
  code = runtime.compile_source(mod.source, 'j.hip', cache_dir=str(tmp_path))
  res = runtime.kernel_resources(code)
  assert set(res) == {k.name for k in mod.kernels}
  plan = runtime.make_plan(mod, res)
  tiles, ns = runtime.plan_geometry(plan, (8192, 8192))
  for k, tile in zip(mod.kernels, tiles):
    assert 16 <= res[k.name]['vgpr'] <= 256 and res[k.name]['scratch'] == 0
    waves = -(-8192 // tile[0]) * -(-8192 // tile[1])
    slots = 1024 * runtime.waves_per_simd(res[k.name]['vgpr'])
    assert tile[0] == k.tile[0] and tile[1] >= 64
    assert waves <= max(slots, 8192 // 64 * 36)
  assert all(v > 0 for v in ns)
  # a 1224-row slab of the same grid (one GPU's share of an 8-GPU run): far
  # shorter chunks, the same strips
  slab_tiles, slab_ns = runtime.plan_geometry(plan, (8192, 1224))
  assert slab_tiles[0][0] == tiles[0][0] and slab_tiles[0][1] < 40
  assert slab_ns[0] < ns[0] / 3
  # without register counts the chunk stays as generated
  bare = runtime.make_plan(mod)
  assert runtime.plan_geometry(bare, (8192, 8192))[0][0][:2] == mod.kernels[
      0].tile[:2]
  assert runtime.waves_per_simd(64) == 8 and runtime.waves_per_simd(65) == 7
  assert runtime.waves_per_simd(128) == 4 and runtime.waves_per_simd(167) == 3
  assert runtime.waves_per_simd(300) == 1


def test_schedule_follows_the_extent(built, tmp_path):
  """The multiset of passes that advances `iterate` iterations is chosen per
  extent from the library's time model (soda_hip_plan_schedule): always adds
  up, deepest fusion on the full grid, and never the one-iteration pass where
  a fused one can do the work."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  opts = runtime.resolve_options(stencil, lower.LowerOptions(fuse=(12, 8, 4)),
                                 (8192, 8192))
  mod = lower.lower(stencil, opts)
  code = runtime.compile_source(mod.source, 'j.hip', cache_dir=str(tmp_path))
  plan = runtime.make_plan(mod, runtime.kernel_resources(code))
  depth = [p.fused_iters for p in mod.sorted_passes()]
  assert depth == [12, 8, 4, 1]
  for extent in ((8192, 8192), (8192, 1224), (8192, 200), (640, 480)):
    for iterate in (1, 7, 100, 1000):
      count = runtime.plan_schedule(plan, extent, iterate)
      assert sum(c * t for c, t in zip(count, depth)) == iterate
      if iterate >= 4:
        assert count[-1] <= 3
  full = runtime.plan_schedule(plan, (8192, 8192), 100)
  assert full[0] >= 6          # mostly the 12-deep pass on the full grid


def test_library_refuses_extents_a_kernel_cannot_run(built, tmp_path):
  """The constraints the kernels were generated under are checked behind the
  C ABI, not only in the Python wrapper: rows that are not a multiple of the
  vector width, planes too large for the 1 GiB buffer window."""
  from soda_amd import runtime, util
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'))
  mod = lower.lower(stencil, lower.LowerOptions(fuse=(), vec=4))
  code = runtime.compile_source(mod.source, 'j.hip', cache_dir=str(tmp_path))
  plan = runtime.make_plan(mod, runtime.kernel_resources(code))
  runtime.plan_geometry(plan, (8192, 64))
  with pytest.raises(util.BackendError, match='multiple of 4 cells'):
    runtime.plan_geometry(plan, (8190, 64))
  direct = lower.lower(stencil, lower.LowerOptions(strategy='direct', vec=4))
  dplan = runtime.make_plan(direct)
  with pytest.raises(util.BackendError, match='multiple of 4 cells'):
    runtime.plan_geometry(dplan, (8190, 64))
  heat = core.from_file(soda_path('heat3d.soda'))
  hmod = lower.lower(heat, lower.LowerOptions(fuse=(), vec=4))
  hcode = runtime.compile_source(hmod.source, 'h.hip', cache_dir=str(tmp_path))
  hplan = runtime.make_plan(hmod, runtime.kernel_resources(hcode))
  # 4096 x 4096 floats = 64 MiB per plane -> 16 planes per GiB, 2 of them halo
  tiles, _ = runtime.plan_geometry(hplan, (4096, 4096, 100))
  assert tiles[0][2] <= 14
  with pytest.raises(util.BackendError, match='1 GiB buffer window'):
    runtime.plan_geometry(hplan, (32768, 16384, 9))


def test_vector_width_follows_row_length():
  from soda_amd import runtime
  j = core.from_file(soda_path('jacobi2d.soda'))
  b = core.from_file(soda_path('blur.soda'))
  assert runtime.pick_vec(j, (8192, 8192)) == 4
  assert runtime.pick_vec(j, (1002, 64)) == 2
  assert runtime.pick_vec(j, (1001, 64)) == 1
  assert runtime.pick_vec(b, (16384, 16384)) == 8
  assert runtime.pick_vec(b, (2000, 1024)) == 8
  assert runtime.pick_vec(b, (2004, 10)) == 4


def test_unsupported_programs_are_rejected():
  from soda_amd.codegen.hip import lower
  with pytest.raises(util.SemanticError, match='8/16/32/64-bit'):
    lower.lower(core.from_text(
        'kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
        'input uint6: a(8, *)\noutput uint6: o(0, 0) = a(0, 0)'))
  with pytest.raises(util.SemanticError, match='param'):
    lower.lower(core.from_text(
        'kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
        'input float: a(8, *)\nparam float: p[4]\n'
        'output float: o(0, 0) = a(0, 0) * p[1]'))
  with pytest.raises(util.SemanticError, match='march'):
    lower.lower(core.from_text(
        'kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
        'input float: a\noutput float: o(0) = a(0) + a(1)'),
                lower.LowerOptions(strategy='march'))


def test_no_cpu_fallback_without_gpu(built):
  """On a box without a GPU the product path must raise, not compute."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  if runtime.device_count() > 0:
    pytest.skip('a GPU is present')
  stencil = core.from_file(soda_path('jacobi2d.soda'))
  with pytest.raises(util.BackendError, match='GPU|device'):
    runtime.Program(stencil, lower.LowerOptions(), extent=(64, 64))


def test_product_code_never_imports_the_oracle():
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'soda_amd')):
    for f in files:
      if f.endswith(('.py', '.cpp', '.h')):
        text = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f
        assert 'oracle/' not in text.replace('CPU oracle', ''), f


def _sodac(*argv):
  return subprocess.run([sys.executable, '-m', 'soda_amd.sodac'] + list(argv),
                        cwd=ROOT, capture_output=True, text=True)


def test_sodac_prints_hip_kernel(built, tmp_path):
  out = tmp_path / 'k.hip'
  r = _sodac(soda_path('blur.soda'), '--hip-kernel', str(out))
  assert r.returncode == 0, r.stderr
  text = out.read_text()
  assert 'extern "C" __global__' in text and 'blur_march2d' in text
  r = _sodac(soda_path('jacobi2d.soda'), '--hip-kernel', '-', '--iterate', '8',
             '--hip-fuse', '8')
  assert r.returncode == 0 and 'jacobi2d_march2d_T8' in r.stdout
  r = _sodac('-', '--hip-kernel', '-')
  assert r.returncode != 0   # empty stdin is a syntax error -> exit 1


def test_sodac_error_exit_codes(tmp_path):
  bad = tmp_path / 'bad.soda'
  bad.write_text('kernel: k\nburst width: 64\n')
  r = _sodac(str(bad), '--hip-kernel', '-')
  assert r.returncode == 1 and 'expected' in r.stderr
  r = _sodac(soda_path('blur.soda'), '--iterate', '0', '--hip-kernel', '-')
  assert r.returncode == 1 and 'cannot iterate 0 times' in r.stderr
  r = _sodac(soda_path('blur.soda'), '--iterate', '2', '--hip-kernel', '-')
  assert r.returncode == 0   # blur: 1 input, 1 output, same type -> iterable


def test_backend_plugin_surface():
  """add_arguments / print_code, the reference's backend API
  (sodac.py:99-102,198-200)."""
  import argparse
  from soda_amd.codegen.hip import core as hip
  parser = argparse.ArgumentParser()
  hip.add_arguments(parser)
  args = parser.parse_args(['--hip-kernel', '-', '--hip-fuse', '2', '6'])
  assert args.hip_kernel == '-' and args.hip_fuse == [2, 6]
  assert not args.hip_backend
  opts = hip.options_from_args(args)
  assert opts.fuse == (2, 6) and opts.nt_load is None and opts.nt_store is None
  assert opts.resolved(2).nt_load and not opts.resolved(3).nt_load
  one_shot = opts.resolved(2, iterated=False)
  assert one_shot.nt_store and not one_shot.nt_load
  stencil = core.from_file(soda_path('jacobi2d.soda'))
  assert hip.default_extent(stencil) == [32, 6]   # frt/host.py:454-461


def test_march_loop_body_is_branch_free():
  """The point of buffer addressing: no `if` around any load or store of the
  marching loop, and a load-only prologue as deep as the prefetch."""
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('jacobi2d.soda'))
  mod = lower.lower(stencil, lower.LowerOptions(fuse=()))
  src = mod.source
  loop = src[src.index('for (; tau < tau_end'):]
  assert 'soda_buf_load_frag' in loop and 'soda_buf_store_frag' in loop
  assert 'if (' not in loop
  assert src.count('// prologue: loads of plane') == lower.default_prefetch(1)
  # a stage without inputs forbids the peeling (its planes start at tick 0)
  text = open(soda_path('jacobi2d.soda')).read()
  const = core.from_text(
      'kernel: c\nburst width: 64\nunroll factor: 2\niterate: 1\n'
      'input float: a(32, *)\noutput float: b(0, 0) = a(0, 1) + a(0, -1)\n'
      'output float: c(0, 0) = 38\n')
  assert '// prologue' not in lower.lower(const, lower.LowerOptions(fuse=())).source


def test_print_code_accepts_a_foreign_stencil_object(tmp_path):
  """The plug-in called from the reference's driver gets the reference's
  Stencil (a different class, haoda expression tree): anything that prints the
  DSL normal form works."""
  import argparse
  from soda_amd.codegen.hip import core as hip

  class Foreign:                      # stands for reference core.Stencil
    def __init__(self, text):
      self._text = text

    def __str__(self):
      return self._text

  ours = core.from_file(soda_path('jacobi2d.soda'), iterate=4)
  parser = argparse.ArgumentParser()
  hip.add_arguments(parser)
  out = str(tmp_path / 'k.hip')
  hip.print_code(Foreign(str(ours)), parser.parse_args(['--hip-kernel', out]))
  text = open(out).read()
  assert 'jacobi2d_march2d_T4' in text and 'extern "C" __global__' in text
  ref = str(tmp_path / 'r.hip')
  hip.print_code(ours, parser.parse_args(['--hip-kernel', ref]))
  assert open(ref).read() == text


def test_row_covering_blocks_in_the_plan(built):
  """3-D fused kernels built for a known row length: the block's waves cover
  the row (x-halos through LDS), the descriptor says so, and the library --
  without a GPU -- shapes launches for rows that fit and refuses longer ones."""
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  stencil = core.from_file(soda_path('heat3d.soda'), iterate=6)
  opts = lower.LowerOptions(fuse=(2,), vec=4, peel=0, row_cells=512)
  mod = lower.lower(stencil, opts)
  fused = [k for k in mod.kernels if k.tune.get('fused') == 2][0]
  assert fused.name.endswith('_xs2') and fused.block == (128, 1, 1)
  assert fused.tile[0] == 512 and fused.tune['max_extent0'] == 512
  assert fused.tune['lane_redundancy'] == 1.0       # all 64 lanes store
  text = mod.source
  assert '__shared__ float soda_xs_g_in[' in text      # the inputs' end cells
  assert '__shared__ float soda_xs_t0_out[' in text    # and iteration 0's
  assert text.count('soda_bcast(hv_') > 0
  # no narrow edge loads in the fused kernel any more
  body = text[text.index('void __launch_bounds__(128) ' + fused.name):]
  body = body[:body.index('\nextern "C"')] if '\nextern "C"' in body else body
  assert 'exb0_' not in body
  plan = runtime.make_plan(mod, {k.name: dict(vgpr=245) for k in mod.kernels})
  d = [plan.kernels[i] for i in range(plan.num_kernels)
       if plan.kernels[i].name.decode() == fused.name][0]
  assert d.max_extent0 == 512
  tiles, _ = runtime.plan_geometry(plan, (512, 512, 512))
  assert tiles[0][0] == 512
  runtime.plan_geometry(plan, (300, 64, 64))
  with pytest.raises(util.BackendError, match='at most 512 cells'):
    runtime.plan_geometry(plan, (768, 64, 64))
  # without the row length nothing is assumed
  plain = lower.lower(stencil, lower.LowerOptions(fuse=(2,), vec=4, peel=0))
  assert not any('_xs' in k.name for k in plain.kernels)


@pytest.mark.parametrize('shift', ['bperm', 'swzh', 'mixh'])
def test_sliding_sum_setup_with_lds_pipe_shifts_compiles(built, shift, tmp_path):
  """A sliding sum whose taps sit a lane away (store index off-centre in x):
  the accumulator's set-up in front of the loop needs the lane-shifted copies
  too -- with shifts on the LDS pipe they are collected apart ("early") and
  the set-up once forgot to emit them (found by tools/fuzz_scan.py options:
  a loud compile error, never a wrong result)."""
  import sys
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import fuzz
  from soda_amd import runtime
  from soda_amd.codegen.hip import lower
  text, dim, _ = fuzz.window_program(712)
  assert 'w0(1, -1) = in0(0, -9) + ' in text
  stencil = core.from_text(text)
  opts = runtime.resolve_options(
      stencil, lower.LowerOptions(fuse=(3,), lane_shift=shift), (1100, 220),
      probe=False)
  mod = lower.lower(stencil, opts)
  assert 'xa_t0_w0_r0' in mod.source
  assert runtime.compile_source(mod.source, 'w.hip', cache_dir=str(tmp_path))


def test_counter_evidence_is_for_the_kernels_head_builds(built):
  """profiles/traffic.json (PMC bytes and VALU counts bench.py reports as
  `roofline.traffic` / checks `roofline.valu` against) names, per kernel, the
  hash of the machine code it was measured on (soda_amd/isa.py isa_key).  The
  default bench module lowered and compiled HERE must carry the same keys --
  else the driver's bench line would drop the evidence as it did in round 4
  (`traffic: null` after a late soda_rt.h edit).  A red test means: run
  tools/profile_round.sh on the GPU box again and commit profiles/traffic.json
  as the round's last act."""
  import json
  from soda_amd import core, isa, runtime
  from soda_amd.codegen.hip import lower
  path = os.path.join(ROOT, 'profiles', 'traffic.json')
  with open(path) as f:
    table = json.load(f)
  st = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  extent = (8192, 8192)
  opts = runtime.resolve_options(st, lower.LowerOptions(fuse=lower.DEFAULT_FUSE),
                                 extent)
  mod = lower.lower(st, opts)
  code = runtime.compile_source(mod.source, '%s.hip' % st.app_name)
  plan = runtime.make_plan(mod, runtime.kernel_resources(code))
  keyed = {n: e for n, e in table.items() if e.get('isa_key')}
  if not keyed:
    pytest.skip('profiles/traffic.json predates per-kernel keys (rounds 1-4)')
  if runtime.compiler_version() not in {e.get('compiler') for e in
                                        keyed.values()}:
    pytest.skip('another compiler than the evidence was taken with')
  # (which passes a step runs is the CLOCK's choice on the GPU box -- 4 x T13 +
  # 4 x T12 on 8192^2 -- so: every kernel of the module that the table holds
  # must match, and the deepest pass, which every schedule uses, must be there)
  names = [k.name for k in mod.kernels]
  deepest = mod.kernels[mod.sorted_passes()[0].kernels[0]].name
  assert deepest in keyed, (
      '%s is what HEAD builds for the deepest pass but profiles/traffic.json '
      'has no counters for it: re-run tools/profile_round.sh' % deepest)
  # (the one-iteration kernel's name is shared with the module of bench.py's
  # single_iter leg, other code under the same name: the fused kernels only)
  fused = {k.name for k in mod.kernels if (k.tune or {}).get('fused', 1) > 1}
  scheduled = [n for n in names if n in keyed and n in fused]
  for name in scheduled:
    assert keyed[name]['isa_key'] == isa.isa_key(code, name), (
        'profiles/traffic.json was measured on other machine code of %s: '
        're-run tools/profile_round.sh as the last GPU act' % name)
    assert keyed[name].get('hbm_bytes_per_launch', 0) > 0
  # and the static count agrees with the counter where both exist
  tiles, _ = runtime.plan_geometry(plan, extent)
  static = isa.module_static(mod, code, {k.name: t for k, t in
                                         zip(mod.kernels, tiles)}, extent)
  for name in scheduled:
    pmc = keyed[name].get('valu_wave_instructions_per_launch')
    if pmc and name in static:
      assert abs(static[name]['valu_per_launch'] / pmc - 1) < 0.02, (
          name, static[name]['valu_per_launch'], pmc)


def test_static_valu_count_of_the_benched_kernels(built):
  """soda_amd/isa.py on the default bench module, no GPU: every fused kernel
  has exactly one loop, issues ~20 VALU instructions per 4 cells per fused
  iteration in it (5 per cell: four adds and a multiply -- the arithmetic
  floor; the lane shifts ride on the adds as DPP modifiers or run on the LDS
  pipe), and the per-launch count lands where round 4's SQ_INSTS_VALU
  counters did (T13: 9.97e7, T12: 8.74e7)."""
  from soda_amd import core, isa, runtime
  from soda_amd.codegen.hip import lower
  if isa.objdump() is None:
    pytest.skip('llvm-objdump not installed')
  st = core.from_file(soda_path('jacobi2d.soda'), iterate=100)
  extent = (8192, 8192)
  opts = runtime.resolve_options(st, lower.LowerOptions(fuse=lower.DEFAULT_FUSE),
                                 extent)
  mod = lower.lower(st, opts)
  code = runtime.compile_source(mod.source, '%s.hip' % st.app_name)
  plan = runtime.make_plan(mod, runtime.kernel_resources(code))
  tiles, _ = runtime.plan_geometry(plan, extent)
  static = isa.module_static(mod, code, {k.name: t for k, t in
                                         zip(mod.kernels, tiles)}, extent)
  assert len(static) == len(mod.kernels)
  for k in mod.kernels:
    s = static[k.name]
    depth = k.tune['fused']
    assert s['loops'] == 1 and s['isa_key']
    per_level = s['valu_per_row_step'] / depth
    assert 20.0 <= per_level <= (27.0 if depth == 1 else 21.5), (k.name,
                                                                  per_level)
    if depth >= 8:      # mixh: one DPP and one swizzle per level and row step
      assert s['dpp_per_row_step'] == depth
      assert s['lds_crossbar_per_row_step'] == depth
  t13 = [s for n, s in static.items() if '_T13_' in n][0]
  t12 = [s for n, s in static.items() if '_T12_' in n][0]
  assert abs(t13['valu_per_launch'] / 9.97e7 - 1) < 0.03
  assert abs(t12['valu_per_launch'] / 8.74e7 - 1) < 0.03
  # two kernels of one module have different keys; the key is stable
  assert t13['isa_key'] != t12['isa_key']
  assert t13['isa_key'] == isa.isa_key(code, [n for n in static
                                              if '_T13_' in n][0])

"""The HIP backend's plug-in surface: `add_arguments` + `print_code`.

Every backend of the reference is a module with exactly these two functions,
registered and invoked by the driver (reference src/soda/sodac.py:99-102,
198-200; src/soda/codegen/frt/core.py:10-27; src/soda/codegen/xilinx/
opencl.py:55-142).  This module slots into the same place:

  --hip-kernel FILE   print the generated HIP source ('-' = stdout), the
                      counterpart of --xocl-kernel;
  --hip-host FILE     print the C++ host: soda::app::<app>() with the reference's
                      --frt-host signature, on libsoda_hip.so (host.py);
  --hip-backend       JIT-build the kernels for gfx950 and RUN the program on
                      the GPU with the reference harness's inputs, checking
                      nothing by itself (the oracle lives in tests/); prints
                      one JSON line with timing.
"""
import argparse
import json
import shutil
import sys
import tempfile
import time

from soda_amd import core, util
from soda_amd.codegen.hip import lower


def add_arguments(parser) -> None:
  parser.add_argument('--hip-kernel', type=str, dest='hip_kernel',
                      metavar='file', help='HIP kernel code for gfx950')
  parser.add_argument('--hip-host', type=str, dest='hip_host', metavar='file',
                      help='C++ host: soda::app::<app>() with the signature '
                      'of --frt-host, on libsoda_hip.so')
  parser.add_argument('--hip-wire-kernel', type=str, dest='hip_wire_kernel',
                      metavar='file', help='C++ definition of extern "C" '
                      '<app>_kernel(banks..., coalesced_data_num) -- the '
                      "reference kernel's ABI on its tiled, banked streams -- "
                      'on libsoda_hip.so: links under the unmodified '
                      '--frt-host output built with -DSODA_CPP_BINDING')
  parser.add_argument('--hip-backend', action='store_true', dest='hip_backend',
                      help='JIT-build the HIP kernels and run them on the GPU')
  parser.add_argument('--hip-strategy', type=str, dest='hip_strategy',
                      choices=('auto', 'direct', 'march', 'lds', 'ldswin'), default='auto',
                      help='kernel family: register-marching wavefront strips '
                      '(2-D / 3-D programs) or the direct kernels')
  parser.add_argument('--hip-fuse', type=int, nargs='*', dest='hip_fuse',
                      metavar='T', default=list(lower.DEFAULT_FUSE),
                      help='temporal blocking: the numbers of iterations a '
                      'launch may fuse; the library mixes them per extent '
                      '(default: %s; depths that do not fit the registers ' %
                      ' '.join(map(str, lower.DEFAULT_FUSE)) +
                      'are dropped, 3-D programs fuse at most 2)')
  parser.add_argument('--hip-vec', type=int, dest='hip_vec', metavar='V',
                      help='cells per lane per row (default: 16 bytes worth)')
  parser.add_argument('--hip-chunk-rows', type=int, dest='hip_chunk_rows',
                      default=None, help='rows one wavefront marches over '
                      '(default: sized at load time so the grid fills the GPU '
                      'in one round of waves)')
  parser.add_argument('--hip-prefetch', type=int, dest='hip_prefetch',
                      default=None, help='input rows loaded ahead of use')
  parser.add_argument('--hip-tile-rows', type=int, dest='hip_tile_rows',
                      default=None, help='3-D: output rows held per wavefront')
  parser.add_argument('--hip-extent', type=int, nargs='+', dest='hip_extent',
                      metavar='N', help='grid size for --hip-backend')
  parser.add_argument('--hip-waves', type=str, dest='hip_waves', default='1x1',
                      metavar='XxY', help='wavefronts per block along '
                      'dimension 0 and 1')
  parser.add_argument('--hip-pipe', type=int, dest='hip_pipe', default=None,
                      metavar='W', help='split the fused iterations of a 2-D '
                      'kernel over W wavefronts of a block (rows handed on '
                      'through LDS); the fusion depth must be a multiple '
                      '(default 1)')
  parser.add_argument('--hip-nt-store', action='store_true',
                      dest='hip_nt_store',
                      help='non-temporal instead of plain output stores')
  parser.add_argument('--hip-no-nt-load', action='store_true',
                      dest='hip_no_nt_load',
                      help='plain instead of non-temporal input loads')
  parser.add_argument('--hip-no-xcd-swizzle', action='store_true',
                      dest='hip_no_xcd_swizzle',
                      help='do not remap blocks so neighbours share an XCD')
  parser.add_argument('--hip-device', type=int, dest='hip_device', default=0)
  parser.add_argument('--hip-no-probe', action='store_true',
                      dest='hip_no_probe',
                      help='--hip-kernel / --hip-host: do not size the peeled '
                      'warm-up by trial compilations (needs libsoda_hip.so + '
                      'hiprtc); the emitted text is then the same on every '
                      'box and toolchain')
  parser.add_argument('--hip-gpus', type=int, dest='hip_gpus', default=1,
                      metavar='N', help='--hip-backend / --hip-host: cut the '
                      'grid into N slabs along the streamed dimension, one '
                      'per GPU, halo exchange by peer copies '
                      '(soda_hip_group_*)')
  parser.add_argument('--hip-exchange-every', type=int, default=0,
                      dest='hip_exchange_every', metavar='K',
                      help='with --hip-gpus N: iterations between halo '
                      'exchanges (default 0: the library picks)')
  parser.add_argument('--hip-virtual', action='store_true', dest='hip_virtual',
                      help='with --hip-gpus N: all N slabs on --hip-device '
                      '(the whole schedule on one GPU)')


def options_from_args(args: argparse.Namespace) -> lower.LowerOptions:
  wx, wy = (int(v) for v in args.hip_waves.lower().split('x'))
  return lower.LowerOptions(strategy=args.hip_strategy,
                            fuse=tuple(args.hip_fuse or ()),
                            vec=args.hip_vec,
                            chunk_rows=args.hip_chunk_rows,
                            prefetch=args.hip_prefetch, waves_x=wx, waves_y=wy,
                            nt_store=True if args.hip_nt_store else None,
                            nt_load=False if args.hip_no_nt_load else None,
                            tile_rows=args.hip_tile_rows,
                            xcd_swizzle=not args.hip_no_xcd_swizzle,
                            pipe=args.hip_pipe)


def print_code(stencil: core.Stencil, args: argparse.Namespace) -> None:
  if not isinstance(stencil, core.Stencil):
    # called from the reference's own driver with ITS Stencil (expression tree
    # in the un-vendored haoda): both print the same DSL normal form
    # (reference src/soda/core.py:157-166, src/tests/test_grammar.py:24-61), so
    # re-read it
    stencil = core.from_text(str(stencil))
  if args.hip_kernel is not None:
    from soda_amd import runtime
    opts = runtime.resolve_options(stencil, options_from_args(args), None,
                                   probe=not getattr(args, 'hip_no_probe',
                                                     False))
    with tempfile.TemporaryFile(mode='w+') as tmp:
      tmp.write(lower.lower(stencil, opts).source)
      tmp.seek(0)
      if args.hip_kernel == '-':
        shutil.copyfileobj(tmp, sys.stdout)
      else:
        with open(args.hip_kernel, 'w') as f:
          shutil.copyfileobj(tmp, f)
  if getattr(args, 'hip_host', None) is not None:
    from soda_amd.codegen.hip import host
    text = host.print_host(stencil, options_from_args(args), args.hip_extent,
                           gpus=max(1, int(getattr(args, 'hip_gpus', 1) or 1)),
                           probe=not getattr(args, 'hip_no_probe', False))
    if args.hip_host == '-':
      sys.stdout.write(text)
    else:
      with open(args.hip_host, 'w') as f:
        f.write(text)
  if getattr(args, 'hip_wire_kernel', None) is not None:
    from soda_amd.codegen.hip import wire
    text = wire.print_wire_kernel(stencil)
    if args.hip_wire_kernel == '-':
      sys.stdout.write(text)
    else:
      with open(args.hip_wire_kernel, 'w') as f:
        f.write(text)
  if args.hip_backend:
    run(stencil, args)


def default_extent(stencil: core.Stencil):
  """The reference harness's default problem size: tile sizes, and one more
  than the stencil's height in the streamed dimension (frt/host.py:454-461)."""
  dims = list(stencil.tile_size[:-1])
  dims.append(stencil.stencil_dim[-1] + 1)
  return dims


def run(stencil: core.Stencil, args: argparse.Namespace) -> None:
  import numpy as np
  from soda_amd import runtime
  extent = list(args.hip_extent or default_extent(stencil))
  if len(extent) != stencil.dim:
    raise util.InputError('--hip-extent needs %d values' % stencil.dim)
  shape = tuple(extent[::-1])
  rng = np.random.default_rng(0)
  inputs = {}
  for name, t in zip(stencil.input_names, stencil.input_types):
    if t.is_float:
      inputs[name] = rng.random(shape, dtype=np.float64).astype(t.np_name)
    else:  # p + q (+ r): the reference harness's integer init
      grids = np.indices(shape).sum(axis=0)
      inputs[name] = grids.astype(t.np_name)
  for pstmt in stencil.param_stmts:
    # the reference harness's param init: the sum of the indices
    # (frt/host.py:530-541)
    size = pstmt.size or (1,)
    inputs[pstmt.name] = np.indices(size).sum(axis=0).astype(
        pstmt.haoda_type.np_name)
  gpus = max(1, int(getattr(args, 'hip_gpus', 1) or 1))
  extra = {}
  if gpus > 1:
    # one host thread, N GPUs: slabs along the streamed dimension
    devices = ([args.hip_device] * gpus if getattr(args, 'hip_virtual', False)
               else list(range(gpus)))
    prog = runtime.Group(stencil, extent, devices, options_from_args(args),
                         exchange_every=int(getattr(args, 'hip_exchange_every',
                                                    0) or 0))
    t0 = time.time()
    outputs = prog.run_host(inputs)
    seconds = time.time() - t0
    st = prog.stats()
    extra = {'gpus': gpus, 'devices': devices,
             'exchange_every': st['exchange_every'],
             'exchanges': st['exchanges'], 'split_passes': st['split_passes']}
  else:
    prog = runtime.Program(stencil, options_from_args(args),
                           device=args.hip_device, extent=extent)
    t0 = time.time()
    outputs = prog.run(inputs)
    seconds = time.time() - t0
  cells = float(np.prod(extent)) * stencil.iterate
  print(json.dumps({
      'kernel': stencil.app_name, 'extent': extent,
      'iterate': stencil.iterate, 'seconds_incl_copies': seconds,
      'cells_iters_per_s_incl_copies': cells / seconds, **extra,
      'kernels': [k.name for k in prog.module.kernels],
      'checksum': {n: float(np.asarray(v, dtype=np.float64).sum())
                   for n, v in outputs.items()},
  }))

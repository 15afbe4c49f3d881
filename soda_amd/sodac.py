"""`sodac`-compatible driver with the HIP backend plugged in.

Same positional argument and the same override flags as the reference driver
(reference src/soda/sodac.py:26-116: --burst-width, --unroll-factor,
--replication-factor, --tile-size, --dram-in, --dram-out, --iterate, --border,
--cluster, -v/-q), the same error-to-exit-code mapping (:231-238), the same
backend plug-in call sequence (:198-200).  The FPGA backends (--xocl-*,
--iocl-*, --frt-host) are out of scope here; the HIP backend takes their place.
"""
import argparse
import logging
import sys
from typing import List, Optional

from soda_amd import core, grammar, util
from soda_amd.codegen.hip import core as hip

logger = logging.getLogger('sodac')


def build_parser() -> argparse.ArgumentParser:
  parser = argparse.ArgumentParser(
      prog='sodac',
      description='Stencil with Optimized Dataflow Architecture (SODA) '
      'compiler, MI355X HIP backend')
  parser.add_argument('--verbose', '-v', action='count', dest='verbose',
                      help='increase verbosity')
  parser.add_argument('--quiet', '-q', action='count', dest='quiet',
                      help='decrease verbosity')
  parser.add_argument('--burst-width', type=int, dest='burst_width',
                      help='override burst width')
  parser.add_argument('--unroll-factor', type=int, metavar='UNROLL_FACTOR',
                      dest='unroll_factor', help='override unroll factor')
  parser.add_argument('--replication-factor', type=int,
                      metavar='REPLICATION_FACTOR', dest='replication_factor',
                      help='override replication factor')
  parser.add_argument('--tile-size', type=int, nargs='+', metavar='TILE_SIZE',
                      dest='tile_size',
                      help='override tile size; 0 means no overriding on that '
                      'dimension')
  parser.add_argument('--dram-in', type=str, dest='dram_in',
                      help='override DRAM configuration for input')
  parser.add_argument('--dram-out', type=str, dest='dram_out',
                      help='override DRAM configuration for output')
  parser.add_argument('--iterate', type=int, metavar='#ITERATION',
                      dest='iterate',
                      help='override iterate directive; repeat execution '
                      'multiple times iteratively')
  parser.add_argument('--border', type=str, metavar='(ignore|preserve)',
                      dest='border', help='override border handling strategy')
  parser.add_argument('--cluster', type=str,
                      metavar='(none|fine|coarse|full)', dest='cluster',
                      help='module clustering level (FPGA only; accepted, '
                      'no effect on results)')
  parser.add_argument(type=str, dest='soda_src', metavar='file',
                      help='soda source code')
  hip.add_arguments(parser.add_argument_group('HIP (MI355X) backend'))
  return parser


def main(argv: Optional[List[str]] = None) -> None:
  parser = build_parser()
  args = parser.parse_args(sys.argv[1:] if argv is None else argv)
  level = ((args.quiet or 0) - (args.verbose or 0)) * 10 + logging.WARNING
  logging.basicConfig(
      level=min(max(level, logging.DEBUG), logging.CRITICAL),
      format='%(levelname)s:%(name)s:%(lineno)d: %(message)s', force=True)
  try:
    if args.soda_src == '-':
      program = grammar.parse(sys.stdin.read())
    else:
      program = grammar.parse_file(args.soda_src)
    logger.debug('soda program parsed:\n  %s',
                 str(program).replace('\n', '\n  '))
    stencil = core.from_program(
        program, burst_width=args.burst_width, border=args.border,
        iterate=args.iterate, cluster=args.cluster, dram_in=args.dram_in,
        dram_out=args.dram_out, tile_size=args.tile_size,
        unroll_factor=args.unroll_factor,
        replication_factor=args.replication_factor)
    logger.debug('stencil obtained: %s', stencil)
    hip.print_code(stencil, args)
  except util.SodaSyntaxError as e:
    logger.error(e)
    sys.exit(1)
  except (util.SemanticError, util.InputError, util.BackendError) as e:
    logger.error(e)
    sys.exit(1)
  except util.SemanticWarn as w:
    logger.warning(w)


if __name__ == '__main__':
  main()

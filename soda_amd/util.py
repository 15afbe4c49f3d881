"""Small shared helpers: error types and the dim-0-fastest linearisation.

Reference behaviour restated here (not copied):
  * error classes mirror the ones `sodac` maps to exit codes
    (reference src/soda/sodac.py:231-238; haoda.util.SemanticError & co).
  * serialize/deserialize follow reference src/soda/util.py:9-24 -- dimension 0
    is the fastest-varying one, the last dimension is unbounded (tile size 0).
"""
from typing import Iterable, Sequence, Tuple

COORDS_TILED = 'xyzw'
COORDS_IN_TILE = 'ijkl'
COORDS_IN_ORIG = 'pqrs'
MAX_DIM = 4


class SodaError(Exception):
  """Base class of every error this package raises on purpose."""


class SodaSyntaxError(SodaError):
  """The .soda text does not match the grammar (textX syntax error analogue)."""

  def __init__(self, message: str, line: int = 0, col: int = 0):
    super().__init__(message)
    self.message = message
    self.line = line
    self.col = col

  def __str__(self) -> str:
    if self.line:
      return '%d:%d: %s' % (self.line, self.col, self.message)
    return self.message


class SemanticError(SodaError):
  """The program parses but means nothing executable."""


class SemanticWarn(SodaError):
  """Suspicious but executable."""


class InternalError(SodaError):
  """A bug in this package."""


class InputError(SodaError):
  """Bad user input other than the DSL text itself (names, shapes, flags)."""


class BackendError(SodaError):
  """The HIP backend failed (JIT, module load, launch); never swallowed."""


def serialize(vec: Sequence[int], tile_size: Sequence[int]) -> int:
  """Linear offset of `vec` inside a tile; dim 0 fastest (ref util.py:9-12)."""
  offset = 0
  pitch = 1
  for d, v in enumerate(vec):
    offset += v * pitch
    if d + 1 < len(vec):
      pitch *= tile_size[d]
  return offset


def serialize_iter(vecs: Iterable[Sequence[int]],
                   tile_size: Sequence[int]) -> list:
  return [serialize(v, tile_size) for v in vecs]


def deserialize(offset: int, tile_size: Sequence[int]) -> Tuple[int, ...]:
  """Inverse of serialize for in-tile coordinates (ref util.py:17-24)."""
  out = []
  for size in tile_size[:-1]:
    out.append(offset % size)
    offset //= size
  out.append(offset)
  return tuple(out)


def lst2str(items: Iterable) -> str:
  return '[%s]' % ', '.join(map(str, items))


def idx2str(idx: Iterable) -> str:
  return '(%s)' % ', '.join(map(str, idx))

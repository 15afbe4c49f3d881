import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

SODA_DIR = os.path.join(ROOT, 'tests', 'golden', 'soda')
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_addoption(parser):
  parser.addoption(
      '--fuzz-budget', default=None,
      help='scale the random-program GPU seed sets (tests/fuzz.py '
      'budget_seeds): 1 = sized for the ~10-minute driver run (default), 2 = '
      'the full sets of earlier rounds, 0.5 = half')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by gpurun)')
  # (test modules read it at import, i.e. during collection, after this hook)
  if config.getoption('--fuzz-budget') is not None:
    os.environ['SODA_FUZZ_BUDGET'] = str(config.getoption('--fuzz-budget'))


# kernel names a GPU test of this session has compared with the oracle
VERIFIED_KERNELS = set()

# Order of the GPU tests under `-x`: what the bench line stands on first, the
# timing-sensitive multi-rank rehearsals last, so that one failure there cannot
# hide the hot-path parity tests (round 3's driver run: 587 of 599 unreached).
# (file, prefix of the test name) -> rank; first match wins, unlisted GPU tests
# of a file take the file's default.
_GPU_ORDER = [
    # 0: closed forms, committed golden vectors, lane-shift directions
    ('test_hip_parity.py', 'test_blur_reference_init_closed_form', 0),
    ('test_hip_parity.py', 'test_heat3d_ramp_is_fixed_point', 0),
    ('test_hip_parity.py', 'test_committed_golden_vectors_on_gpu', 0),
    ('test_hip_parity.py', 'test_dpp_wave_shift_direction', 0),
    ('test_hip_parity.py', 'test_swizzle_lane_shift_direction', 0),
    # 1: BASELINE C2 / C3 at full size and every kernel of the benched schedule
    ('test_hip_parity.py', 'test_full_size_properties', 1),
    ('test_hip_parity.py', 'test_every_kernel_of_the_benched_schedule', 1),
    # 2: the corpus
    ('test_hip_parity.py', 'test_corpus_', 2),
    # 3: BASELINE C4 / C5 as written, one GPU
    ('test_baseline_configs.py', 'test_c5_', 3),
    ('test_baseline_configs.py', 'test_c4_', 3),
    # 4: the rest of the single-GPU parity file, then independent nests + fuzz
    ('test_hip_parity.py', '', 4),
    ('test_fuzz_nest.py', '', 5),
    ('test_fuzz.py', '', 5),
    ('test_optimization.py', '', 6),
    ('test_stream.py', '', 6),
    ('test_host.py', '', 6),
    ('test_codegen.py', '', 6),
    # 7: slab groups (one process, events between streams)
    ('test_group.py', '', 7),
    # 8: ranks as threads / processes on the one GPU -- last
    ('test_baseline_configs.py', '', 8),
    ('test_dist.py', '', 9),
]


def pytest_collection_modifyitems(config, items):
  def rank(item):
    if item.get_closest_marker('gpu') is None:
      return 6
    fname = os.path.basename(str(item.fspath))
    for f, prefix, r in _GPU_ORDER:
      if f == fname and item.name.startswith(prefix):
        return r
    return 6
  order = {id(it): i for i, it in enumerate(items)}
  items.sort(key=lambda it: (rank(it), order[id(it)]))


@pytest.fixture(scope='session')
def built():
  """Native pieces built once per session (hipcc / gcc; no GPU needed)."""
  import __graft_entry__ as entry
  entry.build_library()
  import subprocess
  subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True,
                 capture_output=True)
  return True


def soda_path(name):
  path = os.path.join(SODA_DIR, name)
  if os.path.exists(path):
    return path
  return os.path.join(GOLDEN_DIR, name)

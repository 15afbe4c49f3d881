import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

SODA_DIR = os.path.join(ROOT, 'tests', 'golden', 'soda')
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by gpurun)')


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

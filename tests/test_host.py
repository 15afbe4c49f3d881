"""`sodac --hip-host`: the generated C++ host (soda_amd/codegen/hip/host.py)
has the reference's operator signature (frt/host.py:62-88) and runs without
Python, on the C ABI alone."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, soda_path


def _generate(tmp_path, soda, *flags):
  out = os.path.join(str(tmp_path), 'host.cpp')
  subprocess.run([sys.executable, '-m', 'soda_amd.sodac', soda_path(soda),
                  '--hip-host', out, *flags], cwd=ROOT, check=True)
  return out


@pytest.mark.parametrize('soda,flags,needle', [
    ('blur.soda', (), 'int blur(const uint16_t* var_input_ptr,'),
    ('jacobi2d.soda', ('--iterate', '24', '--hip-fuse', '12', '4'),
     'soda_hip_run_host_box(program, inputs, outputs, 24,'),
    ('heat3d.soda', (), 'const int tile_size_1 = 32'),
    ('conv2d.soda', (), 'const float* var_w_ptr'),      # params after outputs
])
def test_generated_host_compiles(built, tmp_path, soda, flags, needle):
  src = _generate(tmp_path, soda, *flags)
  text = open(src).read()
  assert needle in text
  assert 'const char* bitstream' in text and 'namespace soda' in text
  subprocess.run(['g++', '-std=c++17', '-Wall', '-Werror', '-c', src,
                  '-I', os.path.join(ROOT, 'include'), '-o',
                  os.path.join(str(tmp_path), 'host.o')], check=True)


@pytest.mark.gpu
def test_generated_host_runs_without_python(built, tmp_path):
  src = _generate(tmp_path, 'blur.soda')
  exe = os.path.join(str(tmp_path), 'blur_host')
  libdir = os.path.join(ROOT, 'soda_amd')
  subprocess.run(['g++', '-std=c++17', '-O1', src,
                  os.path.join(ROOT, 'tests', 'host', 'blur_main.cpp'),
                  '-I', os.path.join(ROOT, 'include'), '-L', libdir,
                  '-lsoda_hip', '-Wl,-rpath,' + libdir, '-o', exe], check=True)
  env = dict(os.environ)
  env['LD_LIBRARY_PATH'] = '/opt/rocm/lib:' + env.get('LD_LIBRARY_PATH', '')
  run = subprocess.run([exe], capture_output=True, text=True, env=env,
                       timeout=300)
  assert run.returncode == 0 and run.stdout.startswith('OK'), (
      run.stdout + run.stderr)

import os, sys, runpy
node = int(sys.argv[1])
cpus = set(range(0, 64)) | set(range(128, 192)) if node == 0 else set(range(64, 128)) | set(range(192, 256))
if node >= 0:
  os.sched_setaffinity(0, cpus)
sys.argv = ['tools/hostpath.py', '--only', sys.argv[2]]
runpy.run_path('tools/hostpath.py', run_name='__main__')

"""Host-side (de)interleave of banked wire streams (soda_hip_host_weave_banks): GB/s of
the dense side, one thread and the pool, per element type and bank count."""
import sys, time, ctypes; sys.path.insert(0,'.')
import numpy as np
from soda_amd import runtime
lib = runtime.library()
for dt, nb in ((np.uint16,4),(np.float32,2),(np.uint16,2),(np.float32,4),(np.uint8,3)):
  groups = (256<<20)//(np.dtype(dt).itemsize*nb)
  banks=[np.ones(groups,dt) for _ in range(nb)]
  ptrs=(ctypes.c_void_p*nb)(*[b.ctypes.data for b in banks])
  dense=np.zeros(groups*nb,dt)
  for threads in (1,0):
    for to_dense in (1,0):
      ts=[]
      for _ in range(3):
        t0=time.perf_counter()
        lib.soda_hip_host_weave_banks(ptrs,nb,ctypes.c_void_p(dense.ctypes.data),0,groups*nb,dense.itemsize,to_dense,threads)
        ts.append(time.perf_counter()-t0)
      print(np.dtype(dt).name,nb,'threads',threads or 8,'to_dense',to_dense,'%.1f GB/s'%(dense.nbytes/min(ts)/1e9))

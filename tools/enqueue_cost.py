import sys, time; sys.path.insert(0,'/root/repo')
import torch
from soda_amd import core, runtime
from soda_amd.codegen.hip import lower
for ext in ((8192,8192),(8192,1224)):
    st = core.from_file('/root/repo/tests/golden/soda/jacobi2d.soda', iterate=100)
    a = torch.rand(ext[::-1], device='cuda'); b = torch.empty_like(a)
    prog = runtime.Program(st, lower.LowerOptions(fuse=(12,4)), extent=ext)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3): prog.run_device([b.data_ptr()],[a.data_ptr()], ext, stream=s)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(20): prog.run_device([b.data_ptr()],[a.data_ptr()], ext, stream=s)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(ext, 'host enqueue per step: %.1f us' % ((t1-t0)/20*1e6), 'GPU per step: %.1f us' % ((t2-t0)/20*1e6))

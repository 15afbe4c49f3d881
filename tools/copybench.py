#!/usr/bin/env python3
"""Measured streaming roofline: float4 copy kernels, two access regimes.
  fixed     src -> dst, same two buffers every launch
  pingpong  a -> b -> c -> a ... (what an iterated stencil does)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soda_amd import runtime, streamcopy

def main():
  n = 8192 * 8192 * (int(sys.argv[1]) if len(sys.argv) > 1 else 1)   # floats per array
  dev = torch.device('cuda', 0)
  bufs = [torch.rand(n, device=dev) for _ in range(3)]
  stream = torch.cuda.current_stream().cuda_stream
  out = []
  for nts in (0, 1):
    for ntl in (0, 1):
      for unroll in (1, 4):
        c = streamcopy.StreamCopy(unroll=unroll, nt_store=bool(nts), nt_load=bool(ntl))
        for mode in ('fixed', 'pingpong'):
          best = 1e9
          for r in range(3):
            a, b = runtime.Event(), runtime.Event()
            c.run(bufs[1].data_ptr(), bufs[0].data_ptr(), n, stream)
            a.record(stream)
            for i in range(30):
              if mode == 'fixed':
                c.run(bufs[1].data_ptr(), bufs[0].data_ptr(), n, stream)
              else:
                c.run(bufs[(i + 1) % 3].data_ptr(), bufs[i % 3].data_ptr(), n, stream)
            b.record(stream)
            best = min(best, a.elapsed_ms(b) / 30)
          out.append(dict(nt_store=nts, nt_load=ntl, unroll=unroll, mode=mode, us=best * 1e3, GBs=n * 8 / best / 1e6))
          print(json.dumps(out[-1]))

if __name__ == '__main__':
  main()

"""The rank-per-GPU overlapped exchange, rehearsed on ONE GPU (TEST
INFRASTRUCTURE, shared by tests/test_dist.py and tools/flake_loop.py).

Every rank is a thread with a compute stream and an exchange stream of its own
(dist.StreamOverlap), runs the REAL slab decomposition, exchange schedule and
kernels (Program.run_device(ghosts=, sends=, ghosts_ready=, sendable=) =
soda_hip_run_device_slab) and talks to its neighbours through tests/fabric.py.
`Case` builds the programs once and can be run many times; knobs skew the
ranks against each other so that an ordering hole has room to show:

  spin      {rank: GPU cycles}  a spinning kernel on that rank's compute
            stream in front of every interval (the rank falls behind ON THE
            GPU: its events fire late, its neighbours' receives queue up);
  sleep     {rank: seconds}     a host sleep in front of every interval (the
            rank ENQUEUES late: its neighbours block in the fabric);
  runs      chained dist.run calls per trial (every one after the first opens
            with an exchange of stale ghosts).
"""
import time

import numpy as np

import fabric


class Case:

  def __init__(self, stencil, extent, every, fuse, world, calibrate=False):
    from soda_amd import dist as sdist, runtime
    from soda_amd.codegen.hip import lower
    self.stencil, self.extent = stencil, tuple(extent)
    self.every, self.world = every, world
    self.slabs = [sdist.Slab(stencil, extent, world, r, every)
                  for r in range(world)]
    # rehearsal runs are scheduled by the model: every rank-thread calibrating
    # its own program on the one GPU, beside the others, makes schedules depend
    # on timing (bit-exactness must not, but reproduction should not either)
    self.progs = [runtime.Program(stencil, lower.LowerOptions(fuse=fuse),
                                  extent=s.local_extent,
                                  calibrate=True if calibrate else False)
                  for s in self.slabs]
    self.splits = 0
    self.intervals = 0

  def close(self):
    for p in self.progs:
      p.close()

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  def trial(self, fields, iterate, runs=2, spin=None, sleep=None):
    """`runs` chained runs of `iterate` iterations each from `fields` ({input
    name: global array}); returns ({output name: stitched own rows}, messages
    per rank)."""
    import torch
    from soda_amd import dist as sdist
    st = self.stencil
    spin, sleep = spin or {}, sleep or {}
    # per rank: `d[k] += 1` from 5-6 rank threads is not atomic under the GIL
    counts = [{'splits': 0, 'intervals': 0} for _ in range(self.world)]

    def rank_fn(rank, endpoint):
      slab, prog = self.slabs[rank], self.progs[rank]
      compute = torch.cuda.Stream()
      with torch.cuda.stream(compute):
        hider = sdist.StreamOverlap(0)
        src = [torch.from_numpy(
            np.ascontiguousarray(fields[n][slab.begin:slab.end])).cuda()
               for n in st.input_names]
        work = [[torch.empty_like(t) for t in src] for _ in range(2)]

        def step(dst, cur, lext, iters, **kw):
          if rank in sleep:
            time.sleep(sleep[rank])
          if rank in spin:
            torch.cuda._sleep(int(spin[rank]))
          prog.run_device([t.data_ptr() for t in dst],
                          [t.data_ptr() for t in cur], lext, iterate=iters,
                          stream=compute.cuda_stream, origin=slab.origin,
                          global_extent=slab.extent, **kw)
          counts[rank]['splits'] += prog.last_split()
          counts[rank]['intervals'] += 1

        res = sdist.run(slab, src, work[0], work[1], step, iterate, endpoint,
                        overlap=hider)
        pool = [src] + work
        for _ in range(runs - 1):
          # chained: the state's ghosts are stale, the run opens with an exchange
          others = [x for x in pool if x[0] is not res[0]]
          res = sdist.run(slab, res, others[0], others[1], step, iterate,
                          endpoint, ghosts_fresh=False, overlap=hider)
        compute.synchronize()
        hider.comm.synchronize()
        own = [r[slab.ghost_lo:slab.ghost_lo + slab.own_rows].cpu().numpy()
               for r in res]
      return own, endpoint.messages

    results = fabric.run_ranks(self.world, rank_fn)
    self.splits += sum(c['splits'] for c in counts)
    self.intervals += sum(c['intervals'] for c in counts)
    got = {o: np.concatenate([r[0][i] for r in results], axis=0)
           for i, o in enumerate(st.output_names)}
    return got, [r[1] for r in results]


def mismatches(stencil, extent, got, want, iterate, whole_grid=False,
               by_value=False):
  """Cells of the valid box of `iterate` iterations (the whole grid under
  `border: preserve`) where got and want differ, summed over the outputs.
  Floats bit for bit unless `by_value` (random programs whose values overflow:
  a NaN equals a NaN whatever its sign and payload, as in the other scans)."""
  bad = 0
  for o in stencil.output_names:
    if whole_grid:
      g, w = got[o], want[o]
    else:
      lo, hi = stencil.valid_box(extent, o, iterate)
      idx = tuple(slice(l, h) for l, h in zip(lo[::-1], hi[::-1]))
      g, w = got[o][idx], want[o][idx]
    if g.dtype.kind == 'f' and by_value:
      bad += int((~((g == w) | (np.isnan(g) & np.isnan(w)))).sum())
      continue
    if g.dtype.kind == 'f':     # bit for bit (NaN-safe, -0.0 != +0.0)
      bits = {4: np.uint32, 8: np.uint64}[g.dtype.itemsize]
      g = np.ascontiguousarray(g).view(bits)
      w = np.ascontiguousarray(w).view(bits)
    bad += int((g != w).sum())
  return bad

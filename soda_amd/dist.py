"""Multi-GPU execution: slab decomposition + halo exchange, one process per GPU.

The reference has no multi-device support at all (its only
scale-out is tiling with a replicated halo done by the generated host,
reference src/soda/codegen/frt/host.py:124-128,181-249, and cyclic interleave
over <= 4 DRAM banks, docs/data-layout.md:62-127).  This is the MI355X-native
counterpart of that host-side tiling: the LAST dimension -- the one SODA streams
-- is cut into one contiguous slab per GPU, and what the FPGA host replicates
once (the halo between tiles) is exchanged between neighbours as the iterations
advance.

Communication-avoiding schedule: xGMI P2P messages of a few hundred KiB are
latency-bound, so instead of one exchange per iteration every rank keeps
`K * radius` ghost rows per side, runs K iterations with no communication
(ghost rows decay from the outside in, own rows stay exact), then refreshes
the ghosts with ONE send/recv pair per neighbour.  No reduction collective
exists anywhere on the path.  The result on the global valid box is bit-for-bit
the single-GPU result: every cell runs the same arithmetic.

The local compute engine is injected (`step(dst, src, local_extent, iters)`):
on GPUs it is `runtime.Program.run_device`; the CPU tests drive the same
decomposition/exchange code over gloo with a CPU engine.

The transport (`dist_module`: torch.distributed or a stand-in) must order a
message on a DEVICE tensor against the HIP stream that is current when it is
posted -- RCCL (backend "nccl") does.  gloo does not: its send / recv threads
touch the GPU's memory through the CPU whenever they get to it, so a rehearsal
over gloo must stage device tensors through host buffers itself
(bench.py HostStagedP2P, tests/test_dist.py _gpu_worker).
"""
import logging
from typing import Callable, List, Optional, Sequence, Tuple

from soda_amd import core, util

_log = logging.getLogger(__name__)


class Slab:
  """Geometry of one rank's share of the grid along the last dimension."""

  def __init__(self, stencil: core.Stencil, extent: Sequence[int], world: int,
               rank: int, exchange_every: int):
    if world < 1 or not 0 <= rank < world:
      raise util.InputError('bad rank %d of %d' % (rank, world))
    if exchange_every < 1:
      raise util.InputError('exchange_every must be >= 1')

    self.stencil = stencil
    self.extent = tuple(extent)
    self.world = world
    self.rank = rank
    self.exchange_every = exchange_every
    n = self.extent[-1]
    lo, hi = stencil.radius
    self.reach_lo = -lo[-1]          # rows an iteration reads below a cell
    self.reach_hi = hi[-1]           # ... and above
    self.ghost_lo = self.reach_lo * exchange_every if rank > 0 else 0
    self.ghost_hi = self.reach_hi * exchange_every if rank < world - 1 else 0
    # own rows: as even as possible, first ranks take the remainder
    base, extra = divmod(n, world)
    self.own_begin = rank * base + min(rank, extra)
    self.own_rows = base + (1 if rank < extra else 0)
    self.own_end = self.own_begin + self.own_rows
    # every rank must reach the same verdict, or the ones that pass block in
    # the exchange while a neighbour has raised: judge the THINNEST slab of the
    # decomposition (n // world rows), not this rank's own
    need = max(self.reach_lo, self.reach_hi) * exchange_every
    if world > 1 and base < need:
      raise util.InputError(
          'slabs of %d rows (%d rows over %d ranks) are thinner than the '
          '%d-row halo; use fewer GPUs or a smaller exchange interval' %
          (base, n, world, need))
    self.begin = self.own_begin - self.ghost_lo   # first global row held
    self.end = self.own_end + self.ghost_hi
    self.rows = self.end - self.begin
    self.local_extent = self.extent[:-1] + (self.rows,)
    self.row_cells = 1
    for e in self.extent[:-1]:
      self.row_cells *= e

  @property
  def origin(self) -> Tuple[int, ...]:
    """Global position of cell 0 of this rank's arrays (what `border:
    preserve` needs to tell the grid's border from a slab edge:
    Program.run_device(..., origin=slab.origin, global_extent=slab.extent))."""
    return (0,) * (len(self.extent) - 1) + (self.begin,)

  @property
  def keep(self) -> Tuple[int, int]:
    """Local rows a run between two exchanges has to deliver: this rank's own
    ones (the ghosts are overwritten by the next exchange).  Passed as
    Program.run_device(..., keep=slab.keep), it lets every pass skip the ghost
    rows whose results can no longer reach an own row."""
    return (self.ghost_lo, self.ghost_lo + self.own_rows)

  # neighbour traffic, in local row indices: (peer, send rows, recv rows)
  def messages(self) -> List[Tuple[int, Tuple[int, int], Tuple[int, int]]]:
    out = []
    k = self.exchange_every
    if self.rank > 0:
      # lower neighbour needs my first own rows as ITS upper ghosts
      send = (self.ghost_lo, self.ghost_lo + self.reach_hi * k)
      recv = (0, self.ghost_lo)
      out.append((self.rank - 1, send, recv))
    if self.rank < self.world - 1:
      top = self.ghost_lo + self.own_rows
      send = (top - self.reach_lo * k, top)
      recv = (top, top + self.ghost_hi)
      out.append((self.rank + 1, send, recv))
    return out


def exchange(slab: Slab, tensors: Sequence, dist_module, group=None) -> None:
  """Refreshes the ghost rows of every tensor (torch tensors whose first axis
  is the last DSL dimension) from the neighbours' own rows.  One batched
  isend/irecv per call; blocks until the receives have landed."""
  if slab.world == 1:
    return
  ops = []
  for t in tensors:
    for peer, (s0, s1), (r0, r1) in slab.messages():
      if s1 > s0:
        ops.append(dist_module.P2POp(dist_module.isend, t[s0:s1], peer, group))
      if r1 > r0:
        ops.append(dist_module.P2POp(dist_module.irecv, t[r0:r1], peer, group))
  if ops:
    for req in dist_module.batch_isend_irecv(ops):
      req.wait()


class StreamOverlap:
  """Runs a rank's halo exchanges on a stream of their own so that they hide
  under the compute (GPU ranks; the counterpart of what soda_hip_group_* does
  with peer copies inside one process).  `start` orders the exchange behind
  the `sendable` event -- recorded by the previous interval's last pass as soon
  as the rows the neighbours fetch are complete -- and returns the event the
  refreshed ghosts are ready at; the engine passes both to
  Program.run_device(ghosts=..., sends=..., ghosts_ready=..., sendable=...),
  which launches the chunks next to the ghosts behind the one and fires the
  other ahead of the interior of its last pass (soda_hip_run_device_slab).
  RCCL's send/recv kernels run on RCCL's stream; `req.wait()` only orders
  the exchange stream behind them, the host does not block.

  The event chain of one rank, interval j (compute stream C, exchange stream X;
  state arrays rotate A -> B -> A ..., interval j reads cur_j, writes cur_j+1):

    X:  wait sendable(j-1) | send own edge rows of cur_j, recv ghost rows of
        cur_j | record ready(j)
    C:  first pass: interior chunks | wait ready(j) | boundary chunks
        ... middle passes (program temporaries only) ...
        last pass: boundary chunks (deliver the send rows of cur_j+1 and are
        the only launches that still write its ghost rows) | record
        sendable(j) | interior chunks

  Why nothing is overwritten while it is still read: (1) the exchange touches
  cur_j only; the compute stream writes cur_j again in interval j+1 at the
  earliest (it is that interval's output) and has waited for ready(j) -- which
  sits behind the sends AND the receives of exchange j -- in interval j, also
  when a slab has nothing to receive (one-sided reach: its first pass still
  waits for its own sends).  (2) X receives into the ghost rows of cur_j+1
  only behind sendable(j), after which no launch of interval j writes them.
  (3) The neighbour's rows arrive behind ITS sendable(j-1): complete.
  `sendable` is ONE event re-recorded every interval; X's wait is enqueued
  after the run that records it has returned, so it binds to that record.

  `sendable` only speaks for the run that recorded it.  If anything else was
  enqueued on the compute stream since (a step without this object, a timing
  loop, the caller's own kernels), call `invalidate()`: the next exchange then
  waits for everything enqueued on the compute stream so far."""

  def __init__(self, device: int = 0):
    import torch
    from soda_amd import runtime
    self._torch = torch
    self.comm = torch.cuda.Stream(device=device)
    self.sendable = runtime.Event()
    self.fence = runtime.Event()
    self.ready = [runtime.Event(), runtime.Event()]
    self.turn = 0
    self.recorded = False     # did the step just before record `sendable`?
    self._lib = runtime.library()

  def invalidate(self) -> None:
    """The compute stream has run something that did not record `sendable`."""
    self.recorded = False

  def start(self, slab: Slab, tensors: Sequence, dist_module, group=None,
            main_stream: Optional[int] = None):
    """Enqueues the exchange of `tensors`' ghost rows; returns the raw event
    handle to wait for (0 when there is nothing to exchange).  `main_stream`:
    the compute stream's raw handle (default: torch's current stream)."""
    import ctypes
    from soda_amd import runtime
    if slab.world == 1:
      return 0
    torch = self._torch
    comm_handle = ctypes.c_void_p(self.comm.cuda_stream)
    if self.recorded:
      behind = self.sendable
    else:       # nothing speaks for the rows: behind everything enqueued so far
      if main_stream is None:
        main_stream = torch.cuda.current_stream().cuda_stream
      self.fence.record(main_stream)
      behind = self.fence
    runtime.check(self._lib.soda_hip_hipstream_wait_event(
        comm_handle, behind._h), 'hipstream_wait_event')
    # consumed: only a step parameterised by step_kwargs renews it
    self.recorded = False
    ops = []
    with torch.cuda.stream(self.comm):
      for t in tensors:
        for peer, (s0, s1), (r0, r1) in slab.messages():
          if s1 > s0:
            ops.append(dist_module.P2POp(dist_module.isend, t[s0:s1], peer,
                                         group))
          if r1 > r0:
            ops.append(dist_module.P2POp(dist_module.irecv, t[r0:r1], peer,
                                         group))
      if ops:
        for req in dist_module.batch_isend_irecv(ops):
          req.wait()          # stream order only
    ready = self.ready[self.turn]
    self.turn ^= 1
    ready.record(self.comm.cuda_stream)
    return ready.handle()

  def step_kwargs(self, slab: Slab, ghosts_ready: int):
    """Keyword arguments for Program.run_device of the interval that follows
    `start` (ghosts_ready = what it returned; 0: the ghosts are fresh).  The
    caller MUST issue that run next on the compute stream: the exchange after
    it is ordered behind the `sendable` this run records."""
    self.recorded = True
    k = slab.exchange_every
    return dict(
        keep=slab.keep,
        ghosts=(slab.ghost_lo, slab.ghost_hi) if ghosts_ready else (0, 0),
        sends=(slab.reach_hi * k if slab.rank > 0 else 0,
               slab.reach_lo * k if slab.rank < slab.world - 1 else 0),
        ghosts_ready=ghosts_ready, sendable=self.sendable.handle())


def run(slab: Slab, src: Sequence, work_a: Sequence, work_b: Sequence,
        step: Callable[..., None],
        iterate: int, dist_module, group=None, ghosts_fresh: bool = True,
        overlap: Optional[StreamOverlap] = None):
  """Advances the program `iterate` iterations.  `src` holds the inputs (own
  rows + ghosts); `work_a` / `work_b` are same-shaped work arrays the state
  ping-pongs through (`work_b` is only touched when more than one exchange
  interval is needed).  Returns the list that holds the result.
  `ghosts_fresh`: the ghosts of `src` already hold neighbour data (true right
  after slicing them out of the global input); `src` is then never written.
  With ghosts_fresh=False the run opens with an exchange that OVERWRITES the
  ghost rows of `src` with the neighbours' rows (bench.py's chained steps rely
  on exactly that): do not pass a view of data you need unchanged.
  `overlap`: hide the exchanges under the compute; `step` is then called as
  step(dst, src, extent, iters, **overlap.step_kwargs(...))."""
  cur, nxt = list(src), list(work_a)
  spare = list(work_b)
  done = 0
  fresh = ghosts_fresh
  while done < iterate:
    k = min(slab.exchange_every, iterate - done)
    if overlap is not None:
      token = 0 if fresh else overlap.start(slab, cur, dist_module, group)
      step(nxt, cur, slab.local_extent, k, **overlap.step_kwargs(slab, token))
      if done == 0:
        cur, nxt = nxt, spare
      else:
        cur, nxt = nxt, cur
      done += k
      fresh = False
      continue
    if not fresh:
      exchange(slab, cur, dist_module, group)
    step(nxt, cur, slab.local_extent, k)
    if done == 0:
      cur, nxt = nxt, spare          # src drops out of the rotation
    else:
      cur, nxt = nxt, cur
    done += k
    fresh = False
  return cur


def auto_exchange_every(stencil: core.Stencil, extent: Sequence[int],
                        world: int, iterate: int, multiple_of: int = 1) -> int:
  """Iterations between halo exchanges: as many as keep the ghost rows (both
  sides together) at or below a quarter of a slab.  jacobi2d 8192 rows on 8
  GPUs: 128 >= 100, so the halo travels once with the input and a 100-iteration
  step needs no exchange; heat3d 512 planes on 8 GPUs: 8 (16 ghost planes per
  64-plane slab)."""
  if world <= 1:
    return iterate
  lo, hi = stencil.radius
  reach = max(1, -lo[-1] + hi[-1])
  own = max(1, extent[-1] // world)
  k = max(1, min(iterate, own // (4 * reach)))
  if multiple_of > 1 and k < iterate:
    k = max(multiple_of, k // multiple_of * multiple_of)
  return k


def planned_exchange_every(stencil: core.Stencil, extent: Sequence[int],
                           world: int, iterate: int, opts=None,
                           multiple_of: int = 1, overlap: bool = True) -> int:
  """Iterations between halo exchanges by the LIBRARY's cost choice -- the pass
  times of the slab extent a candidate implies against a transfer model, the
  same function the one-process group uses (soda_hip_group_plan, pure: no GPU)
  -- instead of the quarter-slab rule above.  Needs the kernels' register
  counts, hence a JIT build of the module (hiprtc; cached); any failure falls
  back to auto_exchange_every.  jacobi2d 8192^2 x 100 on 2 / 4 / 8 GPUs: 100
  either way; x 1000 on 8: 169 (the rule: 117); heat3d 512^3 x 50 on 8: 10
  (8)."""
  import ctypes
  fallback = auto_exchange_every(stencil, extent, world, iterate, multiple_of)
  if world <= 1:
    return iterate
  try:
    from soda_amd import runtime
    from soda_amd.codegen.hip import lower
    lib = runtime.library()
    reach_lo, reach_hi = stencil.reach_along(stencil.dim - 1)
    own = extent[-1] // world
    local = tuple(extent[:-1]) + (own,)
    every = ctypes.c_int32(0)
    for _ in range(4):          # the slab extent depends on K and K on it
      o = runtime.resolve_options(stencil, opts or lower.LowerOptions(), local)
      mod = lower.lower(stencil, o)
      code = runtime.compile_source(mod.source, '%s.hip' % stencil.app_name)
      plan = runtime.make_plan(mod, runtime.kernel_resources(code))
      desc = runtime.GroupDesc()
      desc.num_slabs = world
      for i, e in enumerate(extent):
        desc.extent[i] = int(e)
      desc.reach_lo, desc.reach_hi = reach_lo, reach_hi
      desc.iterate, desc.exchange_every = iterate, 0
      desc.flags = 0 if overlap else runtime.GROUP_NO_OVERLAP
      runtime.check(lib.soda_hip_group_plan(ctypes.byref(plan),
                                            ctypes.byref(desc),
                                            ctypes.byref(every)), 'group_plan')
      ghosts = (reach_lo + reach_hi if world > 2 else
                max(reach_lo, reach_hi)) * every.value
      grown = tuple(extent[:-1]) + (own + ghosts,)
      if grown == local:
        break
      local = grown
    k = int(every.value)
    if k < 1:
      return fallback
    if multiple_of > 1 and multiple_of <= k < iterate:
      k = k // multiple_of * multiple_of
    return min(k, iterate)
  except Exception as e:     # noqa: BLE001 -- a planning aid must not stop a run
    # ... but a broken planner must not degrade K silently for ever either
    _log.warning('planned_exchange_every: planner failed (%s: %s); exchanging '
                 'every %d iterations instead', type(e).__name__, e, fallback)
    return fallback


def rounds(iterate: int, exchange_every: int) -> int:
  return -(-iterate // exchange_every)

"""In-process stand-in for torch.distributed's point-to-point calls (TEST
INFRASTRUCTURE): N ranks are N threads of one process, a message is a tensor
copy handed over through a queue.  `Fabric.endpoint(rank)` quacks like the
`dist_module` argument of soda_amd.dist.exchange/run (P2POp, isend, irecv,
batch_isend_irecv, barrier), so a multi-rank slab run -- every rank executing
the REAL decomposition, exchange schedule and kernels -- can be rehearsed on
the single GPU of a test box, where RCCL needs one device per rank."""
import os
import queue
import threading
from typing import List

TIMEOUT_S = 300

# Round-3 defect of THIS file, kept reproducible (tools/flake_loop.py): the
# staging copy of a message is allocated on the sender's stream and read on the
# receiver's.  Dropping the last reference right after ENQUEUEING the read hands
# the block back to PyTorch's caching allocator, which knows of the sender's
# stream only: the sender's next message may be staged in the same block while
# the receiver's copy is still queued behind the events of a slower neighbour
# -- the receiver then reads the NEXT interval's rows.  (The driver's round-3
# GPU run: test_exchange_hidden_under_the_compute, 4 ranks, two messages per
# exchange.)  SODA_FABRIC_UNSAFE_LIFETIME=1 brings the defect back.
UNSAFE_LIFETIME = bool(os.environ.get('SODA_FABRIC_UNSAFE_LIFETIME'))


class _Recv:

  def __init__(self, q: 'queue.Queue', tensor):
    self.q, self.tensor = q, tensor

  def wait(self) -> None:
    payload, ready = self.q.get(timeout=TIMEOUT_S)
    if ready is not None:
      # a device tensor copied on the SENDER's current stream: order the
      # receiver's stream behind that copy (what RCCL's recv does by itself)
      import torch
      torch.cuda.current_stream().wait_event(ready)
    self.tensor.copy_(payload)
    if ready is not None and not UNSAFE_LIFETIME:
      # the staging block stays the receiver stream's until this copy has run
      import torch
      payload.record_stream(torch.cuda.current_stream())


class _Done:

  def wait(self) -> None:
    pass


class Endpoint:
  isend = 'isend'
  irecv = 'irecv'

  def __init__(self, fabric: 'Fabric', rank: int):
    self.fabric, self.rank = fabric, rank
    self.messages = 0
    self.bytes = 0

  @staticmethod
  def P2POp(op, tensor, peer, group=None):
    return (op, tensor, peer)

  def batch_isend_irecv(self, ops) -> List:
    reqs = []
    for op, tensor, peer in ops:      # sends first: they never block
      if op == self.isend:
        payload, ready = tensor.clone(), None
        if payload.is_cuda:
          import torch
          ready = torch.cuda.Event()
          ready.record(torch.cuda.current_stream())
        self.fabric.q[(self.rank, peer)].put((payload, ready))
        self.messages += 1
        self.bytes += tensor.numel() * tensor.element_size()
        reqs.append(_Done())
    for op, tensor, peer in ops:
      if op == self.irecv:
        reqs.append(_Recv(self.fabric.q[(peer, self.rank)], tensor))
    return reqs

  def barrier(self) -> None:
    self.fabric.barrier_obj.wait(timeout=TIMEOUT_S)

  def get_world_size(self) -> int:
    return self.fabric.world

  def get_rank(self) -> int:
    return self.rank


class Fabric:

  def __init__(self, world: int):
    self.world = world
    self.q = {(s, d): queue.Queue()
              for s in range(world) for d in range(world) if s != d}
    self.barrier_obj = threading.Barrier(world)

  def endpoint(self, rank: int) -> Endpoint:
    return Endpoint(self, rank)


def run_ranks(world: int, fn) -> list:
  """Runs fn(rank, endpoint) on `world` threads; returns the results in rank
  order, re-raising the first exception."""
  fabric = Fabric(world)
  results = [None] * world
  errors = []

  def body(rank):
    try:
      results[rank] = fn(rank, fabric.endpoint(rank))
    except BaseException as e:  # noqa: surface it in the caller's thread
      errors.append(e)
      fabric.barrier_obj.abort()

  threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  if errors:
    raise errors[0]
  return results

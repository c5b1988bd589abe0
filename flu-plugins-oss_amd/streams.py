"""Multi-GPU layout of the render path: independent subtitle
streams shard one per GPU/rank, with NO data-path collective; only when a single
ordered output is required are finished frames gathered to rank 0 (RCCL over
xGMI on the GPUs; the same code runs over gloo on CPU tensors in the tests).

Host logic only -- nothing here touches pixels.
"""
import torch
import torch.distributed as dist


def streams_of_rank(n_streams, rank, world):
    """Contiguous block partition of stream ids over ranks (weak scaling keeps
    n_streams == world: one stream per GPU)."""
    per, rem = divmod(n_streams, world)
    lo = rank * per + min(rank, rem)
    return list(range(lo, lo + per + (1 if rank < rem else 0)))


def state_schedule(n_states, rank, step, batch, world=8):
    """Scene states a rank renders in `step`: every rank walks its own stream's
    states, phase-shifted by rank (spread evenly over the document) so that
    ranks replaying the same recorded document are never in lock step."""
    base = (step * batch + (rank * n_states) // max(world, 1)) % n_states
    return [(base + j) % n_states for j in range(batch)]


def gather_frames(frame, dst=0, group=None):
    """Gather one finished frame (flat uint8 tensor) of every rank to `dst`, in
    rank order: the "single ordered output" case.  Implemented as a collective
    gather (grouped send/recv under RCCL: each peer uses its direct xGMI link to
    rank 0).  Returns the list of frames on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = [torch.empty_like(frame) for _ in range(world)] if rank == dst else None
    dist.gather(frame, out, dst=dst, group=group)
    return out


class FrameGatherer:
    """The same gather, overlapped with rendering: each call issues the collective
    asynchronously (under RCCL it runs on the communicator's own stream, ordered
    after the render work already queued, while the caller's stream goes on to the
    next launch) into one of `slots` pre-allocated destination sets on `dst`.  A
    slot is waited for only when it comes round again, so up to `slots` gathers
    are in flight.  The caller must not overwrite a submitted source block before
    `slots` further submissions or flush() (bench.py's output ring is deeper)."""

    def __init__(self, block_bytes, device, slots=2, dst=0, group=None):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dst, self.group, self.slots = dst, group, slots
        self.bufs = [[torch.empty(block_bytes, dtype=torch.uint8, device=device)
                      for _ in range(self.world)] if self.rank == dst else None
                     for _ in range(slots)]
        self.work = [None] * slots
        self.n = 0

    def submit(self, block):
        k = self.n % self.slots
        self.n += 1
        if self.work[k] is not None:
            self.work[k].wait()
        self.work[k] = dist.gather(block, self.bufs[k], dst=self.dst, group=self.group,
                                   async_op=True)
        return self.bufs[k]

    def flush(self):
        for k, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[k] = None


def max_over_ranks(seconds, device):
    """Timing convention of bench.py: the slowest rank defines the step time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

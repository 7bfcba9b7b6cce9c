"""Clip-batch data parallelism: one process per GPU, clips sharded across ranks, logits
all-gathered (the only collective on the inference path).

The ECO forward is per-sample (inference BN uses stored statistics), so the clip batch is a
set of independent units: rank r of W owns clips [r*B/W, (r+1)*B/W), weights are replicated
(same seed / same file on every rank), and a step ends with ONE collective that mirrors the
reference's ``Gather`` layer (caffe_3d/src/caffe/layers/gather_layer.cpp:19-55: top shape[0] =
bottom shape[0] * world, ``MPI_Allgather`` through util/channel.cpp:100-105): an all-gather of
the ``[B/W, num_classes]`` fp32 logits.  On MI355X that is ``ncclAllGather`` (RCCL, backend
"nccl") over xGMI -- 51 KB per rank at B/W=32: latency-bound, far from the per-link ring bound.
On CPU (tests) the same code runs over gloo.  No all-reduce, no halo exchange: the temporal
axis is never sharded.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def init_process_group(backend: Optional[str] = None, device=None) -> Tuple[int, int]:
    """Join the job described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run sets them).
    Returns (rank, world).  backend defaults to "nccl" (= RCCL on ROCm) when a device is given,
    else "gloo"."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if device is not None else "gloo"
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(total_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Clips [start, stop) owned by `rank`; the batch must split evenly (the reference's Gather
    layer likewise assumes equal per-rank counts)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} / world {world}")
    if total_clips % world:
        raise ValueError(f"clip batch {total_clips} is not divisible by the {world} ranks")
    per = total_clips // world
    return rank * per, (rank + 1) * per


def all_gather_logits(local, out=None):
    """[B/W, C] per rank -> [B, C] on every rank, rank-major (Gather layer semantics).

    Ordering on the GPU: the collective runs on the communicator's own stream.  It is issued asynchronously and
    ``work.wait()`` then makes the CURRENT stream -- the one the engine launches on -- wait for it (a stream-side
    wait, the host does not block): the next step's launches, which overwrite ``local`` (the engine's logits
    buffer), are ordered behind the all-gather that reads it, and the collective itself was ordered behind this
    step's launches when it was enqueued.  At 15 ms steps the window was never hit; at the 1.3 ms online step with
    N > 1 it could be."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=True)
    work.wait()
    return out

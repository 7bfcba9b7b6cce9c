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
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple


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


# ---- host-side placement: one rank per GPU, each on cores next to ITS GPU ------------------------------------------------
# The reference is one MPI process per GPU with no pinning at all (tools/caffe.cpp, util/mpi_functions.cpp); on a
# two-socket, SMT host eight unpinned Python launch loops (or an even slice of the sorted logical-CPU list, which gives
# ranks r and r+4 the two hyper-threads of the same cores and puts half the ranks on the far socket) is the first suspect
# when the clips/s curve bends.  Everything here reads plain sysfs files so that it can be unit-tested on a fake tree.

def parse_cpulist(text: str) -> List[int]:
    """Kernel cpulist syntax ("0-3,8-11,16") -> sorted CPU numbers."""
    out: Set[int] = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.update(range(int(lo), int(hi) + 1))
        else:
            out.add(int(part))
    return sorted(out)


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_local_cpus(pci_bus_id: Optional[str], sysfs: str = "/sys") -> Optional[List[int]]:
    """CPUs of the NUMA node the GPU at `pci_bus_id` ("0000:c1:00.0") hangs off: /sys/bus/pci/devices/<id>/local_cpulist,
    or the node's cpulist through numa_node.  None when the platform does not say (numa_node -1, file missing)."""
    if not pci_bus_id:
        return None
    base = os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower())
    node = _read(os.path.join(base, "numa_node"))
    if node is not None and node.lstrip("-").isdigit() and int(node) >= 0:
        txt = _read(os.path.join(sysfs, "devices", "system", "node", f"node{int(node)}", "cpulist"))
        if txt:
            return parse_cpulist(txt)
    txt = _read(os.path.join(base, "local_cpulist"))
    if txt and (node is None or not node.startswith("-")):
        return parse_cpulist(txt)
    return None


def physical_cores(cpus: Iterable[int], sysfs: str = "/sys") -> List[Tuple[int, ...]]:
    """Group logical CPUs into physical cores by topology/thread_siblings_list (restricted to `cpus`), ordered by
    their lowest CPU number.  A CPU without topology information is a core of its own."""
    cpus = sorted(set(cpus))
    allowed = set(cpus)
    seen: Set[int] = set()
    cores: List[Tuple[int, ...]] = []
    for c in cpus:
        if c in seen:
            continue
        txt = _read(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list"))
        sib = [s for s in (parse_cpulist(txt) if txt else [c]) if s in allowed] or [c]
        if c not in sib:
            sib.append(c)
        sib = tuple(sorted(set(sib) - seen))
        seen.update(sib)
        cores.append(sib)
    return cores


def plan_rank_cpus(gpu_pci_ids: Sequence[Optional[str]], allowed: Iterable[int], sysfs: str = "/sys") -> List[List[int]]:
    """CPU set of every local rank (rank r drives GPU r).  Ranks whose GPUs sit on the same NUMA node share that node's
    allowed PHYSICAL cores in equal contiguous runs -- a core's SMT siblings always go to the same rank, so no two ranks
    ever time-share a core; a GPU without NUMA information takes part in an even split of whatever cores are left over
    (all of them when no GPU has any).  Deterministic: every rank computes the same plan."""
    allowed = sorted(set(allowed))
    n = len(gpu_pci_ids)
    local = [gpu_local_cpus(p, sysfs) for p in gpu_pci_ids]
    groups: Dict[Tuple[int, ...], List[int]] = {}
    for r in range(n):
        cand = tuple(c for c in (local[r] or []) if c in allowed)
        groups.setdefault(cand, []).append(r)          # () = "no information" (or none of its CPUs allowed)
    plan: List[List[int]] = [[] for _ in range(n)]
    claimed: Set[int] = set()
    overflow: List[int] = []                           # ranks of nodes with fewer allowed cores than ranks (round-4 advisor:
    for cand, ranks in groups.items():                 # adding them to `groups` here changed the dict under its own iteration)
        if not cand:
            continue
        cores = physical_cores(cand, sysfs)
        if len(cores) < len(ranks):                    # fewer cores than ranks on this node: they join the even split below
            overflow.extend(ranks)
            continue
        for i, r in enumerate(ranks):
            mine = cores[len(cores) * i // len(ranks):len(cores) * (i + 1) // len(ranks)]
            plan[r] = sorted(c for core in mine for c in core)
            claimed.update(plan[r])
    rest = sorted(groups.get((), []) + overflow)
    if rest:
        free = [c for c in allowed if c not in claimed] or allowed
        cores = physical_cores(free, sysfs)
        for i, r in enumerate(rest):
            mine = cores[len(cores) * i // len(rest):len(cores) * (i + 1) // len(rest)] if len(cores) >= len(rest) else cores
            plan[r] = sorted(c for core in mine for c in core)
    return plan


def pin_rank(local_rank: int, gpu_pci_ids: Sequence[Optional[str]], sysfs: str = "/sys") -> Optional[dict]:
    """sched_setaffinity this process to its share of plan_rank_cpus; returns what was done (for the bench line) or None
    when the platform has no affinity calls."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    plan = plan_rank_cpus(gpu_pci_ids, allowed, sysfs)
    mine = plan[local_rank] or allowed
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    numa = _read(os.path.join(sysfs, "bus", "pci", "devices", (gpu_pci_ids[local_rank] or "").lower(), "numa_node"))
    return {"cpus": len(mine), "first_cpu": mine[0], "last_cpu": mine[-1], "physical_cores": len(physical_cores(mine, sysfs)),
            "gpu_pci": gpu_pci_ids[local_rank], "gpu_numa_node": int(numa) if numa and numa.lstrip("-").isdigit() else None}

"""Batch sharding across the GPUs of one node: one process per GPU, no data-path collective.

Every feeder instance's transition depends only on its own state and action (reference
environments/grid_env.py:410-619 has no cross-environment term), so a batch of B_total
instances splits into contiguous blocks, rank r owning ``shard_range(B_total, r, world)``.
Seeds and RNG counters are keyed by the GLOBAL instance index, so results do not depend on the
number of ranks.  The only exchange is optional and happens after the step: an all-gather of
the observation block so that every rank (or a host-side learner) sees all observations --
RCCL over xGMI through libgridstep (``transport="rccl"``, device buffers, equal shards), the same device-side
exchange with every rank's shard held by ONE process (``LoopbackShards``: device-to-device copies where RCCL would
cross xGMI -- the rehearsal of N ranks on a box with fewer GPUs), or host
arrays (``transport="host"``, uneven shards allowed) through a framework-free file rendezvous
(``rendezvous.FileRendezvous``) or any ``torch.distributed`` process group (what the world_size-2
gloo tests exercise).  Nothing here imports torch unless a torch group is what the caller hands in.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import numpy as np

from .env import BatchedGridEnvironment


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of rank's contiguous block; the first ``total % world`` ranks get one extra."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def instance_seeds(global_seed: int, start: int, stop: int) -> np.ndarray:
    """Seed of global instance b is ``global_seed + b`` (uint64 wrap-around)."""
    return (np.uint64(global_seed) + np.arange(start, stop, dtype=np.uint64)).astype(np.uint64)


def block_checksum(block: np.ndarray) -> bytes:
    """SHA-256 of a block of observations as the bytes it holds (float64, row-major)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(block, dtype=np.float64).tobytes()).digest()


def verify_gathered_block(full: np.ndarray, own: np.ndarray, rank: int, world: int, all_gather_bytes) -> dict:
    """Does the block an all-gather left on THIS rank hold, in slot r, exactly what rank r computed?  Every rank checksums
    its own observations, the checksums travel beside the data path (``all_gather_bytes``: a callable that all-gathers a
    ``bytes`` object over the ranks in rank order, e.g. ``FileRendezvous.all_gather_bytes``), and every rank compares the
    checksum of every slot of ITS gathered block with the owner's.  The per-rank verdicts are gathered the same way, so all
    ranks return the same record: ``gather_verified`` is True only if every slot matched on every rank."""
    rows = own.shape[0]
    if full.shape[0] != world * rows:
        raise ValueError(f"gathered block has {full.shape[0]} rows, expected {world} x {rows}")
    sums = all_gather_bytes(block_checksum(own))
    bad = [r for r in range(world) if block_checksum(full[r * rows:(r + 1) * rows]) != sums[r]]
    verdicts = all_gather_bytes(bytes([0 if r in bad else 1 for r in range(world)]))
    table = [list(v) for v in verdicts]                      # table[q][r]: rank q found slot r equal to rank r's block
    return {"gather_verified": all(all(row) for row in table),
            "slots_matching_per_rank": [int(sum(row)) for row in table],
            "own_slot_is_own_block": bool(rank not in bad),
            "distinct_shard_checksums": len(set(sums)),
            "how": "SHA-256 of each rank's own [B][obs_dim] block, exchanged beside the data path; every rank compares every slot of its gathered block"}


def host_all_gather(local: np.ndarray, total: int, rank: int, world: int, group: Any = None) -> np.ndarray:
    """All-gather of row blocks on host arrays; shards may be uneven.  Returns the [total, ...] array in global order.
    ``group``: a ``rendezvous.FileRendezvous`` (no framework at all), or a ``torch.distributed`` process group / None for
    the default one (any backend that takes CPU tensors -- what the world_size-2 gloo tests exercise)."""
    from .rendezvous import FileRendezvous
    counts = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    if isinstance(group, FileRendezvous):
        parts = group.all_gather_array(np.ascontiguousarray(local))
        if [len(p) for p in parts] != counts:
            raise ValueError("shard sizes do not match shard_range()")
        return np.concatenate(parts, axis=0)
    import torch
    import torch.distributed as dist
    width = int(np.prod(local.shape[1:], dtype=np.int64)) if local.ndim > 1 else 1
    mx = max(counts)
    pad = np.zeros((mx, width), dtype=local.dtype)
    pad[:counts[rank]] = local.reshape(counts[rank], width)
    outs = [torch.empty((mx, width), dtype=torch.from_numpy(pad).dtype) for _ in range(world)]
    dist.all_gather(outs, torch.from_numpy(pad), group=group)
    full = np.concatenate([outs[r].numpy()[:counts[r]] for r in range(world)], axis=0)
    return full.reshape((total,) + tuple(local.shape[1:]))


class ShardedGridEnvironment:
    """This rank's block of a ``global_num_envs``-instance batched environment.

    ``transport`` is "rccl" (all-gather of device-resident observations through libgridstep;
    needs equal shards and a 128-byte RCCL unique id shared by the ranks), "loopback" (a member of a
    ``LoopbackShards`` group) or "host".
    """

    def __init__(self, feeder: Any, global_num_envs: int, rank: int, world: int, device: Optional[int] = None,
                 transport: str = "host", group: Any = None, **env_kwargs: Any) -> None:
        self.global_num_envs, self.rank, self.world = int(global_num_envs), int(rank), int(world)
        self.start, self.stop = shard_range(self.global_num_envs, self.rank, self.world)
        self.transport, self.group = transport, group
        if transport in ("rccl", "loopback") and self.global_num_envs % self.world != 0:
            raise ValueError("the RCCL all-gather needs equal shards (global_num_envs % world == 0)")
        self.env = BatchedGridEnvironment(feeder, num_envs=self.stop - self.start,
                                          device=self.rank if device is None else device,
                                          first_instance=self.start, **env_kwargs)
        self._comm = False

    def init_rccl(self, unique_id: Any = None) -> None:
        """``unique_id``: the 128 bytes of rank 0's ``Handle.comm_unique_id()``, handed round by the launcher -- or a
        ``rendezvous.FileRendezvous``, through which rank 0 creates and broadcasts it."""
        from .rendezvous import FileRendezvous
        from ._lib import Handle
        if isinstance(unique_id, FileRendezvous):
            unique_id = unique_id.broadcast_bytes(Handle.comm_unique_id() if self.rank == 0 else None)
        self.env.handle.comm_init(unique_id, self.rank, self.world)
        self._comm = True

    def reset(self, seed: int = 0):
        return self.env.reset(seed=instance_seeds(seed, self.start, self.stop))

    def step(self, local_actions):
        return self.env.step(local_actions)

    def gather_observations(self, local_obs: Optional[np.ndarray] = None) -> np.ndarray:
        """[global_num_envs, obs_dim] on every rank."""
        if self.transport == "rccl":
            if not self._comm:
                raise RuntimeError("init_rccl() first")
            return self.env.handle.allgather_obs(to_host=True)
        if self.transport == "loopback":
            raise RuntimeError("a loopback member gathers through its LoopbackShards group")
        if local_obs is None:
            raise ValueError("host transport gathers the array it is given")
        return host_all_gather(local_obs, self.global_num_envs, self.rank, self.world, self.group)

    def close(self) -> None:
        if self._comm:
            self.env.handle.comm_destroy()
        self.env.close()


class LoopbackShards:
    """All ``world`` shards of a ``global_num_envs``-instance environment in THIS process (``gs_comm_init_loopback``):
    shard r is exactly what rank r of an N-GPU job builds (``first_instance = r * B``, global-index seeds), and the
    observation exchange runs the RCCL transport's device code with in-process copies in place of ``ncclAllGather``.
    For rehearsing / testing N ranks on fewer GPUs; ``devices`` maps shard -> device (default: all on device 0)."""

    def __init__(self, feeder: Any, global_num_envs: int, world: int, devices: Optional[list] = None, **env_kwargs: Any) -> None:
        if global_num_envs % world != 0:
            raise ValueError("the device all-gather needs equal shards (global_num_envs % world == 0)")
        self.global_num_envs, self.world = int(global_num_envs), int(world)
        self.shards = [ShardedGridEnvironment(feeder, global_num_envs, r, world, device=(devices[r] if devices else 0),
                                              transport="loopback", **env_kwargs) for r in range(world)]
        from ._lib import Handle
        Handle.comm_init_loopback([s.env.handle for s in self.shards])
        for s in self.shards:
            s._comm = True

    @property
    def handles(self):
        return [s.env.handle for s in self.shards]

    def reset(self, seed: int = 0):
        return [s.reset(seed) for s in self.shards]

    def step(self, global_actions):
        """``global_actions`` [global_num_envs, action_dim]; every shard steps its block.  Returns the shards' step tuples."""
        a = np.asarray(global_actions)
        return [s.step(a[s.start:s.stop]) for s in self.shards]

    def gather_observations(self, to_host: bool = True):
        """One exchange for every member (``gs_allgather_obs_shards``); [global_num_envs, obs_dim] from shard 0's block."""
        from ._lib import Handle
        return Handle.allgather_obs_shards(self.handles, to_host=to_host)

    def close(self) -> None:
        for s in self.shards:
            s.close()

"""Observation sharding across GPUs (one process per GPU, torch.distributed for the exchange).

The reference is single-process; this is new.  Observations are point-major, so a cut
at a point boundary gives every rank whole points: V_p, g_p and the point's Schur contribution live on
exactly one rank, camera parameters are replicated, and the only data-path collective is an all-reduce
(sum) of camera-sized buffers -- the 27 C doubles of [U | g_c] once per outer iteration and the 6 C
doubles of the implicit Schur product once per PCG iteration -- plus a few scalars.  The C++ solver
calls back into :class:`Exchange` at those points; ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests) performs the reduction in place on a torch tensor that the
library uses as its exchange arena.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .synthetic import BAProblem, make_problem


@dataclasses.dataclass
class Shard:
    rank: int
    world: int
    point_begin: int          # global point range [point_begin, point_end)
    point_end: int
    obs_begin: int            # global observation range (point-major order)
    obs_end: int


def partition_points(point_indices, n_points: int, world: int) -> list[Shard]:
    """Cut the point-major observation list into ``world`` contiguous shards at point boundaries,
    balancing observations.  ``point_indices`` must be non-decreasing."""
    pi = np.asarray(point_indices, dtype=np.int64)
    if pi.size and np.any(np.diff(pi) < 0):
        raise ValueError("point_indices must be point-major (non-decreasing) to shard")
    ptr = np.zeros(n_points + 1, dtype=np.int64)
    np.add.at(ptr, pi + 1, 1)
    ptr = np.cumsum(ptr)
    n_obs = int(pi.size)
    shards, p0 = [], 0
    for r in range(world):
        if r == world - 1:
            p1 = n_points
        else:
            target = (n_obs * (r + 1)) // world
            p1 = int(np.searchsorted(ptr, target, side="left"))
            p1 = min(max(p1, p0), n_points)
        shards.append(Shard(r, world, p0, p1, int(ptr[p0]), int(ptr[p1])))
        p0 = p1
    return shards


def shard_problem(pb: BAProblem, shard: Shard) -> BAProblem:
    """Local problem of one rank: all cameras, its own points (re-indexed from 0) and observations."""
    C = pb.n_cameras
    o0, o1, p0, p1 = shard.obs_begin, shard.obs_end, shard.point_begin, shard.point_end
    x0 = np.concatenate([pb.x0[:6 * C], pb.x0[6 * C + 3 * p0:6 * C + 3 * p1]])
    xt = np.concatenate([pb.x_true[:6 * C], pb.x_true[6 * C + 3 * p0:6 * C + 3 * p1]])
    return BAProblem(C, p1 - p0, pb.camera_indices[o0:o1].copy(), pb.point_indices[o0:o1] - p0,
                     pb.points_2d[o0:o1].copy(), pb.K, x0, xt)


def merge_solutions(x_locals, shards, n_cameras: int, n_points: int) -> np.ndarray:
    """Inverse of :func:`shard_problem` for the parameter vector (cameras are identical on all ranks)."""
    x = np.empty(6 * n_cameras + 3 * n_points)
    x[:6 * n_cameras] = x_locals[0][:6 * n_cameras]
    for xl, s in zip(x_locals, shards):
        x[6 * n_cameras + 3 * s.point_begin:6 * n_cameras + 3 * s.point_end] = xl[6 * n_cameras:]
    return x


def make_sharded_problem(n_cameras, n_points_per_rank, n_obs_per_rank, rank, world, seed=0) -> BAProblem:
    """Weak-scaling workload of bench.py: rank r generates its own points/observations (seeded by r),
    all ranks the same cameras and the same perturbed camera start."""
    return make_problem(n_cameras, n_points_per_rank, n_obs_per_rank, seed=1000 * (seed + 1) + rank,
                        camera_seed=seed)


class Exchange:
    """Registers a torch tensor as the library's exchange arena and serves its all-reduce callback.

    ``backend`` needs ``exchange_doubles()`` and ``set_exchange(ptr, n, callback, n_obs_total)``
    (sfmba.Backend).  The reduction runs on torch's current stream; make it the stream the backend
    launches on (``backend.set_stream(torch.cuda.current_stream().cuda_stream)``) so that kernels and
    collectives are ordered without host synchronisation.
    """

    def __init__(self, backend, n_obs_local: int, group=None, device=None):
        import torch
        import torch.distributed as td
        self._td, self._torch = td, torch
        self.group = group
        self.n_calls = 0
        self.n_doubles = 0
        if device is None:
            device = "cuda" if td.get_backend(group) == "nccl" else "cpu"
        n = int(backend.exchange_doubles())
        self.arena = torch.zeros(n, dtype=torch.float64, device=device)
        tot = torch.tensor([float(n_obs_local)], dtype=torch.float64, device=device)
        td.all_reduce(tot, op=td.ReduceOp.SUM, group=group)
        self.n_obs_total = int(round(float(tot.item())))
        self._base = self.arena.data_ptr()
        backend.set_exchange(self._base, n, self._callback, self.n_obs_total)

    def _callback(self, dev_ptr: int, count: int, op: int) -> None:
        off = (dev_ptr - self._base) // 8
        if off < 0 or off + count > self.arena.numel() or (dev_ptr - self._base) % 8:
            raise ValueError("all-reduce request outside the exchange arena")
        rop = self._td.ReduceOp.SUM if op == 0 else self._td.ReduceOp.MAX
        self._td.all_reduce(self.arena[off:off + count], op=rop, group=self.group)
        self.n_calls += 1
        self.n_doubles += count


class NativeComm:
    """The fast path: libsfmba.so all-reduces its own arena with RCCL (ncclAllReduce enqueued on the
    solver's stream from C++), so no Python and no host synchronisation sit between a sweep and its
    collective.  ``torch.distributed`` is only the bootstrap channel for the 128-byte RCCL id and the
    observation count (any backend)."""

    def __init__(self, backend, n_obs_local: int, group=None):
        import torch
        import torch.distributed as td
        rank, world = td.get_rank(group), td.get_world_size(group)
        dev = "cuda" if td.get_backend(group) == "nccl" else "cpu"
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = torch.frombuffer(bytearray(backend.comm_unique_id()), dtype=torch.uint8).to(dev)
        td.broadcast(uid, src=0, group=group)
        tot = torch.tensor([float(n_obs_local)], dtype=torch.float64, device=dev)
        td.all_reduce(tot, op=td.ReduceOp.SUM, group=group)
        self.n_obs_total = int(round(float(tot.item())))
        self.rank, self.world = rank, world
        backend.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world, self.n_obs_total)


class DirectLink:
    """Latency path for the solver's collectives inside one node: every rank maps its peers' staging
    buffers (hipIpc over ``torch.distributed`` as the bootstrap channel) and ``libsfmba.so`` all-reduces
    with one small kernel per collective over xGMI instead of calling RCCL (``include/sfmba.h``:
    ``sfmba_p2p_*``).  Create it AFTER :class:`NativeComm` or :class:`Exchange` (which stay as the
    fallback transport and for vectors larger than a slot).  ``active`` is False when any rank could not
    map a peer or failed the self-test: then every rank has detached and the previous transport serves.
    ``SFMBA_P2P=0`` disables it.  ``allow_single``: also at world size 1 (the rank maps only its own buffer) -- what
    ``bench.py --force-exchange`` uses to time the sharded code path of one rank's share on one GPU."""

    def __init__(self, backend, group=None, allow_single=False):
        import os
        import torch
        import torch.distributed as td
        self.backend = backend
        self.active = False
        rank, world = td.get_rank(group), td.get_world_size(group)
        if (world < 2 and not allow_single) or world > 16 or os.environ.get("SFMBA_P2P", "1") == "0":
            return
        dev = "cuda" if td.get_backend(group) == "nccl" else "cpu"
        try:
            mine, rc = backend.p2p_export(world), 0
        except Exception:                                     # noqa: BLE001 -- agreed on below
            mine, rc = bytes(64), -5
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8).to(dev)
        parts = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(world)]
        td.all_gather(parts, t, group=group)
        ok = torch.tensor([rc], dtype=torch.int32, device=dev)
        td.all_reduce(ok, op=td.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            handles = b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts)
            rc = backend.p2p_attach(handles, rank, world)
            ok = torch.tensor([rc], dtype=torch.int32, device=dev)
            td.all_reduce(ok, op=td.ReduceOp.MIN, group=group)
        if int(ok.item()) != 0:
            backend.p2p_detach()
            return
        self.active = True
        # (sfmba_p2p_attach also settled, over the link itself, what the ranks must agree on before the per-camera sums may
        # be exchanged inside the producing kernels: whether any rank's shard needs combine launches, and how many ranks
        # share one GPU -- a rehearsal of N > 1 on one device, where waiting workgroups must not fill the card.)

    def close(self):
        if self.active:
            self.backend.p2p_detach()
            self.active = False


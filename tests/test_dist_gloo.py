"""The N>1 path on CPU: world_size-2 gloo.  Covers the shard partition (cuts at point boundaries),
the Exchange all-reduce callback the C++ solver calls, and that per-shard normal-equation blocks summed
through it equal the unsharded ones (oracle as the checker)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from oracle import ba_oracle as orc
from sfmba import dist as sdist
from sfmba.synthetic import make_problem


def test_partition_cuts_at_point_boundaries():
    pb = make_problem(5, 200, 1500, seed=3)
    for world in (1, 2, 3, 8):
        shards = sdist.partition_points(pb.point_indices, pb.n_points, world)
        assert shards[0].obs_begin == 0 and shards[-1].obs_end == pb.n_obs
        assert shards[0].point_begin == 0 and shards[-1].point_end == pb.n_points
        for a, b in zip(shards[:-1], shards[1:]):
            assert a.obs_end == b.obs_begin and a.point_end == b.point_begin
        for s in shards:
            loc = sdist.shard_problem(pb, s)
            assert loc.n_obs == s.obs_end - s.obs_begin
            if loc.n_obs:
                assert loc.point_indices.min() >= 0 and loc.point_indices.max() < loc.n_points
                # every observation of every local point is local
                assert set(np.unique(pb.point_indices[s.obs_begin:s.obs_end])) <= set(range(s.point_begin, s.point_end))
        sizes = [s.obs_end - s.obs_begin for s in shards]
        assert max(sizes) - min(sizes) <= 2 * 64
        # round trip of the parameter vector
        xs = [sdist.shard_problem(pb, s).x0 for s in shards]
        assert np.array_equal(sdist.merge_solutions(xs, shards, pb.n_cameras, pb.n_points), pb.x0)
    with pytest.raises(ValueError):
        sdist.partition_points(pb.point_indices[::-1], pb.n_points, 2)


class _StubBackend:
    """Stands in for sfmba.Backend: records the registration and lets the test fire the callback the
    way libsfmba.so does (pointer into the arena, count, op)."""
    def __init__(self, n_cameras):
        self.n_cameras = n_cameras
        self.cb = None

    def exchange_doubles(self):
        return 45 * self.n_cameras + 32

    def set_exchange(self, ptr, n, cb, n_obs_total):
        self.base, self.n, self.cb, self.n_obs_total = ptr, n, cb, n_obs_total


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pb = make_problem(6, 120, 900, seed=5)
        shards = sdist.partition_points(pb.point_indices, pb.n_points, world)
        loc = sdist.shard_problem(pb, shards[rank])
        C = pb.n_cameras
        stub = _StubBackend(C)
        ex = sdist.Exchange(stub, n_obs_local=loc.n_obs)
        assert ex.n_obs_total == pb.n_obs and stub.n_obs_total == pb.n_obs
        # local normal-equation blocks with the oracle, packed as the library packs them:
        # arena = [acc0 6C | acc1 6C | acc2 6C | Ugc 27C | scalars 32]
        r, Jc, Jp = orc.jacobian_blocks(loc.x0, *loc.args)
        nb = orc.normal_blocks(r, Jc, Jp, C, loc.n_points, loc.camera_indices, loc.point_indices)
        iu = np.triu_indices(6)
        ugc = np.concatenate([nb.U[:, iu[0], iu[1]], nb.gc], axis=1)           # (C,27)
        ex.arena[18 * C:45 * C] = torch.from_numpy(ugc.ravel())
        ex.arena[45 * C + 0] = float(np.sum(r * r))                            # cost slot
        ex.arena[45 * C + 12] = float(np.abs(nb.gp).max())                     # max|g| slot
        stub.cb(stub.base + 8 * 18 * C, 27 * C, 0)                              # sum
        stub.cb(stub.base + 8 * 45 * C, 12, 0)
        stub.cb(stub.base + 8 * (45 * C + 12), 1, 1)                           # max
        with pytest.raises(ValueError):
            stub.cb(stub.base + 8 * (45 * C + 10), 100, 0)                     # outside the arena
        if rank == 0:
            rg, Jcg, Jpg = orc.jacobian_blocks(pb.x0, *pb.args)
            nbg = orc.normal_blocks(rg, Jcg, Jpg, C, pb.n_points, pb.camera_indices, pb.point_indices)
            ref = np.concatenate([nbg.U[:, iu[0], iu[1]], nbg.gc], axis=1).ravel()
            got = ex.arena[18 * C:45 * C].numpy()
            q.put(dict(ok_blocks=bool(np.allclose(got, ref, rtol=1e-12, atol=1e-9 * np.abs(ref).max())),
                       cost=float(ex.arena[45 * C]), cost_ref=float(np.sum(rg * rg)),
                       gmax=float(ex.arena[45 * C + 12]), gmax_ref=float(np.abs(nbg.gp).max()),
                       calls=ex.n_calls))
    finally:
        td.destroy_process_group()


def test_exchange_allreduce_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out["ok_blocks"]
    assert abs(out["cost"] - out["cost_ref"]) <= 1e-12 * out["cost_ref"]
    assert out["gmax"] == out["gmax_ref"]
    assert out["calls"] == 3


# ---- the sharded ALGORITHM: same collectives as the HIP path, oracle arithmetic, 2 gloo ranks -------------

class _TorchComm:
    """comm hooks of oracle.trf_schur on top of torch.distributed (what the C++ solver's exchange() is)."""
    def __init__(self):
        self.n_sum = self.n_max = 0

    def sum(self, a):
        t = torch.from_numpy(np.atleast_1d(np.asarray(a, dtype=np.float64)).copy())
        td.all_reduce(t, op=td.ReduceOp.SUM)
        self.n_sum += 1
        out = t.numpy()
        return float(out[0]) if np.ndim(a) == 0 else out.reshape(np.shape(a))

    def max(self, a):
        t = torch.tensor([float(a)], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        self.n_max += 1
        return float(t[0])


def _solve_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pb = make_problem(5, 90, 700, seed=8)
        shards = sdist.partition_points(pb.point_indices, pb.n_points, world)
        loc = sdist.shard_problem(pb, shards[rank])
        comm = _TorchComm()
        res = orc.trf_schur(loc.x0, *loc.args, ftol=1e-10, linear="pcg", pcg_tol=1e-3, comm=comm)
        xs = [None] * world
        td.all_gather_object(xs, res.x)
        if rank == 0:
            x = sdist.merge_solutions(xs, shards, pb.n_cameras, pb.n_points)
            ref = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, linear="pcg", pcg_tol=1e-3)
            q.put(dict(cams_equal=all(np.array_equal(xi[:30], xs[0][:30]) for xi in xs),
                       dx=float(np.abs(x - ref.x).max()), cost=res.cost, cost_ref=ref.cost,
                       nfev=(res.nfev, ref.nfev), status=(res.status, ref.status), n_sum=comm.n_sum))
    finally:
        td.destroy_process_group()


def test_sharded_solve_equals_unsharded_world2_gloo():
    """Two ranks, each with its own points/observations and all cameras, running the oracle's TRF with
    the HIP path's collectives (camera blocks, reduced rhs, Schur products, scalars) reproduce the
    single-process solve: replicated cameras stay bitwise identical across ranks, the merged solution
    agrees to summation-order rounding."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_solve_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert out["cams_equal"]
    assert out["status"][0] == out["status"][1] and out["nfev"][0] == out["nfev"][1]
    assert abs(out["cost"] - out["cost_ref"]) <= 1e-10 * out["cost_ref"]
    assert out["dx"] < 1e-7
    assert out["n_sum"] > 50

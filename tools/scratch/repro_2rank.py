"""Scratch: the 300-camera two-rank case of test_direct_allreduce_over_peer_mapped_memory, repeated, with the parent
process holding a GPU context as pytest does; every rank reports how its solves ended.
    python tools/scratch/repro_2rank.py [repeats]"""
import os, sys, socket, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np

DIMS = (300, 2000, 16000, 9, 0.01)


def worker(rank, world, port, q, debug):
    import torch, torch.distributed as td, sfmba
    from sfmba import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    pb = sfmba.make_problem(DIMS[0], DIMS[1], DIMS[2], seed=DIMS[3], x0_noise=DIMS[4])
    shards = sdist.partition_points(pb.point_indices, pb.n_points, world)
    loc = sdist.shard_problem(pb, shards[rank])
    be = sfmba.Backend(0)
    stream = torch.cuda.Stream()
    msgs = []
    with torch.cuda.stream(stream):
        be.set_stream(stream.cuda_stream)
        for n, v in debug: be.debug_option(n, v)
        be.set_problem(*loc.args)
        ex = sdist.Exchange(be, n_obs_local=loc.n_obs, device="cuda")
        link = sdist.DirectLink(be)
        opt = be.default_options(); opt.ftol = 1e-10
        for k in range(2):
            t0 = time.time()
            try:
                x, res, fun, grad = be.solve(loc.x0, opt)
                msgs.append(f"ok nfev {res.nfev} cost {res.cost:.10e} {time.time() - t0:.3f}s")
            except Exception as e:
                msgs.append(f"FAILED after {time.time() - t0:.3f}s: {e}")
                break
            torch.cuda.synchronize()
    td.barrier()
    q.put((rank, msgs))
    link.close(); be.close(); td.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import sfmba
    pb = sfmba.make_problem(11, 3000, 10000, seed=0)
    sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf", args=pb.args)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    debug = tuple((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[2:])
    for it in range(n):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        ctx = mp.get_context("spawn"); q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, 2, port, q, debug)) for r in range(2)]
        for p in procs: p.start()
        got = sorted(q.get(timeout=200) for _ in range(2))
        print(it, got, flush=True)
        for p in procs: p.join(timeout=100)
        if any("FAILED" in m for _, ms in got for m in ms):
            break

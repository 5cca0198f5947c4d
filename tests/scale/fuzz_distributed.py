"""Randomised sweep (CPU, gloo, NumPy stand-in kernels; not a pytest) of the multi-rank host logic of engine.AlsEngine: world sizes
2 .. 8, shapes that are not multiples of anything, 1 .. 7 chunks, biases, the three exchange modes, the need-list gather forced on /
off / automatic, replicated and distributed cost-balanced builds, power-law matrices -- two ALS iterations and the eval sums against
the single-process oracle.  Usage: python tests/scale/fuzz_distributed.py [cases] [seed]"""
import os, sys, socket, tempfile, time
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, cfg, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from oracle import wmf_oracle as orc
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine
        n_users, n_items, dim, bias = cfg["n_users"], cfg["n_items"], cfg["dim"], cfg["bias"]
        indptr, indices, counts = synth.make_counts(n_users, n_items, cfg["dbar"], seed=cfg["seed"], zipf_a=cfg["zipf"])
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        kw = {"gather": {"reduce_mode": False, "pipe_mode": False}, "reduce": {"reduce_mode": True}, "pipe": {"reduce_mode": False, "pipe_mode": True},
              "auto": {}}[cfg["mode"]]
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cpu", kernels=NumpyKernels(), chunks=cfg["chunks"], sparse_mode=cfg["sparse"], **kw)
        per = n_users // world
        lo = rank * per
        hi = n_users if rank == world - 1 else lo + per
        e0, e1 = int(indptr[lo]), int(indptr[hi])
        if cfg["distributed"]:
            eng.set_interactions_distributed(lo, indptr[lo: hi + 1] - e0, indices[e0:e1], values[e0:e1], balance=cfg["balance"])
        else:
            eng.set_interactions(indptr, indices, values, balance=cfg["balance"])
        for side in ("users", "items"):
            t = torch.tensor([eng.csr[side].nnz], dtype=torch.int64)
            dist.all_reduce(t)
            assert int(t) == indices.numel()
        eng.set_factors("items", orc.init_items(n_items, dim, bias))
        for _ in range(cfg["iters"]):
            eng.half_step("users")
            eng.half_step("items")
        eng.check_numerics()
        if cfg["distributed"]:
            shard = eng.make_eval_shard_distributed(lo, indptr[lo: hi + 1] - e0, indices[e0:e1], counts[e0:e1])
        else:
            shard = eng.make_eval_shard(indptr, indices, counts)
        sq, ab, cnt = eng.eval_sums(shard)
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items, sums=np.array([sq, ab, cnt]),
                     modes=np.array([str(eng.reduce), str(eng.pipe), str(eng.sparse)]))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.perf_counter()
    for case in range(cases):
        world = int(rng.choice([2, 3, 4, 5, 8]))
        dim = int(rng.choice([3, 6, 16]))
        lo_rows = 2 * (dim + 1) + 10       # (fewer rows than features: an ill-conditioned Gramian, where float32 stand-ins and the oracle part ways)
        cfg = dict(n_users=int(rng.integers(max(world + 5, lo_rows), 500)), n_items=int(rng.integers(max(world + 3, lo_rows), 200)), dim=dim,
                   bias=bool(rng.integers(2)), dbar=int(rng.integers(1, 10)), seed=int(rng.integers(1 << 20)), zipf=float(rng.choice([0.0, 0.0, 1.1])),
                   mode=str(rng.choice(["gather", "reduce", "pipe", "auto"])), chunks=int(rng.choice([1, 2, 3, 4, 7])),
                   sparse=[None, True, False][int(rng.integers(3))], distributed=bool(rng.integers(2)), balance=bool(rng.integers(2)),
                   iters=int(rng.integers(1, 3)))
        if cfg["mode"] == "reduce": cfg["sparse"] = None
        print(f"case {case:3d}: W={world} {cfg}", end=" ", flush=True)
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "out.npz")
            mp.spawn(worker, args=(world, _free_port(), cfg, out), nprocs=world, join=True)
            got = dict(np.load(out))
        indptr, indices, counts = synth.make_counts(cfg["n_users"], cfg["n_items"], cfg["dbar"], seed=cfg["seed"], zipf_a=cfg["zipf"])
        raw = synth.to_scipy(indptr, indices, counts, (cfg["n_users"], cfg["n_items"]))
        C = raw.astype(np.float64); C.data = 10 * np.log(1 + C.data); CT = C.T.tocsr()
        items = orc.init_items(cfg["n_items"], cfg["dim"], cfg["bias"])
        step = orc.recompute_factors_bias if cfg["bias"] else orc.recompute_factors
        for _ in range(cfg["iters"]):
            users = step(items, C, 0.1); items = step(users, CT, 0.1)
        np.testing.assert_allclose(got["users"], users, rtol=5e-4, atol=5e-5)
        np.testing.assert_allclose(got["items"], items, rtol=5e-4, atol=5e-5)
        mse = orc.eval_prec(users, items, raw, cfg["bias"])
        assert abs(got["sums"][0] / got["sums"][2] - mse) <= 2e-4 * mse and got["sums"][2] == raw.nnz
        print("ok", " ".join(got["modes"].tolist()))
    print(f"{cases} cases in {time.perf_counter() - t0:.1f} s")

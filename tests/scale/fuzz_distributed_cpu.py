"""Randomised sweep of the multi-rank host logic on CPU ranks (gloo, stand-in kernels): world size, shapes, chunk counts
and exchange mode are drawn at random; the sharded result must equal the single-process oracle.  Not a pytest.
Usage: python tests/scale/fuzz_distributed_cpu.py [cases] [seed]"""
import os, socket, sys, tempfile
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, cfg, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from oracle import wmf_oracle as orc
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine
        n_users, n_items, dim, bias, chunks, mode, seed, deg = cfg
        indptr, indices, counts = synth.make_counts(n_users, n_items, deg, seed=seed)
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cpu", kernels=NumpyKernels(), chunks=chunks,
                        reduce_mode=(mode == "reduce"), pipe_mode=(mode == "pipe"))
        eng.set_interactions(indptr, indices, values)
        eng.set_factors("items", orc.init_items(n_items, dim, bias))
        for _ in range(2):
            eng.half_step("users"); eng.half_step("items")
        eng.check_numerics()
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items)
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); return s.getsockname()[1]


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(cases):
        world = int(rng.choice([2, 3, 4, 5]))
        cfg = (int(rng.integers(world, 160)), int(rng.integers(world, 90)), int(rng.integers(1, 9)), bool(rng.integers(2)),
               int(rng.integers(1, 7)), str(rng.choice(["gather", "reduce", "pipe"])), int(rng.integers(1 << 20)), int(rng.integers(1, 9)))
        out = os.path.join(tempfile.mkdtemp(), "o.npz")
        mp.spawn(worker, args=(world, free_port(), cfg, out), nprocs=world, join=True)
        got = np.load(out)
        n_users, n_items, dim, bias, chunks, mode, seed, deg = cfg
        indptr, indices, counts = synth.make_counts(n_users, n_items, deg, seed=seed)
        C = synth.to_scipy(indptr, indices, counts, (n_users, n_items)).astype(np.float64)
        C.data = 10 * np.log(1 + C.data)
        CT = C.T.tocsr()
        items = orc.init_items(n_items, dim, bias)
        step = orc.recompute_factors_bias if bias else orc.recompute_factors
        for _ in range(2):
            users = step(items, C, 0.1); items = step(users, CT, 0.1)
        eu = np.abs(got["users"] - users).max() / max(np.abs(users).max(), 1e-30)
        ei = np.abs(got["items"] - items).max() / max(np.abs(items).max(), 1e-30)
        ok = eu < 2e-4 and ei < 2e-4
        bad += not ok
        print(f"case {case:2d}: world={world} users={n_users} items={n_items} dim={dim} bias={int(bias)} chunks={chunks} mode={mode:6s} "
              f"max rel err users {eu:.1e} items {ei:.1e} {'' if ok else '<-- MISMATCH'}", flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)

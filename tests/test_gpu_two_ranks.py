"""Two ranks sharing the one GPU of the test box (gloo transport, device tensors): the real HIP
kernels under the sharded host path -- round-robin ids, padding rows, Gramian all-reduce, chunked solve
with asynchronous all-gathers, whitening of the gathered matrix -- against the single-process oracle.  (RCCL itself needs one GPU per rank and
is exercised by the driver's multi-GPU bench; NCCL refuses two ranks on one device.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SMALL = (1203, 257, 24, 9)          # users, items, k, mean user degree


def _worker(rank, world, port, bias, out_path, reduce_mode=None, pipe_mode=None, backend="gloo", shape=SMALL, sparse_mode=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import wmf_oracle as orc
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine
        n_users, n_items, dim, dbar = shape
        indptr, indices, counts = synth.make_counts(n_users, n_items, dbar, seed=11)
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cuda:0", chunks=3, reduce_mode=reduce_mode, pipe_mode=pipe_mode,
                        force_exchange=True, sparse_mode=sparse_mode)
        assert eng.exchange and len(eng.chunk_bounds["users"]) == 3
        partial_ok = eng.pr > 0                       # partial systems exist for f <= 144 only: wider models always gather
        assert eng.reduce["items"] == (bool(reduce_mode) and partial_ok)
        eng_pipe_expected = bool(pipe_mode) and partial_ok
        eng.set_interactions(indptr, indices, values)
        assert eng.pipe["items"] == eng_pipe_expected and eng.pipe["users"] == eng_pipe_expected
        assert eng.sparse["users"] == bool(sparse_mode) and not eng.sparse["items"]
        eng.set_factors("items", orc.init_items(n_items, dim, bias))
        for _ in range(2):
            eng.half_step("users")
            eng.half_step("items")
        eng.check_numerics()
        shard = eng.make_eval_shard(indptr, indices, counts)
        sq, ab, cnt = eng.eval_sums(shard)
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items, sums=np.array([sq, ab, cnt]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bias,reduce_mode,pipe_mode", [(False, None, None), (True, None, None), (False, True, None),
                                                        (True, True, None), (False, False, True), (True, False, True)])
def test_two_ranks_one_gpu_match_oracle(tmp_path, bias, reduce_mode, pipe_mode):
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    try:
        mp.spawn(_worker, args=(2, _free_port(), bias, out, reduce_mode, pipe_mode), nprocs=2, join=True)
    except Exception as exc:                      # gloo without device-tensor collectives on this build
        if "gloo" in str(exc).lower() and "cuda" in str(exc).lower():
            pytest.skip(f"gloo cannot move device tensors here: {exc}")
        raise
    _check_against_oracle(np.load(out), bias)


def _check_against_oracle(got, bias, shape=SMALL):
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    n_users, n_items, dim, dbar = shape
    indptr, indices, counts = synth.make_counts(n_users, n_items, dbar, seed=11)
    raw = synth.to_scipy(indptr, indices, counts, (n_users, n_items))
    C = raw.astype(np.float64)
    C.data = 10 * np.log(1 + C.data)
    CT = C.T.tocsr()
    items = orc.init_items(n_items, dim, bias)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    for _ in range(2):
        users = step(items, C, 0.1)
        items = step(users, CT, 0.1)
    assert np.linalg.norm(got["users"] - users) <= 1e-3 * np.linalg.norm(users)
    assert np.linalg.norm(got["items"] - items) <= 1e-3 * np.linalg.norm(items)
    mse = orc.eval_prec(users, items, raw, bias)
    assert abs(got["sums"][0] / got["sums"][2] - mse) <= 1e-4 * mse


@pytest.mark.parametrize("bias,reduce_mode,pipe_mode", [(False, None, None), (True, None, None), (False, True, None),
                                                        (True, True, None), (False, False, True), (True, False, True)])
def test_one_rank_over_rccl_matches_oracle(tmp_path, bias, reduce_mode, pipe_mode):
    """The RCCL calls of the multi-GPU path themselves (backend nccl: all_reduce of the fp64 Gramian, asynchronous
    all_gather_into_tensor / reduce_scatter_tensor on RCCL's stream, work.wait() ordering against the kernels) with a
    world of ONE rank forced through the exchange machinery (force_exchange): 3 chunks per side, all three modes."""
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(1, _free_port(), bias, out, reduce_mode, pipe_mode, "nccl"), nprocs=1, join=True)
    _check_against_oracle(np.load(out), bias)


@pytest.mark.parametrize("backend,world,k,bias,pipe_mode", [("gloo", 2, 24, False, False), ("gloo", 2, 128, True, False), ("gloo", 2, 128, True, True),
                                                            ("gloo", 2, 256, False, False), ("nccl", 1, 24, True, False), ("nccl", 1, 64, False, True)])
def test_need_list_gather_on_the_device(tmp_path, backend, world, k, bias, pipe_mode):
    """The need-list gather of the users (engine.AlsEngine._setup_sparse) under the real kernels: the item-major CSR re-indexed
    into the compact gathered matrix, the whitening pass over compact chunk ranges, the row kernels gathering from it -- two
    ranks on one GPU (gloo transport) at k = 24, 128 + biases (split layout, also pipelined) and 256, and one rank over real
    RCCL (all_to_all_single with split sizes, asynchronous, ordered by work.wait())."""
    shape = (1203, 257, k, 9 if k < 100 else 40)
    out = str(tmp_path / "out.npz")
    try:
        mp.spawn(_worker, args=(world, _free_port(), bias, out, False, pipe_mode, backend, shape, True), nprocs=world, join=True)
    except Exception as exc:
        if "gloo" in str(exc).lower() and "cuda" in str(exc).lower():
            pytest.skip(f"gloo cannot move device tensors here: {exc}")
        raise
    _check_against_oracle(np.load(out), bias, shape)


@pytest.mark.parametrize("k,bias,mode", [(128, True, "gather"), (128, True, "reduce"), (128, True, "pipe"), (128, False, "pipe"),
                                         (256, False, "gather"), (256, False, "reduce"), (256, True, "pipe")])
def test_two_ranks_one_gpu_at_bench_widths(tmp_path, k, bias, mode):
    """The widths the scaling configurations run at -- k = 128 +- biases (cfg3 / cfg4: LDS-DMA heavy-row kernel, border
    column, partial systems of 2336+ floats per row) and k = 256 (cfg5: four waves per row; no partial systems, so a
    requested reduce / pipe mode must fall back to the all-gather) -- under the sharded host path with three chunks per
    side: chunked CSR views, slot-strided partial systems, the exchange of every mode, against the single-process oracle.
    Items have ~190 entries each (heavy rows), users ~40 (heavy and 17..32 rows)."""
    shape = (1203, 257, k, 40)
    out = str(tmp_path / "out.npz")
    kw = {"gather": (False, False), "reduce": (True, None), "pipe": (False, True)}[mode]
    try:
        mp.spawn(_worker, args=(2, _free_port(), bias, out, kw[0], kw[1], "gloo", shape), nprocs=2, join=True)
    except Exception as exc:
        if "gloo" in str(exc).lower() and "cuda" in str(exc).lower():
            pytest.skip(f"gloo cannot move device tensors here: {exc}")
        raise
    _check_against_oracle(np.load(out), bias, shape)


def _tiny_worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import wmf_oracle as orc
        from recmodel_amd.engine import AlsEngine
        import scipy.sparse as sp
        C = _tiny_matrix()
        eng = AlsEngine(C.shape[0], C.shape[1], 5, False, 0.1, device="cuda:0", chunks=2,
                        reduce_mode=(mode == "reduce"), pipe_mode=(mode == "pipe"))
        eng.set_interactions(torch.from_numpy(C.indptr.astype(np.int64)), torch.from_numpy(C.indices.astype(np.int64)),
                             torch.from_numpy(C.data.astype(np.float32)))
        eng.set_factors("items", orc.init_items(C.shape[1], 5))
        eng.half_step("users")
        eng.half_step("items")
        eng.check_numerics()
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items)
    finally:
        dist.destroy_process_group()


def _tiny_matrix():
    """3 users x 40 items; user 1 (the only user of rank 1) bought nothing: rank 1 holds no entries at all, and
    half of the items have no entries either."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    rows = np.repeat([0, 2], 12)
    cols = np.concatenate([rng.choice(20, 12, replace=False), rng.choice(20, 12, replace=False)])
    return sp.csr_matrix((rng.integers(1, 6, 24).astype(np.float64), (rows, cols)), shape=(3, 40))


@pytest.mark.parametrize("mode", ["gather", "reduce", "pipe"])
def test_rank_without_entries(tmp_path, mode):
    from oracle import wmf_oracle as orc
    out = str(tmp_path / "tiny.npz")
    mp.spawn(_tiny_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    got = np.load(out)
    C = _tiny_matrix()
    items0 = orc.init_items(40, 5)
    users = orc.recompute_factors(items0, C, 0.1)
    items = orc.recompute_factors(users, C.T.tocsr(), 0.1)
    assert np.linalg.norm(got["users"] - users) <= 1e-4 * np.linalg.norm(users)
    assert np.linalg.norm(got["items"] - items) <= 1e-4 * np.linalg.norm(items)
    assert not got["users"][1].any() and not got["items"][20:].any()

#!/usr/bin/env python3
"""WMF/ALS throughput on MI355X: one "step" = one ALS iteration (users half step + items half
step, each with its Gramian, factorisation, whitening, row solve and -- on more than one GPU --
the Gramian all-reduce and the all-gather of the whitened block) over a synthetic confidence
matrix that is already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|tiny] [--zipf A]

For N > 1 the driver launches one rank per GPU through torch.distributed.run; ranks talk RCCL.
Scaling is weak: each GPU brings its own ``n_users`` users (the item catalogue is shared), so the
stored entries per GPU stay fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from recmodel_amd import _lib, synth  # noqa: E402
from recmodel_amd.engine import AlsEngine  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
F32_MFMA_PEAK_TF = 157.3  # dense f32-input MFMA peak
SOLVE_BINS = {4: 0, 5: 1, 11: 2, 6: 3}   # profile slot -> plan bin whose rows that kernel processes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--zipf", type=float, default=0.0, help="item popularity exponent (0 = uniform)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="solve each side in this many chunks (0 = engine default: 1 on one GPU, 4 on several, where "
                         "the all-gather of a chunk overlaps the solve of the next)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-users", type=int, default=0, help="user rows in the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def cpu_share():
    """CPUs this process can actually use: the cgroup quota when there is one (a GPU box hands each job a share
    of a large host), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg_name, user_rows, item_rows, users_f, k, bias, gamma, n_items):
    """The oracle's NumPy restatement of recompute_factors[_bias] (same per-row gather / dot / solve
    structure as RecModel/wmf_model.py:220-239, one thread) on row samples of the same matrices:
    ``user_rows`` = the first rows of the user-major count matrix, ``item_rows`` = the first rows of
    the item-major confidence matrix (full degree), in the job's users:items row ratio."""
    from oracle import wmf_oracle as orc
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        C = user_rows.astype(np.float64)
        C.data = orc.confidence_transform(C.data)
        items = orc.init_items(n_items, k, bias)
        step = orc.recompute_factors_bias if bias else orc.recompute_factors
        t0 = time.perf_counter()
        step(items, C, gamma)
        t_u = time.perf_counter() - t0
        CT = item_rows.astype(np.float64)
        t0 = time.perf_counter()
        step(users_f, CT, gamma)
        t_i = time.perf_counter() - t0
    rows = C.shape[0] + CT.shape[0]
    out = {"value": rows / (t_u + t_i), "unit": "row-updates/s", "cores": 1, "kind": "port",
           "sample": f"{cfg_name}: first {C.shape[0]} user rows ({C.nnz} nnz, {t_u:.1f}s) + first {CT.shape[0]} item rows "
                     f"({CT.nnz} nnz, {t_i:.1f}s) of the same matrix, NumPy per-row loop, 1 BLAS thread",
           "host_cpus": os.cpu_count()}
    # the "fair" CPU figure of SURVEY.md 8(d): the oracle's C restatement (float64, own LU) with OpenMP on every
    # host core, same sample
    try:
        from oracle import c_oracle
        threads = c_oracle.set_threads(cpu_share())
        t0 = time.perf_counter()
        c_oracle.half_step(items, C, gamma, bias)
        c_oracle.half_step(users_f, CT, gamma, bias)
        t_c = time.perf_counter() - t0
        out["all_cores"] = {"value": rows / t_c, "unit": "row-updates/s", "cores": threads, "kind": "port",
                            "sample": f"same rows through oracle/wmf_oracle.c (OpenMP, {threads} threads, float64), {t_c:.2f}s"}
    except Exception as exc:                         # library not built on this box: say so, never substitute
        out["all_cores"] = {"error": str(exc)}
    return out


def host_boundary(user_rows, k, bias, gamma, n_items):
    """Rows/s through the HOST entry point (wmf_recompute_factors_host: host CSR + host factors in, host factors out,
    one half step, including the PCIe transfers and the plan) -- what a caller pays who keeps nothing on the device.
    Never the headline value."""
    from recmodel_amd import WMF
    C = user_rows.astype(np.float32)
    C.data = (10.0 * np.log(1.0 + C.data.astype(np.float64))).astype(np.float32)      # alpha = 10, beta = 1, 'log'
    m = WMF(num_items=n_items, num_users=C.shape[0], dim=k, gamma=gamma, weighted=True, bias=bias)
    step = m.recompute_factors_bias if bias else m.recompute_factors
    step(m.items, C, gamma)                                         # warm-up (library load, first launches)
    t0 = time.perf_counter()
    step(m.items, C, gamma)
    dt = time.perf_counter() - t0
    mb = (C.nnz * 8 + C.shape[0] * 8 + (n_items + C.shape[0]) * m.items.shape[1] * 4) / 1e6
    return {"value": C.shape[0] / dt, "unit": "row-updates/s", "rows": int(C.shape[0]), "seconds": dt,
            "pcie_megabytes": mb, "note": "one user half step through wmf_recompute_factors_host, PCIe-inclusive"}


def measured_traffic(config, slot, f, ld):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile of this
    workload (profiles/rNN_<config>_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command, read side doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if absent."""
    import glob
    nfb, nch = (f + 15) // 16, (ld + 15) // 16
    nfb_heavy = f // 16 if (f > 16 and f % 16 == 1 and (f // 16) % 4 != 3) else nfb     # border variant of the heavy-row kernel
    heavy = f"solve_directw_kernel<{nfb_heavy}, 0"
    if (f, ld) in ((128, 128), (129, 132)):                          # k = 128: the LDS-DMA ring kernel (wmf_directl.hip)
        heavy = "solve_directl_kernel<8, " + ("true" if f == 129 else "false")
    key = {0: f"gram_kernel<{nfb}, 1>", 3: f"transform_kernel<{nfb}, true", 4: f"solve_low_kernel<{nch}, 1,",
           5: f"solve_low_kernel<{nch}, 2", 11: heavy}.get(slot)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_traffic.json")))
    if not key or not files:
        return None
    table = json.load(open(files[-1])).get("kernels", {})
    k = next((v for name, v in table.items() if name.startswith(key)), None)       # template arguments may follow
    if not k or k.get("FETCH_SIZE_KB_mean") is None:
        return None
    return (2.0 * k["FETCH_SIZE_KB_mean"] + k.get("WRITE_SIZE_KB_mean", 0.0)) * 1024.0


def main():
    args = parse()
    # stdout carries exactly one line, the JSON result: RCCL prints a version banner to stdout when a communicator is
    # created, so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL.  WMF_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed on a box with fewer
    # GPUs than ranks (ranks then share devices; timing is meaningless there, correctness of the path is not)
    backend = os.environ.get("WMF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    force_exchange = world == 1 and os.environ.get("WMF_FORCE_EXCHANGE") == "1"
    if force_exchange:      # rehearsal: one rank through the chunked exchange path, collectives over real RCCL (engine.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{local_rank}"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            torch.distributed.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")
    lib = _lib.load()

    n_users_1, n_items, dbar, k, bias = synth.CONFIGS[args.config]
    n_users = n_users_1 * world                     # weak scaling over users
    gamma = 0.1
    seed = 1993 + {"cfg1": 1, "cfg2": 2, "cfg3": 3, "cfg4": 3, "cfg5s": 5, "tiny": 9}[args.config]

    # ---- synthetic workload, generated on the GPU (identical on every rank) -----------------
    t0 = time.perf_counter()
    parts = [synth.make_counts(n_users_1, n_items, dbar, seed, device=dev, zipf_a=args.zipf, first_user=b * n_users_1)
             for b in range(world)]
    ptrs, offset = [parts[0][0][:1]], 0
    for ip_b, _, _ in parts:
        ptrs.append(ip_b[1:] + offset)
        offset += int(ip_b[-1])
    indptr = torch.cat(ptrs)
    indices = torch.cat([p[1] for p in parts])
    counts = torch.cat([p[2] for p in parts])
    del parts, ptrs
    nnz = int(indices.numel())
    eng = AlsEngine(n_users, n_items, k, bias, gamma, device=dev, chunks=args.chunks or None)
    values = counts.clone()
    eng.K.confidence_transform(values, 10.0, 1.0, 0)
    eng.set_interactions(indptr, indices, values)
    from recmodel_amd import WMF                    # the package's own constructor draws the initial item factors
    eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=gamma, weighted=True, bias=bias).items)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        eng.half_step("users")
        eng.half_step("items")

    for _ in range(args.warmup):
        step()
    barrier()
    eng.check_numerics()
    lib.wmf_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.wmf_profile_enable(0)
    eng.check_numerics()
    ms = np.zeros(_lib.WMF_PROF_SLOTS)
    launches = np.zeros(_lib.WMF_PROF_SLOTS, dtype=np.int64)
    lib.wmf_profile_read(ms.ctypes.data_as(ctypes.c_void_p), launches.ctypes.data_as(ctypes.c_void_p))
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- eval pass (not part of the timed step; reported separately) ------------------------
    shard = eng.make_eval_shard(indptr, indices, counts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sq, ab, cnt = eng.eval_sums(shard)
    torch.cuda.synchronize()
    t_eval = time.perf_counter() - t0

    # ---- roofline of the dominant kernel (largest total time inside the timed region) ---------
    f = eng.f
    kernels = {}
    for slot in range(_lib.WMF_PROF_SLOTS):
        if launches[slot]:
            kernels[lib.wmf_profile_slot_name(slot).decode()] = {
                "launches": int(launches[slot]), "avg_ms": ms[slot] / launches[slot], "total_ms": float(ms[slot])}
    n_loc = eng.n_local
    work = {}          # slot -> (bound, algorithmic units per STEP on this rank)
    for slot, b in SOLVE_BINS.items():
        tot = 0
        for side in ("users", "items"):
            c = eng.csr[side]
            rows_b, nnz_b = int(c.bin_rows[b]), int(c.bin_nnz[b])
            tot += nnz_b * (4 * f + 8) + rows_b * (4 * f + 4)      # gathered + written bytes
        work[slot] = ("hbm", float(tot))
    rows_all = n_loc["users"] + n_loc["items"]
    work[0] = ("mfma", 2.0 * f * f * rows_all)                     # Gramian: 2 m f^2 flops per side
    work[3] = ("mfma", 2.0 * f * f * 2 * rows_all)                 # whiten + unwhiten GEMMs
    cand = [s_ for s_ in work if launches[s_] and work[s_][1] > 0]
    roofline = None
    if cand:
        dom = max(cand, key=lambda s_: ms[s_])
        bound, units_per_step = work[dom]
        per_launch = units_per_step * args.steps / launches[dom]
        avg_s = ms[dom] / launches[dom] / 1e3
        peak, unit, scale = (HBM_PEAK_GBS, "GB/s", 1e9) if bound == "hbm" else (F32_MFMA_PEAK_TF, "TFLOP/s", 1e12)
        traffic = measured_traffic(args.config, dom, f, eng.ld)
        roofline = {"kernel": lib.wmf_profile_slot_name(dom).decode(), "bound": bound,
                    "achieved": per_launch / avg_s / scale, "peak": peak, "unit": unit,
                    "frac": per_launch / avg_s / scale / peak, "traffic": traffic,
                    "algorithmic_units_per_launch": per_launch, "avg_launch_ms": avg_s * 1e3,
                    "share_of_step": float(ms[dom] / (elapsed * 1e3))}
    epoch_bytes = sum(eng.algorithmic_bytes_half(s) for s in ("users", "items"))
    if world > 1:
        eb = torch.tensor([epoch_bytes], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(eb)
        epoch_bytes = float(eb.item())

    out = {
        "metric": "ALS user+item row-updates/sec", "value": (n_users + n_items) * args.steps / elapsed,
        "unit": "row-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "precision_note": "float32 storage, float32 accumulation everywhere; Gramian fp64 across waves; where a kernel is "
                          "MFMA-bound its products are three-way bf16 splits of the float32 operands (six bf16 MFMAs per tile, "
                          "every product exact, result within 1.4x of the f32-MFMA error: DESIGN.md section 5); parity tolerance "
                          "unchanged (tests/test_gpu_parity.py)",
        "config": {"workload": f"{args.config}: WMF k={k}{'+bias' if bias else ''}, {n_users}x{n_items} CSR, "
                               f"{nnz} nnz, alpha-log confidence, gamma={gamma}, zipf_a={args.zipf}",
                   "n_users": n_users, "n_items": n_items, "nnz": nnz, "k": k, "bias": bias,
                   "sharding": f"users+items round-robin over {world} GPU(s); exchange users: "
                               f"{'reduce-scatter of partial systems' if eng.reduce['users'] else 'all-gather'} in "
                               f"{len(eng.chunk_bounds['users'])} chunk(s), items: "
                               f"{'reduce-scatter of partial systems' if eng.reduce['items'] else 'all-gather'} in "
                               f"{len(eng.chunk_bounds['items'])} chunk(s); accumulation pipelined over arriving chunks: "
                               f"{[s_ for s_ in ('users', 'items') if eng.pipe[s_]] or 'no'}"},
        "nnz_per_s": 2.0 * nnz * args.steps / elapsed,
        "epoch_algorithmic_GBps": epoch_bytes * args.steps / elapsed / 1e9,
        "epoch_hbm_frac": epoch_bytes * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world),
        "roofline": roofline, "kernels": kernels,
        "eval": {"ms": t_eval * 1e3, "mse": sq / cnt if cnt else None},
        "setup_s": t_setup,
        "row_bins": {s: {"rows": eng.csr[s].bin_rows.tolist(), "nnz": eng.csr[s].bin_nnz.tolist()} for s in ("users", "items")},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        su = args.cpu_users or {64: 300_000, 128: 60_000, 16: 943, 256: 12_000}.get(k, 20_000)
        su = min(su, n_users_1)
        si = max(1, min(n_items, su * n_items // n_users_1))
        ip = indptr[: su + 1].cpu().numpy()
        user_rows = synth.to_scipy(indptr[: su + 1], indices[: ip[-1]], counts[: ip[-1]], (su, n_items))
        ci = eng.csr["items"]                                    # item-major confidence matrix (world == 1: positions = ids)
        ipi = ci.indptr[: si + 1].cpu().numpy()
        item_rows = synth.to_scipy(ci.indptr[: si + 1], ci.indices[: ipi[-1]].to(torch.int64), ci.values[: ipi[-1]],
                                   (si, n_users))
        if n_users <= 2_000_000:
            users_f = eng.get_factors("users")
        else:                                                    # timing does not depend on the values
            users_f = np.random.default_rng(0).random((n_users, eng.f), dtype=np.float32)
        out["cpu_baseline"] = cpu_baseline(args.config, user_rows, item_rows, users_f, k, bias, gamma, n_items)
        out["host_boundary"] = host_boundary(user_rows, k, bias, gamma, n_items)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_exchange:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

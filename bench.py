#!/usr/bin/env python3
"""WMF/ALS throughput on MI355X.  One "step" = one ALS iteration (users half step + items half step, each with its
Gramian, factorisation, whitening, row solve and -- on more than one GPU -- the Gramian all-reduce and the exchange of
the freshly solved block) over a synthetic confidence matrix that is already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3] [--also cfg2|none] [--scaling strong|weak] [--zipf A]

Default workload = BASELINE.json configs[2] (the configuration the north-star target is quoted on): k = 128 with
user / item biases, 10 M x 1 M, ~100 M stored entries.  configs[1] (cfg2: k = 64, 1 M x 100 K, 20 M entries) is timed
in the same process afterwards and reported under "also".  For N > 1 the driver launches one rank per GPU through
torch.distributed.run (RCCL); the default there is STRONG scaling -- BASELINE.json configs[3]: the same 10 M x 1 M
matrix at k = 128, users and items dealt over the N GPUs -- with the bias-free variant (cfg4) under "also".
Rank 0 prints ONE JSON line.

Roofline bookkeeping.  Every kernel launch is timed with HIP events on its own stream
(wmf_profile_* in the C ABI), keyed by the symbol rocprofv3 prints and by the half step it ran in.  Algorithmic bytes
of a row kernel = sum over ITS rows of (4f + 8) x entries + (4f + 4): what the reference's Y[idx] gather, the CSR
entry and the written row amount to, whatever the kernel really moves.  "traffic" is NOT measured in this run: it is
the HBM byte count of the same kernel in the newest committed rocprofv3 --pmc profile of this workload
(profiles/rNN_<config>_traffic.json, named in "traffic_source").
"""
import argparse
import glob
import json
import re
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from recmodel_amd import _lib, synth  # noqa: E402
from recmodel_amd.engine import AlsEngine  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
F32_MFMA_PEAK_TF = 157.3    # dense f32-input MFMA peak: what an exact-f32 Gramian is priced against
SIDES = ("users", "items")
SEEDS = {"cfg1": 1994, "cfg2": 1995, "cfg3": 1996, "cfg4": 1996, "cfg3m": 1996, "cfg5s": 1998, "tiny": 2002}   # 1993 + config index


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", choices=sorted(synth.CONFIGS))
    ap.add_argument("--also", default="auto", help="second workload reported under 'also' (auto: cfg2 on one GPU, the "
                    "bias-free cfg4 on several; none: skip)")
    ap.add_argument("--scaling", default="auto", choices=("auto", "strong", "weak"),
                    help="N > 1: strong = the same matrix dealt over the ranks (default), weak = N times the users")
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
           "nnz_per_s": (C.nnz + CT.nnz) / (t_u + t_i), "host_cpus": os.cpu_count()}
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


# --------------------------------------------------------------------------------------- roofline bookkeeping
def kernel_work(name, csr, eng, side):
    """(bound, algorithmic units per launch) of kernel ``name`` launched in the half step that updates ``side``;
    None for kernels without a byte / flop model here (factorisation, reductions, fallbacks)."""
    f = eng.f
    row_b, ent_b = 4 * f + 4, 4 * f + 8
    rows, nnz = csr.bin_rows, csr.bin_nnz
    fixed = "items" if side == "users" else "users"
    if name.startswith("solve_pair_kernel"):
        return "hbm", csr.nnz8 * ent_b + csr.rows8 * row_b
    if name.startswith("solve_low_kernel") and ", 1," in name:
        return "hbm", (nnz[0] - csr.nnz8) * ent_b + (rows[0] - csr.rows8) * row_b
    if name.startswith("solve_low_kernel"):
        return "hbm", nnz[1] * ent_b + rows[1] * row_b
    if name.endswith("[bounced]"):
        return None       # the elimination kernels over the rows the iteration kernel handed back: a device-side list, no model here
    if name.startswith("solve_iter_kernel"):
        # every candidate row is gathered once by this kernel (the rows it hands back are gathered again by the elimination
        # kernels; that second gather is not algorithmic)
        return "hbm", csr.nnz_iter * ent_b + csr.rows_iter * row_b
    m = re.match(r"solve_(directl|directw|rowsplit|wide)_kernel<([^>]*)>", name)
    if m:
        # MODE of the launch: 0 = whole rows (everything of the bin that is not split), 1 = the 2048-entry segments of the
        # rows above 4096 entries (their gathers; the written partial systems are not algorithmic bytes), 2 = the elimination
        # of the summed segments (reads partial systems only: no HBM model, no frac)
        targs = [a.strip() for a in m.group(2).split(",")]
        kind = m.group(1)
        mode = int(targs[1]) if kind == "directw" else (int(targs[-1]) if kind in ("directl", "rowsplit") else 0)
        b = 2 if kind in ("directl", "directw") else 3
        if mode == 0:             # (the rows that were never candidates of the iteration kernel)
            it_rows, it_nnz = (csr.rows_iter, csr.nnz_iter) if eng.iter_on else (0, 0)
            return "hbm", (nnz[b] - csr.nnz_split - it_nnz) * ent_b + (rows[b] - csr.rows_split - it_rows) * row_b
        if mode == 1:
            return "hbm", csr.nnz_split * ent_b
        return None
    if name.startswith("gram") and "reduce" not in name:
        # one pass over the fixed side (B_gram = 4 m f).  The f32-MFMA Gramian (gram_kernel) is priced against
        # the exact-f32 matrix peak; the split-bf16 one (gram6_kernel, f = 97 .. 144) runs its products at the bf16 rate and
        # is bound by that read
        if name.startswith("gram6"):
            return "hbm", 4.0 * f * eng.n_local[fixed]
        # (the kernel forms the upper 16 x 16 tiles only: f (f + 1) multiply-adds per row, not 2 f^2)
        return "mfma", 1.0 * f * (f + 1) * eng.n_local[fixed]
    return None


F16_MFMA_PEAK_TF = 2500.0   # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md): what the split-f16 tile products run on
HEAVY_T = 4096              # wmf_internal.h WMF_HEAVY_T: rows above it are accumulated in segments (MODE 1 / 2)


def mfma_issue_line(name, csr, avg_ms):
    """The second roofline of the split-f16 heavy-row kernels: the v_mfma_f32_16x16x32_f16 instructions they ISSUE (three per
    upper-triangle tile and 32-entry chunk: Al.Bh + Ah.Bl + Ah.Bh; two per tile product of the block elimination, hi | lo packed
    into K = 32) at 16 384 flop each against the dense f16 matrix peak.  Where this floor is above the HBM floor (k = 256: 9 ms
    against 5 ms per cfg5s item launch) the matrix pipe, not HBM, is what bounds the kernel as designed."""
    m = re.match(r"solve_(directl|rowsplit)_kernel<(\d+), (true|false), (true|false)(?:, \d+)?, (\d)>", name)
    if not m or m.group(4) != "true":
        return None
    nfb, mode = int(m.group(2)), int(m.group(5))
    tiles = nfb * (nfb + 1) // 2
    elim = nfb * (nfb - 1) + (nfb - 1) * nfb * (nfb + 1) // 3
    deg = csr.indptr[1:] - csr.indptr[:-1]
    normal, heavy = deg[(deg > 32) & (deg <= HEAVY_T)], deg[deg > HEAVY_T]
    if mode == 0:
        n_mfma = 3 * tiles * int(((normal + 31) // 32).sum().item()) + elim * int(normal.numel())
    elif mode == 1:
        n_mfma = 3 * tiles * int(((heavy + 31) // 32).sum().item())
    else:
        n_mfma = elim * int(heavy.numel())
    flops = 16384.0 * n_mfma
    if flops <= 0 or avg_ms <= 0:
        return None
    ach = flops / (avg_ms / 1e3) / 1e12
    return {"bound": "mfma", "issued_mfma_per_launch": n_mfma, "issued_tflop_per_launch": flops / 1e12, "achieved": ach,
            "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / F16_MFMA_PEAK_TF, "floor_ms": flops / (F16_MFMA_PEAK_TF * 1e12) * 1e3,
            "what": "v_mfma_f32_16x16x32_f16 issued by the split-f16 accumulation (3 per tile and 32 entries) and elimination (2 per tile product)"}


def transform_work(eng, rows):
    return "hbm", 8.0 * eng.f * rows                                         # one row read, one row written


def traffic_table(config):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_traffic.json")))
    if not files:
        return None, None
    return json.load(open(files[-1])).get("kernels", {}), os.path.relpath(files[-1], ROOT)


def traffic_of(table, name, side, both_sides):
    """HBM bytes per launch of ``name`` from the committed PMC profile (2 x FETCH_SIZE + WRITE_SIZE, the gfx950
    correction of MI355X_MICROARCH.md); per side when the profile kept the dispatches of the two half steps apart."""
    if not table:
        return None
    k = next((v for n, v in table.items() if n.startswith(name)), None)
    if not k:
        return None
    if both_sides and k.get("hbm_bytes_by_side"):
        return k["hbm_bytes_by_side"].get(side)
    return k.get("hbm_bytes_mean_corrected")


def run_workload(cfg_name, args, world, rank, dev, lib, scaling, with_cpu, zipf=None):
    zipf = args.zipf if zipf is None else float(zipf)
    n_users_1, n_items, dbar, k, bias = synth.CONFIGS[cfg_name]
    n_users = n_users_1 * world if scaling == "weak" else n_users_1
    gamma = 0.1
    seed = SEEDS[cfg_name]

    # ---- synthetic workload, generated on the GPU: every rank makes only ITS block of user rows ---------------------
    # The matrix is a fixed sequence of user blocks, each with its own random stream, so it is the same matrix whatever the
    # number of ranks: 8 blocks of n_users / 8 rows (strong scaling; one rank generates 8 / N of them), or one block of
    # the configuration's users per rank (weak scaling).  The engine then deals the rows of both sides by cost and moves
    # every stored entry to the owners of its user and of its item (AlsEngine.set_interactions_distributed).
    t0 = time.perf_counter()
    nb = world if (scaling == "weak" or 8 % world) else 8
    per_block = n_users // nb
    my_blocks = range(rank * nb // world, (rank + 1) * nb // world)
    parts = [synth.make_counts(per_block if b < nb - 1 else n_users - per_block * (nb - 1), n_items, dbar, seed, device=dev,
                               zipf_a=zipf, first_user=b * per_block) for b in my_blocks]
    first_user = my_blocks[0] * per_block
    ptrs, offset = [parts[0][0][:1]], 0
    for ip_b, _, _ in parts:
        ptrs.append(ip_b[1:] + offset)
        offset += int(ip_b[-1])
    indptr = torch.cat(ptrs)                        # this rank's user rows [first_user, first_user + len(indptr) - 1)
    indices = torch.cat([p[1] for p in parts])
    counts = torch.cat([p[2] for p in parts])
    del parts, ptrs
    nnz_t = torch.tensor([indices.numel()], dtype=torch.int64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(nnz_t)
    nnz = int(nnz_t.item())
    eng = AlsEngine(n_users, n_items, k, bias, gamma, device=dev, chunks=args.chunks or None)
    values = counts.clone()
    eng.K.confidence_transform(values, 10.0, 1.0, 0)
    eng.set_interactions_distributed(first_user, indptr, indices, values, balance=True)
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
        lib.wmf_profile_set_tag(0)
        eng.half_step("users")
        lib.wmf_profile_set_tag(1)
        eng.half_step("items")

    for _ in range(args.warmup):
        step()
    barrier()
    eng.check_numerics()
    for s_ in SIDES:
        eng.iter_stats(s_)                          # (clears the counters of the warm-up iterations)
    lib.wmf_profile_reset()
    lib.wmf_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.wmf_profile_enable(0)
    eng.check_numerics()
    table = _lib.profile_table(lib)
    lib.wmf_profile_reset()
    # what the matrix-free iteration kernel did with its candidates in the timed iterations (data dependent: see csrc/wmf_iter.hip)
    paths = {}
    for s_ in SIDES:
        st = eng.iter_stats(s_)
        cand = int(st[0] + st[1])
        if cand:
            n_rows = max(1, eng.n_local[s_]) * args.steps
            paths[s_] = {"iterated_share_of_rows": round(float(st[0]) / n_rows, 4), "handed_back_share_of_rows": round(float(st[1]) / n_rows, 6),
                         "applications_per_iterated_row": round(float(st[2]) / max(1, int(st[0])), 2),
                         "chebyshev_share_of_iterated": round(float(st[3]) / max(1, int(st[0])), 4)}
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- eval pass (not part of the timed step; reported separately) ------------------------
    shard = eng.make_eval_shard_distributed(first_user, indptr, indices, counts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sq, ab, cnt = eng.eval_sums(shard)
    torch.cuda.synchronize()
    t_eval = time.perf_counter() - t0

    # ---- per-kernel table and rooflines (this rank's launches) --------------------------------
    f = eng.f
    # counted traffic exists only for the workloads that were profiled (profiles/rNN_<config>_traffic.json are runs of the
    # UNIFORM matrices): a skewed variant has no profile of its own and reports null
    traffic, traffic_src = traffic_table(cfg_name) if not zipf else (None, None)
    names_both = {n for n, t, *_ in table if t == 0} & {n for n, t, *_ in table if t == 1}
    kernels, solve_ms = [], {s: 0.0 for s in SIDES}
    for name, tag, ms, launches, lo, hi in sorted(table, key=lambda e: -e[2]):
        side = SIDES[tag] if tag in (0, 1) else "?"
        entry = {"kernel": name, "half_step": side, "launches": int(launches), "avg_ms": ms / launches, "min_ms": lo,
                 "max_ms": hi, "total_ms": ms}
        work = None
        if side in SIDES:
            if name.startswith("transform"):
                # two launches per half step: the whitening of the fixed side, then the un-whitening of the solved rows
                fixed = "items" if side == "users" else "users"
                work = transform_work(eng, (eng.world * eng.rpr[fixed] + eng.n_local[side]) / 2.0)
            else:
                work = kernel_work(name, eng.csr[side], eng, side)
            if name.startswith(("solve_", "bias_adjust")):
                solve_ms[side] += ms / args.steps
        if work and work[1] > 0:
            bound, units = work
            peak, unit, scale = (HBM_PEAK_GBS, "GB/s", 1e9) if bound == "hbm" else (F32_MFMA_PEAK_TF, "TFLOP/s", 1e12)
            ach = units / (ms / launches / 1e3) / scale
            entry.update({"bound": bound, "algorithmic_units_per_launch": float(units), "achieved": ach, "peak": peak,
                          "unit": unit, "frac": ach / peak,
                          "traffic": traffic_of(traffic, name, side, name in names_both)})
            if bound == "hbm":
                line = mfma_issue_line(name, eng.csr[side], ms / launches)
                if line is not None:
                    entry["hbm_floor_ms"] = units / (HBM_PEAK_GBS * 1e9) * 1e3
                    entry["mfma_issue"] = line
        kernels.append(entry)
    cand = [e for e in kernels if "frac" in e]
    roofline = None
    if cand:
        dom = max(cand, key=lambda e: e["total_ms"])
        roofline = {"kernel": dom["kernel"], "half_step": dom["half_step"], "bound": dom["bound"], "achieved": dom["achieved"],
                    "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"], "traffic": dom["traffic"],
                    "traffic_source": (f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this workload, "
                                       "2 x FETCH + WRITE; not measured in this run") if dom["traffic"] is not None else None,
                    "algorithmic_units_per_launch": dom["algorithmic_units_per_launch"], "avg_launch_ms": dom["avg_ms"],
                    "launches": dom["launches"], "share_of_step": dom["total_ms"] / (elapsed * 1e3)}
        if "mfma_issue" in dom:
            roofline["hbm_floor_ms"] = dom["hbm_floor_ms"]
            roofline["mfma_issue"] = dom["mfma_issue"]
        if paths:
            roofline["paths"] = paths
    # the north star's named target: the per-user solve (every row kernel of the users half step together)
    half = {}
    for s in SIDES:
        c = eng.csr[s]
        b = c.nnz * (4 * f + 8) + eng.n_local[s] * (4 * f + 4)
        half[s] = {"algorithmic_bytes": float(b), "solve_kernels_ms": solve_ms[s],
                   "achieved": b / (solve_ms[s] / 1e3) / 1e9 if solve_ms[s] > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": b / (solve_ms[s] / 1e3) / 1e9 / HBM_PEAK_GBS if solve_ms[s] > 0 else None, "bound": "hbm",
                   "kernels": sorted({e["kernel"] for e in kernels if e["half_step"] == s and e["kernel"].startswith(("solve_", "bias_adjust"))})}
    epoch_bytes = sum(eng.algorithmic_bytes_half(s) for s in SIDES)
    if world > 1:
        eb = torch.tensor([epoch_bytes], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(eb)
        epoch_bytes = float(eb.item())

    if roofline is not None:                       # (the driver's record keeps "roofline" whole and drops keys it does not know)
        roofline["user_solve"] = half["users"]
        roofline["item_solve"] = half["items"]
    nnz_mine = torch.tensor([eng.csr["users"].nnz], dtype=torch.int64, device=dev)
    per_rank_nnz = [int(nnz_mine.item())]
    if world > 1:
        gathered = [torch.zeros_like(nnz_mine) for _ in range(world)]
        torch.distributed.all_gather(gathered, nnz_mine)
        per_rank_nnz = [int(t.item()) for t in gathered]
    out = {
        "value": (n_users + n_items) * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
        "config": {"workload": f"{cfg_name}: WMF k={k}{'+bias' if bias else ''}, {n_users}x{n_items} CSR, "
                               f"{nnz} nnz, alpha-log confidence, gamma={gamma}, zipf_a={zipf}",
                   "n_users": n_users, "n_items": n_items, "nnz": nnz, "k": k, "bias": bias,
                   "exchange": {s_: (("reduce" if eng.reduce[s_] else "need-list" if eng.sparse[s_] else "all-gather")
                                     + ("+pipelined" if eng.pipe[s_] else "")) + f" x{len(eng.chunk_bounds[s_])}"
                                for s_ in SIDES} if world > 1 or eng.exchange else None,
                   "sharding": f"users+items dealt by cost (nnz f^2 + f^3) over {world} GPU(s), every rank loads 1/{world} of the user "
                               f"rows; exchange users: "
                               f"{'reduce-scatter of partial systems' if eng.reduce['users'] else ('need-list all-to-all (%.0f %% of the rows per rank)' % (100 * eng.need['users']['needed_fraction']) if eng.sparse['users'] else 'all-gather')} in "
                               f"{len(eng.chunk_bounds['users'])} chunk(s), items: "
                               f"{'reduce-scatter of partial systems' if eng.reduce['items'] else 'all-gather'} in "
                               f"{len(eng.chunk_bounds['items'])} chunk(s); accumulation pipelined over arriving chunks: "
                               f"{[s_ for s_ in SIDES if eng.pipe[s_]] or 'no'}"},
        "nnz_per_s": 2.0 * nnz * args.steps / elapsed,
        "epoch_algorithmic_GBps": epoch_bytes * args.steps / elapsed / 1e9,
        "epoch_hbm_frac": epoch_bytes * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world),
        "roofline": roofline, "roofline_user_solve": half["users"], "roofline_item_solve": half["items"],
        "kernels": kernels,
        "eval": {"ms": t_eval * 1e3, "mse": sq / cnt if cnt else None},
        "setup_s": t_setup,
        "ranks": {"world_size": (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
                  "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                  "nnz_of_the_user_shard_per_rank": per_rank_nnz,
                  "bytes_this_rank_sends_per_half_step": {s_: eng.exchange_bytes_sent(s_) for s_ in SIDES}},
        "row_bins": {s: {"rows": eng.csr[s].bin_rows.tolist(), "nnz": eng.csr[s].bin_nnz.tolist(),
                         "rows_le8": eng.csr[s].rows8, "rows_split": eng.csr[s].rows_split} for s in SIDES},
    }
    if with_cpu:
        su = args.cpu_users or {64: 300_000, 128: 120_000, 16: 943, 256: 12_000}.get(k, 20_000)       # ~10-15 s of NumPy
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
        out["cpu_baseline"] = cpu_baseline(cfg_name, user_rows, item_rows, users_f, k, bias, gamma, n_items)
        if n_users > 2_000_000:
            out["cpu_baseline"]["note"] = ("the item-rows leg runs against RANDOM user factors (uniform [0, 1), same shape): the "
                                           "10 M x 129 device factors are not copied to the host for a timing that does not depend on "
                                           "the values")
        out["host_boundary"] = host_boundary(user_rows, k, bias, gamma, n_items)
        del users_f
    del eng, indptr, indices, counts, values, shard
    torch.cuda.empty_cache()
    return out


def float64_path(cfg_name, dev, lib, iters=2):
    """The reference's DEFAULT call -- train(cores = 4) on a SciPy (float64) count matrix -- keeps float64 factors through its
    Pool variants (RecModel/wmf_model.py:146-157, :242-309); here that is wmf_half_step_f64.  One ALS iteration of it on the
    same synthetic matrix as ``cfg_name``, timed after one warm-up iteration: ms per iteration, next to the float32 path's."""
    from recmodel_amd import WMF
    from recmodel_amd.engine import HipKernels
    n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg_name]
    K = HipKernels()
    ip, idx, val = synth.make_counts(n_users, n_items, dbar, SEEDS[cfg_name], device=dev)
    w = val.double()
    K.confidence_transform(w, 10.0, 1.0, 0)
    f = k + 1 if bias else k
    users_of = torch.repeat_interleave(torch.arange(n_users, device=dev), ip[1:] - ip[:-1])
    order = torch.argsort(idx * n_users + users_of, stable=True)           # the transpose (wmf_model.py:128), entries by (item, user)
    t_ip = torch.zeros(n_items + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(idx, minlength=n_items), 0, out=t_ip[1:])
    csr = {"users": (ip, idx.to(torch.int32), w, n_users),
           "items": (t_ip, users_of[order].to(torch.int32).contiguous(), w[order].contiguous(), n_items)}
    del order, users_of
    X = {"items": torch.from_numpy(np.ascontiguousarray(WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True,
                                                            bias=bias).items, dtype=np.float64)).to(dev), "users": None}
    n_max = max(n_users, n_items)
    ws = torch.empty(K.half_step_f64_workspace_bytes(f, n_max, n_max), dtype=torch.uint8, device=dev)
    fail = torch.zeros(4, dtype=torch.int32, device=dev)

    def half(side, fixed):
        indptr, indices, vals, n = csr[side]
        out = torch.empty(n, f, dtype=torch.float64, device=dev)
        K.half_step_f64(X[fixed], X[fixed].shape[0], f, bias, indptr, indices, vals, n, 0.1, out, ws, fail)
        X[side] = out

    times = []
    for it in range(iters + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        half("users", "items")
        half("items", "users")
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    assert int(fail[0]) == 0
    ms = min(times[1:])
    return {"workload": f"{cfg_name}: k={k}{'+bias' if bias else ''}, {n_users}x{n_items}, {int(idx.numel())} nnz, float64 counts and factors",
            "ms_per_iteration": ms, "row_updates_per_s": (n_users + n_items) / ms * 1e3, "first_iteration_ms": times[0],
            "what": "wmf_half_step_f64 (users + items): the reference's cores > 1 variants, float64 end to end"}


def _r(x, digits=4):
    """Numbers of the compact line: ``digits`` significant figures."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _side(h):
    return {"ms": _r(h["solve_kernels_ms"]), "algorithmic_GB": _r(h["algorithmic_bytes"] / 1e9), "achieved": _r(h["achieved"]),
            "frac": _r(h["frac"])} if h and h.get("frac") is not None else None


def compact_roofline(r):
    if not r:
        return None
    out = {k: _r(r[k], 6) for k in ("kernel", "half_step", "bound", "achieved", "peak", "unit", "frac", "traffic") if k in r}
    out["traffic_source"] = (r.get("traffic_source") or "")[:60] or None
    out.update({"algorithmic_units_per_launch": _r(r["algorithmic_units_per_launch"], 6), "avg_launch_ms": _r(r["avg_launch_ms"], 6),
                "launches": r["launches"], "share_of_step": _r(r["share_of_step"])})
    if r.get("paths"):
        out["paths"] = r["paths"]
    out["user_solve"], out["item_solve"] = _side(r.get("user_solve")), _side(r.get("item_solve"))
    return out


def compact_line(full, detail_path):
    """The one stdout line: the contract keys, config, roofline (dominant kernel + the two half steps' solves), cpu_baseline,
    ranks and one small object per further workload.  Everything else is in ``detail_path`` / on stderr."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 6), _r(line["ms_per_step"], 6)
    c = full["config"]
    line["config"] = {"workload": c["workload"], "n_users": c["n_users"], "n_items": c["n_items"], "nnz": c["nnz"], "k": c["k"],
                      "bias": c["bias"], "exchange": c["exchange"]}
    line["nnz_per_s"], line["epoch_hbm_frac"] = _r(full["nnz_per_s"]), _r(full["epoch_hbm_frac"])
    line["roofline"] = compact_roofline(full.get("roofline"))
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": _r(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                "sample": cpu["sample"][:200], "host_cpus": cpu.get("host_cpus")}
        ac = cpu.get("all_cores") or {}
        line["cpu_baseline"]["all_cores"] = ({"value": _r(ac["value"]), "cores": ac["cores"], "kind": ac["kind"],
                                              "sample": ac["sample"][:120]} if "value" in ac else ac)
    else:
        line["cpu_baseline"] = None
    rk = full.get("ranks") or {}
    line["ranks"] = {"world_size": rk.get("world_size"), "backend": rk.get("backend"),
                     "bytes_sent_per_half_step": rk.get("bytes_this_rank_sends_per_half_step")}
    hb = full.get("host_boundary")
    if hb:
        line["host_boundary_rows_per_s"] = _r(hb["value"])
    if full.get("also"):
        line["also"] = {}
        for name, r in full["also"].items():
            if "ms_per_step" not in r:                       # float64_path or an error record
                line["also"][name] = {k: _r(v) for k, v in r.items() if k in ("ms_per_iteration", "vs_float32_path", "error", "cfg3_ms_per_iteration", "cfg3_vs_float32_path")}
                continue
            ro = r.get("roofline") or {}
            line["also"][name] = {"ms_per_step": _r(r["ms_per_step"]), "value": _r(r["value"]), "kernel": ro.get("kernel"),
                                  "kernel_ms": _r(ro.get("avg_launch_ms")), "frac": _r(ro.get("frac")),
                                  "user_solve_frac": _r((ro.get("user_solve") or {}).get("frac")),
                                  "item_solve_frac": _r((ro.get("item_solve") or {}).get("frac")),
                                  "paths": ro.get("paths"),
                                  "cpu_baseline": _r((r.get("cpu_baseline") or {}).get("value"))}
    line["detail"] = detail_path
    return line


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

    scaling = args.scaling if args.scaling != "auto" else "strong"     # (one GPU: the first point of the same series)
    main_res = run_workload(args.config, args, world, rank, dev, lib, scaling,
                            with_cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
    # further workloads of the same run, under "also": names of synth.CONFIGS, optionally "name:zipf=A", comma separated.
    # auto on one GPU: cfg2 (configs[1], with its CPU baseline), then the skewed variant of the main workload and cfg5's
    # one-GPU slice with and without the power law (the README's other rows, driver-timed); on several GPUs the bias-free cfg4
    also_arg = args.also
    if also_arg == "auto":
        also_arg = {"cfg3": "cfg2,cfg3:zipf=1.1,cfg5s,cfg5s:zipf=1.1" if world == 1 else "cfg4"}.get(args.config, "none")
        if args.zipf:
            also_arg = "none"
    also = None
    for n_also, item in enumerate([x for x in also_arg.split(",") if x and x != "none"]):
        name, _, opt = item.partition(":")
        if name not in synth.CONFIGS:
            continue
        z = float(opt.split("=")[1]) if opt.startswith("zipf=") else 0.0
        r2 = run_workload(name, args, world, rank, dev, lib, scaling,
                          with_cpu=(n_also == 0 and rank == 0 and world == 1 and not args.no_cpu_baseline), zipf=z)
        also = also or {}
        also[item.replace(":zipf=", "_zipf")] = {"metric": "ALS user+item row-updates/sec", "unit": "row-updates/s", **r2}

    f64_block = None
    if rank == 0 and world == 1 and args.config == "cfg3" and args.also == "auto" and not args.zipf:
        try:
            f64_block = float64_path("cfg2", dev, lib)
            if also and "cfg2" in also:
                f64_block["vs_float32_path"] = f64_block["ms_per_iteration"] / also["cfg2"]["ms_per_step"]
            # ... and at the headline configuration's size (the faster of two timed iterations after one warm-up: 10 M x 129 float64 factors)
            torch.cuda.empty_cache()
            big = float64_path("cfg3", dev, lib, iters=2)
            f64_block["cfg3_ms_per_iteration"] = big["ms_per_iteration"]
            f64_block["cfg3_vs_float32_path"] = big["ms_per_iteration"] / main_res["ms_per_step"]
            del big
        except Exception as exc:                      # reported, never silently dropped
            f64_block = {"error": repr(exc)}
    if also is not None and f64_block is not None:
        also["float64_path"] = f64_block
    out = {
        "metric": "ALS user+item row-updates/sec", "value": main_res.pop("value"),
        "unit": "row-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res.pop("ms_per_step"), "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "precision_note": "float32 storage, float32 accumulation everywhere; Gramian fp64 across waves; the GEMM-shaped "
                          "products run on f16 / bf16 MFMAs from SPLIT float32 operands: whitened rows (|v| <= 1) as two f16 "
                          "parts (22 bits, three MFMAs per tile, every product exact in f32), factor matrices in the Gramian and "
                          "the row transforms as three bf16 parts (exact split, six MFMAs per tile); results within 1.4x of the "
                          "f32-MFMA error (DESIGN.md section 5); parity tolerance unchanged (tests/test_gpu_parity.py)",
    }
    cpu = main_res.pop("cpu_baseline", None)
    out.update(main_res)
    out["cpu_baseline"] = cpu
    out["also"] = also
    if rank == 0:
        # The FULL record (per-kernel tables of every workload, row bins, host boundary, notes) goes to a file and to stderr;
        # stdout carries ONE compact line (a few KB) that the driver's record can hold whole.
        detail_path = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as fh:
                json.dump(out, fh)
        except OSError as exc:
            detail_path = f"(not written: {exc})"
        for tag, r in [(args.config, out)] + [(n, r) for n, r in (also or {}).items() if "kernels" in r]:
            print(f"[bench] {tag}: {r['ms_per_step']:.3f} ms per iteration", file=sys.stderr)
            for e in r["kernels"][:8]:
                frac = f"frac {e['frac']:.3f}" if e.get("frac") is not None else "(no HBM / MFMA model)"
                print(f"[bench]   {e['kernel']:<52} {e['half_step']:<6} avg {e['avg_ms']:8.4f} ms  {frac}", file=sys.stderr)
        print(f"[bench] full record: {detail_path}", file=sys.stderr, flush=True)
        line = compact_line(out, os.path.relpath(detail_path, ROOT) if not detail_path.startswith("(") else detail_path)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or force_exchange:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Turn the output of tools/profile_bench.sh (gpurun_out/<tag>/) into the committed profile files:
    profiles/<prefix>_kernel_stats.csv       rocprofv3 --kernel-trace --stats, this library's kernels only
    profiles/<prefix>_traffic.json           FETCH_SIZE / WRITE_SIZE per dispatch, with the gfx950 correction
    profiles/<prefix>_bench_under_rocprof.json
Usage: python tools/collect_profiles.py gpurun_out/prof_r01_cfg2 r01_cfg2 "<bench arguments used>"
"""
import csv
import json
import os
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
bench_args = sys.argv[3] if len(sys.argv) > 3 else "--steps 5 --warmup 1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
OURS = re.compile(r"(gram\d*_|factorize|transform\d*_kernel<\d|solve_|combine_segments|eval_|predict_|confidence_|bias_adjust|spmm_|topk_|score_|csr_)")


def short(name):
    """'void solve_low_kernel<4, 2, true>(int const*, ...)' -> 'solve_low_kernel<4, 2, true>'"""
    name = name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
mine = [r for r in rows if OURS.search(r["Name"]) and "rocprim" not in r["Name"]]
with open(os.path.join(out, prefix + "_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    total = sum(float(r["TotalDurationNs"]) for r in mine)
    for r in mine:
        r = dict(r)
        r["Percentage"] = f"{100 * float(r['TotalDurationNs']) / total:.2f}"      # share among this library's kernels
        w.writerow(r)

line = open(os.path.join(src, "bench_traced.json")).read().strip().splitlines()[-1]
traced = json.loads(line)
detail = os.path.join(src, "bench_detail_traced.json")       # round 4: the stdout line is compact, the kernel table is in the detail file
if os.path.exists(detail):
    traced = {**json.load(open(detail)), "stdout_line": traced}
# kernels that ran in both half steps of an iteration (bench.py's table): their dispatches alternate users, items
sides = {}
for e in traced.get("kernels", []):
    if e["launches"] == traced["steps"]:               # once per half step (the row transform runs twice: no parity rule)
        sides.setdefault(e["kernel"], set()).add(e["half_step"])
raw = json.load(open(os.path.join(src, "traffic_raw.json")))
kernels = {}
for name, v in raw.items():
    if not OURS.search(name) or "rocprim" in name:
        continue
    fetch, write = v.get("FETCH_SIZE_KB_per_dispatch_mean"), v.get("WRITE_SIZE_KB_per_dispatch_mean", 0.0)
    by_side = None
    fl, wl = v.get("FETCH_SIZE_KB_per_dispatch"), v.get("WRITE_SIZE_KB_per_dispatch")
    if sides.get(short(name)) == {"users", "items"} and fl and wl and len(fl) == len(wl) and len(fl) % 2 == 0:
        by_side = {s: (2.0 * sum(fl[i::2]) + sum(wl[i::2])) / (len(fl) // 2) * 1024.0 for i, s in enumerate(("users", "items"))}
    kernels[short(name)] = {
        "hbm_bytes_by_side": by_side,
        "dispatches": v.get("dispatches"),
        "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
        "FETCH_SIZE_KB_largest_dispatch": v.get("FETCH_SIZE_KB_per_dispatch_max"),
        "WRITE_SIZE_KB_largest_dispatch": v.get("WRITE_SIZE_KB_per_dispatch_max"),
        "hbm_bytes_mean_corrected": None if fetch is None else (2.0 * fetch + (write or 0.0)) * 1024.0,
    }
json.dump({
    "command": f"rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate runs) -- python3 bench.py --no-cpu-baseline {bench_args}",
    "workload": prefix.split("_", 1)[1],
    "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts half of wide/segment reads), write bytes = WRITE_SIZE x 1024",
    "kernels": kernels}, open(os.path.join(out, prefix + "_traffic.json"), "w"), indent=1)

json.dump(traced, open(os.path.join(out, prefix + "_bench_under_rocprof.json"), "w"), indent=1)
print("wrote", [f for f in sorted(os.listdir(out)) if f.startswith(prefix)])
for r in mine[:8]:
    print(short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e6:.3f} ms")

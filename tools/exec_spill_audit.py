"""Audit of a gfx950 kernel's VGPR spill code against its EXEC mask (hipcc -S --cuda-device-only output).

Builds the control flow graph of one kernel from the assembly text (labels, s_branch / s_cbranch_*), propagates over it how
many mask-narrowing operations EXEC is under (s_and_saveexec_b64 / s_andn2_saveexec_b64 / v_cmpx / s_andn2_b64 exec: +1 level
for the saveexec forms; s_or_b64 exec, exec, s[..]: -1; merge = maximum), and reports for every scratch slot the levels its
spill stores and reloads execute at.  A slot stored at a DEEPER level than one of its reloads is a value written under a
partial mask and read back under a fuller one: the lanes that were off get whatever the scratch slot held before, which
depends on which wave last owned that memory -- i.e. on co-residency.
CAVEAT: the nesting level is a conservative over-approximation (text patterns, maximum at joins, capped): the list is a set of
CANDIDATES to read in the assembly, not a proof.  It was written for the round-3 finding in wmf_rowsplit.hip.
Usage: python tools/exec_spill_audit.py file.s <substring of the kernel symbol> [-v]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if "s_endpgm" in lines[i])

# ---- basic blocks
blocks, order, cur = {}, [], "entry"
blocks[cur] = []
order.append(cur)
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        order.append(cur)
        continue
    s = l.strip()
    if not s or s.startswith((";", ".")):
        continue
    blocks[cur].append((i - start, s))
succ = collections.defaultdict(list)
for bi, b in enumerate(order):
    ins = blocks[b]
    fall = True
    for _, s in ins:
        op = s.split()[0]
        if op == "s_branch":
            succ[b].append(s.split()[1])
            fall = False
        elif op.startswith("s_cbranch"):
            succ[b].append(s.split()[1])
        elif op in ("s_endpgm", "s_setpc_b64"):
            fall = False
    if fall and bi + 1 < len(order):
        succ[b].append(order[bi + 1])


def step(depth, s):
    op = s.split()[0]
    if op in ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_or_saveexec_b64") or op.startswith("v_cmpx"):
        return min(depth + 1, 6)            # (capped: an unmatched narrowing inside a loop must not grow without bound)
    if op == "s_or_b64" and re.match(r"s_or_b64\s+exec,\s*exec,", s):
        return max(0, depth - 1)
    if op == "s_mov_b64" and re.match(r"s_mov_b64\s+exec,\s*-1", s):
        return 0
    return depth


state_in = {b: None for b in order}
state_in["entry"] = 0
work = collections.deque(["entry"])
while work:
    b = work.popleft()
    d = state_in[b]
    for _, s in blocks[b]:
        d = step(d, s)
    for t in succ[b]:
        if t in state_in and (state_in[t] is None or d > state_in[t]):
            state_in[t] = d
            work.append(t)

stores, loads = collections.defaultdict(list), collections.defaultdict(list)
for b in order:
    d = state_in[b]
    if d is None:
        continue
    for ln, s in blocks[b]:
        op = s.split()[0]
        m = re.search(r"offset:(\d+)", s)
        off = int(m.group(1)) if m else 0
        if op.startswith("scratch_store") and "Spill" in s:
            stores[off].append((ln, b, d))
        elif op.startswith("scratch_load") and "Reload" in s:
            loads[off].append((ln, b, d))
        d = step(d, s)
print(f"{key}: {end - start} lines, {len(order)} blocks, {sum(map(len, stores.values()))} spill stores, {sum(map(len, loads.values()))} reloads")
bad = []
for off in sorted(set(stores) | set(loads)):
    sd = sorted({d for _, _, d in stores[off]})
    ld = sorted({d for _, _, d in loads[off]})
    risky = bool(sd and ld and max(sd) > min(ld))
    if risky:
        bad.append(off)
    if risky or "-v" in sys.argv:
        print(f"  slot {off:4d}: stored at EXEC level(s) {sd} ({len(stores[off])} stores), reloaded at level(s) {ld} ({len(loads[off])} reloads)"
              + ("   <-- stored under a narrower mask than a reload" if risky else ""))
        if risky:
            for ln, b, d in stores[off]:
                if d > min(ld):
                    print(f"        store  line {ln:6d} {b} level {d}")
            for ln, b, d in loads[off]:
                if d < max(sd):
                    print(f"        reload line {ln:6d} {b} level {d}")
print(f"{len(bad)} slot(s) stored under a narrower EXEC mask than one of their reloads: {bad}")

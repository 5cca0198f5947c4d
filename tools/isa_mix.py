"""Instruction mix per basic block of one kernel in a gfx950 .s file (hipcc -S --cuda-device-only).
Usage: python tools/isa_mix.py file.s <substring of the kernel symbol> [min instructions per block to print]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].strip() == "s_endpgm")


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma_bf16" if "bf16" in op or "f16" in op else ("mfma_f64" if "f64" in op else "mfma_f32")
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane_sgpr"
    if op.startswith(("v_permlane", "v_mov_b32_dpp")) or "_dpp" in op:
        return "valu_dpp"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "valu_cmp"
    if op.startswith("v_cvt") or op.startswith("v_pk_") or op.startswith("v_perm") or op.startswith("v_and_or") or op.startswith("v_lshl") or op.startswith("v_bfi"):
        return "valu_cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, cur, name = [], collections.Counter(), "entry"
total = collections.Counter()
for i in range(start + 1, end + 1):
    t = lines[i].strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            blocks.append((name, cur))
            cur, name = collections.Counter(), m.group(1)
        continue
    m = re.match(r"^(\.LBB\w+):", t)
    if m:
        blocks.append((name, cur))
        cur, name = collections.Counter(), m.group(1)
        continue
    op = t.split()[0]
    c = cls(op)
    cur[c] += 1
    total[c] += 1
blocks.append((name, cur))
print("TOTAL", dict(total), "instructions", sum(total.values()))
for nm, c in blocks:
    n = sum(c.values())
    if n >= min_n:
        print(f"{nm:14s} n={n:5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))

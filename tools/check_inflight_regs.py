"""Build-time check for wmf_directl.hip: the destination registers of an inline-asm ds_read are written by the hardware
some time after the instruction issues; hipcc believes they are written at once.  Between each such read and the
inline-asm `s_waitcnt lgkmcnt(0)` that retires it nothing may touch those registers (a copy the register allocator
inserts there would copy stale data).  Parses the gfx950 assembly of the file and reports violations.
Usage: python tools/check_inflight_regs.py [path/to/wmf_directl.hip] [extra compiler flags ...]
Run by recmodel_amd/csrc/Makefile on the flags of the build itself (a violation fails the build).  Since round 3 the scan
follows the control flow of every kernel (both successors of a conditional branch), not the text order of the assembly."""
import re, subprocess, sys, os, tempfile

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "recmodel_amd", "csrc", "wmf_directl.hip")
extra = sys.argv[2:]
out = os.path.join(tempfile.mkdtemp(prefix="wmf_inflight_"), "wmf_directl_check.s")
subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", *extra, "-S",
                "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
reg_re = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    r = set()
    for m in reg_re.finditer(text):
        if m.group(1) is not None:
            r.add(int(m.group(1)))
        else:
            r.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return r


# ---- control-flow-aware scan (round 3).  Every kernel of the file is walked along its control flow graph: the state is the queue
# of inline-asm reads still in flight (LDS reads retire in issue order), a conditional branch continues on both successors, an
# unconditional one on its target, and a (position, queue) pair is expanded once.  With an empty queue a position is visited once,
# and a non-empty queue lives for a few dozen instructions, so the walk is linear in practice.
def parse_functions(lines):
    funcs, cur = [], None
    for n, ln in enumerate(lines, 1):
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            cur = {"name": m.group(1), "ins": [], "labels": {}}
            funcs.append(cur)
            continue
        if cur is None:
            continue
        if t.startswith(".end_amdhsa_kernel") or t.startswith(".section") or t.startswith(".Lfunc_end"):
            cur = None if t.startswith(".Lfunc_end") else cur
            continue
        m = re.match(r"^([.\w$]+):", t)
        if m and not t.startswith(";"):
            cur["labels"][m.group(1)] = len(cur["ins"])
            continue
        if t.startswith(";;#ASMSTART"):
            cur["ins"].append((n, "#ASMSTART"))
            continue
        if t.startswith(";;#ASMEND"):
            cur["ins"].append((n, "#ASMEND"))
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append((n, t.split(";")[0].strip()))
    return funcs


def touched_by(code, inflight_regs):
    touched = regs_of(code) & inflight_regs
    if touched and code.startswith("v_pk_"):
        # packed op: a source pair v[a:b] is read through op_sel (low result) / op_sel_hi (high result); a half that neither
        # selects is not read
        ops = [o.strip() for o in code.split("op_sel")[0].strip().split(None, 1)[1].split(",")]
        sel, sel_hi = [0, 0, 0], [1, 1, 1]
        m = re.search(r"op_sel:\[([\d,]+)\]", code)
        if m:
            sel = [int(x) for x in m.group(1).split(",")][:3]
        m = re.search(r"op_sel_hi:\[([\d,]+)\]", code)
        if m:
            sel_hi = [int(x) for x in m.group(1).split(",")][:3]
        used = regs_of(ops[0])
        for i, o in enumerate(ops[1:4]):
            mm = re.match(r"v\[(\d+):(\d+)\]", o)
            if not mm:
                used |= regs_of(o)
                continue
            lo_r, hi_r = int(mm.group(1)), int(mm.group(2))
            if sel[i] == 0 or sel_hi[i] == 0:
                used.add(lo_r)
            if sel[i] == 1 or sel_hi[i] == 1:
                used.add(hi_r)
        touched &= used
    return touched


bad_lines = {}
reads_seen = set()
for fn in parse_functions(lines):
    ins, labels = fn["ins"], fn["labels"]
    # is instruction i inside an inline-asm region?  (regions do not span branches: computed in text order)
    in_asm_at, flag = [], False
    for _, code in ins:
        if code == "#ASMSTART":
            flag = True
        elif code == "#ASMEND":
            flag = False
        in_asm_at.append(flag)
    work, seen = [(0, ())], set()
    while work:
        pc, queue = work.pop()
        while pc < len(ins):
            if (pc, queue) in seen:
                break
            seen.add((pc, queue))
            n, code = ins[pc]
            if code.startswith("#"):
                pc += 1
                continue
            if in_asm_at[pc] and code.startswith("ds_read"):
                reads_seen.add(n)
                if any(ln_ == n for ln_, _ in queue):               # the same read issued again before its wait retired it: a loop
                    if n not in bad_lines:                          # whose back edge bypasses the wait
                        bad_lines[n] = f"{fn['name']}: line {n}: `{code}` is issued again while still in flight"
                    break
                queue = queue + ((n, frozenset(regs_of(code.split(",")[0]))),)
                pc += 1
                continue
            m_wait = re.search(r"lgkmcnt\((\d+)\)", code) if code.startswith("s_waitcnt") else None
            if m_wait and (in_asm_at[pc] or int(m_wait.group(1)) == 0):
                # a counted wait of the kernel's own (inline asm) leaves its N youngest reads in flight; a compiler wait is only
                # trusted when it drains the counter (its counts do not include the inline-asm reads)
                keep = int(m_wait.group(1))
                queue = queue[len(queue) - keep:] if keep else ()
                pc += 1
                continue
            if queue:
                inflight = set().union(*[r for _, r in queue])
                t_ = touched_by(code, inflight)
                if t_ and n not in bad_lines:
                    since = min(ln_ for ln_, r in queue if r & t_)
                    bad_lines[n] = f"{fn['name']}: line {n}: `{code}` touches v{sorted(t_)} in flight since line {since}"
            if code.startswith("s_endpgm") or code.startswith("s_setpc") or code.startswith("s_swappc"):
                break
            m_br = re.match(r"^(s_branch|s_cbranch_\w+)\s+([.\w$]+)", code)
            if m_br:
                tgt = labels.get(m_br.group(2))
                if tgt is not None:
                    work.append((tgt, queue))
                if m_br.group(1) == "s_branch":
                    break
            pc += 1
for n in sorted(bad_lines):
    print(bad_lines[n])
print(f"{len(reads_seen)} inline-asm ds_reads checked, {len(bad_lines)} violations")
sys.exit(1 if bad_lines else 0)

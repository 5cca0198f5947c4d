"""Build-time check for wmf_directl.hip: the destination registers of an inline-asm ds_read are written by the hardware
some time after the instruction issues; hipcc believes they are written at once.  Between each such read and the
inline-asm `s_waitcnt lgkmcnt(0)` that retires it nothing may touch those registers (a copy the register allocator
inserts there would copy stale data).  Parses the gfx950 assembly of the file and reports violations.
Usage: python tools/check_inflight_regs.py [path/to/wmf_directl.hip] [extra compiler flags ...]
Run by recmodel_amd/csrc/Makefile on the flags of the build itself (a violation fails the build); the scan follows the
text order of the assembly, not its control flow, which is enough for the straight-line read / wait pairs of this file."""
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


bad = reads = 0
inflight = {}          # register -> line number of the read that targets it
order = []             # in-flight reads, oldest first: (line, registers); LDS reads retire in issue order
in_asm = False
kernel = None
for n, ln in enumerate(lines, 1):
    t = ln.strip()
    if t.startswith("_Z") and t.endswith(":") or re.match(r"^_Z\w+:", t):
        kernel, inflight, order = t.split(":")[0], {}, []
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    code = t.split(";")[0]
    if in_asm and code.startswith("ds_read"):
        dst = code.split(",")[0]
        for r in regs_of(dst):
            inflight[r] = n
        order.append((n, regs_of(dst)))
        reads += 1
        continue
    m_wait = re.search(r"lgkmcnt\((\d+)\)", code) if code.startswith("s_waitcnt") else None
    if m_wait and (in_asm or int(m_wait.group(1)) == 0):
        # a counted wait of the kernel's own (inline asm) leaves its N youngest reads in flight; a compiler wait is only
        # trusted when it drains the counter (its counts do not include the inline-asm reads)
        keep = int(m_wait.group(1))
        order = order[len(order) - keep:] if keep else []
        inflight = {r: ln_ for ln_, regs in order for r in regs}
        continue
    touched = regs_of(code) & set(inflight)
    if touched and code.strip().startswith("v_pk_"):
        # packed op: a source pair v[a:b] is read through op_sel (low result) / op_sel_hi (high result); a half that neither
        # selects is not read
        ops = [o.strip() for o in code.split("op_sel")[0].strip().split(None, 1)[1].split(",")]
        sel = [0, 0, 0]
        sel_hi = [1, 1, 1]
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
    if touched:
        bad += 1
        print(f"{kernel}: line {n}: `{code.strip()}` touches v{sorted(touched)} in flight since line {min(inflight[r] for r in touched)}")
print(f"{reads} inline-asm ds_reads checked, {bad} violations")
sys.exit(1 if bad else 0)

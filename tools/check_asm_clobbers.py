"""Build-time source check: an inline-asm block that contains a scalar-ALU instruction which writes SCC (shifts, logic,
add / sub, compares, min / max, bit counts ...) must name "scc" in its clobber list -- hipcc does not look into the block, so
without the clobber it may keep a live SCC value (s_cmp / s_cbranch_scc, s_add_u32 / s_addc_u32 pairs) across it.
Usage: python tools/check_asm_clobbers.py file.hip file.h ...   (run by recmodel_amd/csrc/Makefile; a violation fails the build)"""
import re
import sys

SCC_WRITERS = re.compile(r"\bs_(lshl|lshr|ashr|and|or|xor|nand|nor|xnor|andn2|orn2|not|add|sub|addc|subb|abs|min|max|cmp|cmpk|bitcmp|"
                         r"bfe|bcnt0|bcnt1|ff0|ff1|flbit|wqm|quadmask|absdiff|lshl[1-4]_add|mul_hi|cselect|cmov)[a-z0-9_]*\b")
# s_cselect / s_cmov READ scc only, s_mul_i32 / s_mul_hi do not write it: not violations
READ_ONLY = re.compile(r"\bs_(cselect|cmov|mul_hi)[a-z0-9_]*\b")


def asm_blocks(text):
    for m in re.finditer(r"\basm\b(\s+volatile)?\s*\(", text):
        depth, i = 1, m.end()
        while i < len(text) and depth:
            c = text[i]
            if c == '"':                                   # skip string literals
                i += 1
                while text[i] != '"':
                    i += 2 if text[i] == "\\" else 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            i += 1
        yield text.count("\n", 0, m.start()) + 1, text[m.start(): i]


bad = checked = 0
for path in sys.argv[1:]:
    src = open(path).read()
    for line, block in asm_blocks(src):
        strings = " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', block))
        writers = [w.group(0) for w in SCC_WRITERS.finditer(strings) if not READ_ONLY.match(w.group(0))]
        if not writers:
            continue
        checked += 1
        if '"scc"' not in block:
            bad += 1
            print(f"{path}:{line}: asm block runs {sorted(set(writers))} without an \"scc\" clobber")
print(f"{checked} asm blocks with SCC-writing scalar instructions checked, {bad} violations")
sys.exit(1 if bad else 0)

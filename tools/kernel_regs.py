"""Registers, spills, scratch and LDS of every kernel in a gfx950 .s file (hipcc -S --cuda-device-only).
Usage: python tools/kernel_regs.py file.s [substring]"""
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
# the metadata of a kernel is one YAML list item: "  - .agpr_count: ..." up to the next "  - .a"
for blk in re.split(r"\n  - (?=\.agpr_count)", text)[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    if key in name:
        print(f"{name[:70]:<70} vgpr {g('vgpr_count'):>3} agpr {g('agpr_count'):>3} vspill {g('vgpr_spill_count'):>3} "
              f"sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")

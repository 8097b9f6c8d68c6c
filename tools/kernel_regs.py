#!/usr/bin/env python3
"""Register / spill / LDS summary per kernel from a --save-temps gfx950 .s file: python tools/kernel_regs.py file.s [name-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    d = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    name = d.get("name", "?")
    if flt in name:
        print(f"{name:60s} vgpr {d.get('vgpr_count'):>4s} agpr {d.get('agpr_count'):>3s} spill v{d.get('vgpr_spill_count')} s{d.get('sgpr_spill_count')} scratch {d.get('private_segment_fixed_size'):>4s} lds {d.get('group_segment_fixed_size')}")

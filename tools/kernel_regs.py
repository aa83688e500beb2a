#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures from a gfx950 ISA listing (hipcc -S --cuda-device-only)."""
import re, sys
txt = open(sys.argv[1]).read()
for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    name = re.sub(r"^_ZN2cm2m[hf]\d+", "", name)
    print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5}  {name[:90]}")

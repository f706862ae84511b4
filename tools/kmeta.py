"""Prints name, VGPRs, SGPR/VGPR spill counts and scratch bytes of every kernel in an assembly file (hipcc -S)."""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
    b = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, b) or [None, "?"])[1]
    print(g("name")[:70], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "sspill", g("sgpr_spill_count"), "vspill", g("vgpr_spill_count"),
          "scratch", g("private_segment_fixed_size"), "lds", g("group_segment_fixed_size"))

#!/usr/bin/env python3
"""Development: instruction mix per basic block of one kernel in hipcc's gfx950 assembly listing.
usage: hipcc ... -save-temps=obj -c jade_hip.hip -o /tmp/isa/jh.o ; isa_blocks.py /tmp/isa/*gfx950.s _Z7k_trace [min_instructions]"""
import re
import sys

path, kernel = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 8
text = open(path).read()
start = text.index("\n%s" % kernel)
body = text[start : text.index("s_endpgm", start)]
blocks, cur = [], ("entry", [])
for line in body.split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
        continue
    t = line.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t)
blocks.append(cur)
tot = 0
for name, ins in blocks:
    count = lambda *pre: sum(1 for i in ins if i.startswith(pre))  # noqa: E731
    tot += len(ins)
    if len(ins) < least:
        continue
    br = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    print("%-10s n=%4d valu=%4d (pk %3d) salu=%3d ds=%3d vmem=%3d  %s" % (
        name, len(ins), count("v_"), count("v_pk_"), count("s_"), count("ds_"),
        count("global_", "flat_", "buffer_", "scratch_"), " ".join(br)))
print("total instructions: %d" % tot)

#!/usr/bin/env python3
"""Scratch (spill) instructions inside loops of one kernel's assembly (tools/kres.sh writes /tmp/asm/k.s): depth, line, block, instruction.
A reload in the PANOC step loop (depth >= 2) costs an L2 round trip per step -- and fabric traffic once the L-BFGS rings have evicted it."""
import re, sys
lines = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/asm/k.s").read().split("\n")
depth, name = 0, ""
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: depth, name = 0, m.group(1)
    m2 = re.search(r"Depth=(\d+)", l)
    if l.startswith(";") and m2: depth = max(depth, int(m2.group(1)))
    if re.match(r"^\s+scratch_", l) and depth >= int(sys.argv[2] if len(sys.argv) > 2 else 2):
        print(depth, i + 1, name, l.strip())

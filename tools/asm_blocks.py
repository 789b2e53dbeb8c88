"""Per-basic-block instruction classes of one kernel in hipcc's -save-temps assembly (static counts; a reading aid for the
solve kernel: which blocks carry the moves, lane reads and SGPR-spill traffic).  usage: asm_blocks.py file.s kernel-substring [min]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("i") or (l.startswith("_Z") and key in l.split(":")[0]))
cls = [("mov64", r"v_mov_b64"), ("mov32", r"v_mov_b32_e32|v_accvgpr"), ("dpp", r"_dpp|permlane"), ("rdl", r"v_readlane|v_readfirstlane"),
       ("wrl", r"v_writelane"), ("f64", r"_f64"), ("cnd", r"v_cndmask"), ("valu", r"^v_"), ("nop", r"s_nop"), ("wait", r"s_waitcnt"),
       ("salu", r"^s_"), ("lds", r"^ds_"), ("vmem", r"^(global|scratch|buffer|flat)")]
blocks, cur = [], None
for i in range(start, len(lines)):
    l = lines[i]
    if l.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+|_Z\S+):\s*(;.*)?", l)
    if m:
        cur = {"name": m.group(1)[:14], "line": i + 1, "note": (m.group(2) or "").strip(), "n": 0, "tot": {}}
        blocks.append(cur); continue
    if cur is not None and l.startswith(";") and ("Loop" in l or "Depth" in l): cur["note"] += " " + l.strip("; ").strip()
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m and cur is not None:
        op = m.group(1); cur["n"] += 1
        for c, pat in cls:
            if re.search(pat, op): cur["tot"][c] = cur["tot"].get(c, 0) + 1; break
tot = {}
for b in blocks:
    for c, v in b["tot"].items(): tot[c] = tot.get(c, 0) + v
    if b["n"] >= minn:
        print(f"{b['line']:6d} {b['name']:14s} n={b['n']:4d} " + " ".join(f"{c}={b['tot'][c]}" for c, _ in cls if c in b["tot"]) + "  " + b["note"][:90])
print("TOTAL", sum(b["n"] for b in blocks), tot)

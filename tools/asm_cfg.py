"""Walks the basic blocks of one kernel (output of tools/kres.sh: /tmp/asm/k.s) and prints, per block, the instruction classes
and the branch targets -- to follow one path of the solver's state machine by hand.  usage: asm_cfg.py [k.s] [first_line]"""
import re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "/tmp/asm/k.s"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lines = open(src).read().split("\n")
cls = [("mov64", r"v_mov_b64"), ("mov32", r"v_mov_b32_e32"), ("dpp", r"_dpp|permlane"), ("rdl", r"v_readlane|v_readfirstlane"),
       ("wrl", r"v_writelane"), ("f64", r"_f64"), ("valu", r"^v_"), ("nop", r"s_nop"), ("wait", r"s_waitcnt"),
       ("smov", r"^s_mov"), ("salu", r"^s_"), ("lds", r"^ds_"), ("scr", r"^scratch"), ("vmem", r"^(global|buffer|flat)")]
cur = None
out = []
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if m:
        cur = {"name": m.group(1), "line": i + 1, "note": (m.group(2) or "").strip()[:60], "n": 0, "tot": {}, "br": []}
        out.append(cur); continue
    m = re.match(r"^; %bb\.(\d+):\s*(;.*)?", l)
    if m and cur is not None:   # fallthrough block without a label
        cur = {"name": "  bb." + m.group(1), "line": i + 1, "note": (m.group(2) or "").strip()[:60], "n": 0, "tot": {}, "br": []}
        out.append(cur); continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", l)
    if m and cur is not None:
        op = m.group(1); cur["n"] += 1
        if "branch" in op: cur["br"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + ":" + m.group(2).split()[0].replace(".LBB", "B"))
        for c, pat in cls:
            if re.search(pat, op): cur["tot"][c] = cur["tot"].get(c, 0) + 1; break
for b in out:
    if b["line"] >= first:
        print(f"{b['line']:5d} {b['name']:12s} n={b['n']:3d} " + " ".join(f"{c}={b['tot'][c]}" for c, _ in cls if c in b["tot"]) + "  -> " + " ".join(b["br"]) + "  " + b["note"])

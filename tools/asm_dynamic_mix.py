#!/usr/bin/env python3
"""DYNAMIC instruction mix of the benchmark solve kernel from its assembly (tools/kres.sh -> /tmp/asm/k.s) and the per-solve
execution counts of the solver's loops (tools/phase_prof.py on a -DMPC_PROFILE build: evaluations per call site, PANOC steps).

hipcc annotates every basic block with the loop it belongs to ("in Loop: Header=BB14_262 Depth=3").  The kernel has three call sites
of the evaluation -- the state machine (ten times per solve), the Lipschitz test and the line search of the step loop -- each inlined
as its own loop nest; this tool finds them by their signature (the 213-instruction heading block with 34 DPP moves is the entry of an
evaluation), assigns every block the execution count of the innermost loop nest it belongs to, and sums instruction classes:

  spill_sgpr   v_writelane_b32 into / v_readlane_b32 out of the VGPRs the register allocator reserved for SGPR spills
  spill_vgpr   scratch_load / scratch_store (VGPR spills: the kernel has no other private memory)
  copies       v_mov_b64 / v_mov_b32 between registers (loop-carried value shuffles, not DPP)
  valu         every VALU instruction

Blocks inside an evaluation execute AT MOST once per evaluation (item loops: their trip counts are passed in), conditional blocks
less: the figure for an evaluation is an UPPER bound where a block is conditional.  usage:
    asm_dynamic_mix.py /tmp/asm/k.s --ls 12171 --lip 4092 --steps 4092 --machine 40 [--item-trips 3]"""
import argparse
import re
import sys


def parse(path):
    lines = open(path).read().split("\n")
    blocks, cur = [], None
    for i, l in enumerate(lines):
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+|_Z\S+):", l)
        if m:
            cur = {"name": m.group(1), "line": i + 1, "ins": [], "loops": []}
            blocks.append(cur)
            note = l
        if cur is None:
            continue
        for h, d in re.findall(r"Header=(BB\d+_\d+) Depth=(\d+)", l):
            cur["loops"].append((int(d), "." + "L" + h if not h.startswith(".L") else h))
        m2 = re.search(r"=>This (?:Inner )?Loop Header: Depth=(\d+)", l)
        if m2:
            cur["loops"].append((int(m2.group(1)), cur["name"]))
        m3 = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", l)
        if m3:
            cur["loops"].append((int(m3.group(2)), ".L" + m3.group(1)))
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", l)
        if m and not l.strip().startswith(";") and not l.strip().startswith("."):
            cur["ins"].append((m.group(1), m.group(2)))
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--ls", type=float, required=True, help="line-search evaluations per solve")
    ap.add_argument("--lip", type=float, required=True, help="Lipschitz-test evaluations per solve")
    ap.add_argument("--steps", type=float, required=True, help="PANOC steps per solve")
    ap.add_argument("--machine", type=float, default=40.0, help="evaluations of the state machine per solve")
    ap.add_argument("--item-trips", type=float, default=1.0, help="mean trips of an inner (depth >= eval depth + 1) loop per evaluation")
    args = ap.parse_args()
    blocks = parse(args.asm)
    # registers that receive SGPR spills
    spill_regs = set()
    for b in blocks:
        for op, rest in b["ins"]:
            if op.startswith("v_writelane_b32"):
                spill_regs.add(rest.split(",")[0].strip())
    # evaluation entries: the heading block (>= 30 DPP instructions and >= 100 f64)
    entries = []
    for b in blocks:
        dpp = sum(1 for op, r in b["ins"] if "dpp" in op or "dpp" in r)
        f64 = sum(1 for op, r in b["ins"] if "_f64" in op)
        if dpp >= 30 and f64 >= 100:
            entries.append(b)
    if len(entries) != 3:
        print(f"expected 3 evaluation entries, found {len(entries)}: {[e['name'] for e in entries]}", file=sys.stderr)
    # the loop header each entry's blocks are annotated with (innermost loop of the entry block)
    def inner(b):
        return max(b["loops"])[1] if b["loops"] else None
    def depth(b):
        return max(b["loops"])[0] if b["loops"] else 0
    ent = sorted(entries, key=lambda b: b["line"])
    # order in the source: state machine first, then (inside the step loop) the Lipschitz test, then the line search
    site_count = {}
    names = ["machine", "lip", "ls"]
    counts = [args.machine, args.lip, args.ls]
    for e, nm, c in zip(ent, names, counts):
        site_count[inner(e)] = (nm, c, depth(e))
    step_hdr = None
    cls = ("valu", "spill_sgpr", "spill_vgpr", "copies", "dpp", "f64", "salu", "lds", "vmem_other", "readlane_other")
    tot = {nm: dict.fromkeys(cls, 0.0) for nm in names + ["step_body", "outside"]}
    static = {nm: dict.fromkeys(cls, 0) for nm in names + ["step_body", "outside"]}
    # region of a block: the evaluation site whose header appears among its enclosing loops (deepest first); else step loop if any depth >= 2; else outside
    lip_hdr = [h for h, (nm, c, d) in site_count.items() if nm == "lip"]
    step_depth = site_count[lip_hdr[0]][2] - 1 if lip_hdr else 2
    for b in blocks:
        region, weight = "outside", 1.0
        loops = sorted(b["loops"], reverse=True)
        hit = None
        for d, h in loops:
            if h in site_count:
                hit = (d, h)
                break
        if hit:
            nm, c, d0 = site_count[hit[1]]
            region = nm
            weight = c * (args.item_trips if depth(b) > d0 else 1.0)
        elif loops and loops[0][0] >= step_depth and site_count:
            region, weight = "step_body", args.steps
        for op, rest in b["ins"]:
            k = []
            isv = op.startswith("v_")
            if isv:
                k.append("valu")
            if op.startswith("v_writelane"):
                k.append("spill_sgpr")
            elif op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
                src = rest.split(",")[1].strip() if "," in rest else ""
                k.append("spill_sgpr" if src in spill_regs else "readlane_other")
            elif op.startswith("scratch_"):
                k.append("spill_vgpr")
            elif op.startswith("v_mov_b64") or (op.startswith("v_mov_b32") and "dpp" not in rest and "dpp" not in op) or op.startswith("v_accvgpr"):
                k.append("copies")
            if "dpp" in op or "dpp" in rest or "permlane" in op:
                k.append("dpp")
            if "_f64" in op:
                k.append("f64")
            if op.startswith("s_"):
                k.append("salu")
            if op.startswith("ds_"):
                k.append("lds")
            if op.startswith(("global_", "buffer_", "flat_")):
                k.append("vmem_other")
            for c in k:
                tot[region][c] += weight
                static[region][c] += 1
    print(f"# {args.asm}: {len(blocks)} blocks; SGPR-spill registers {sorted(spill_regs)}; evaluation call sites at lines {[e['line'] for e in ent]}")
    print(f"# per-solve execution counts: state machine {args.machine:.0f}, Lipschitz test {args.lip:.0f}, line search {args.ls:.0f} evaluations; {args.steps:.0f} PANOC steps; "
          f"inner item loops x{args.item_trips:g}")
    print(f"{'region':10s} {'static VALU':>11s} {'dyn VALU/solve':>15s} {'SGPR spill':>12s} {'VGPR spill':>12s} {'copies':>10s} {'dpp':>10s} {'f64':>10s} {'rdlane(other)':>13s}")
    G = dict.fromkeys(cls, 0.0)
    for nm in names + ["step_body", "outside"]:
        t = tot[nm]
        for c in cls:
            G[c] += t[c]
        print(f"{nm:10s} {static[nm]['valu']:11d} {t['valu']:15.3e} {t['spill_sgpr']:12.3e} {t['spill_vgpr']:12.3e} {t['copies']:10.3e} {t['dpp']:10.3e} {t['f64']:10.3e} {t['readlane_other']:13.3e}")
    v = G["valu"]
    print(f"{'total':10s} {'':11s} {v:15.3e} {G['spill_sgpr']:12.3e} {G['spill_vgpr']:12.3e} {G['copies']:10.3e} {G['dpp']:10.3e} {G['f64']:10.3e} {G['readlane_other']:13.3e}")
    print(f"shares of the dynamic VALU stream (upper bounds where blocks are conditional): SGPR-spill lane moves {100 * G['spill_sgpr'] / v:.2f} %, register copies "
          f"{100 * G['copies'] / v:.2f} %, DPP {100 * G['dpp'] / v:.2f} %, f64 arithmetic {100 * G['f64'] / v:.2f} %; scratch (VGPR spill) instructions per solve "
          f"{G['spill_vgpr']:.3e} = {G['spill_vgpr'] / max(args.steps, 1):.2f} per PANOC step")


if __name__ == "__main__":
    main()

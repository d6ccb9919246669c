#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: basic blocks of one function of a `hipcc -S` listing, with the loops they belong to.

    python tools/isa_blocks.py file.s kernel_symbol_substring [--all]

Per basic block (label): instruction count, scratch loads / stores, v_readlane / v_writelane (SGPR spills), global / LDS loads, and the
innermost backward branch that spans it (a loop: `LBBx_y <- LBBx_z`).  Cross-compiles tell where; only the GPU tells how often."""
import re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    show_all = "--all" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "ins": []}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "ins": []}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        cur["ins"].append(s)
    blocks.append(cur)
    index = {b["label"]: i for i, b in enumerate(blocks)}
    loops = []                                   # (head index, tail index)
    for i, b in enumerate(blocks):
        for s in b["ins"]:
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", s)
            if m:
                t = m.group(1) or m.group(2)
                if t in index and index[t] <= i:
                    loops.append((index[t], i))
    def innermost(i):
        best = None
        for h, t in loops:
            if h <= i <= t and (best is None or (t - h) < (best[1] - best[0])):
                best = (h, t)
        return best
    def depth(i):
        return sum(1 for h, t in loops if h <= i <= t)
    tot = {"ins": 0, "ss": 0, "sl": 0}
    print(f"{'block':>12} {'ins':>5} {'s_st':>4} {'s_ld':>4} {'wrl':>3} {'rdl':>3} {'gld':>3} {'dsr':>3} {'dsw':>3} depth loop")
    for i, b in enumerate(blocks):
        c = lambda pat: sum(1 for s in b["ins"] if re.match(pat, s))
        ss, sl = c(r"scratch_store"), c(r"scratch_load")
        wl, rl = c(r"v_writelane"), c(r"v_readlane")
        gl, dr, dw = c(r"global_load"), c(r"ds_read|ds_load"), c(r"ds_write|ds_store")
        tot["ins"] += len(b["ins"]); tot["ss"] += ss; tot["sl"] += sl
        if show_all or ss or sl or wl or rl or gl:
            lp = innermost(i)
            lps = f"{blocks[lp[0]]['label']}..{blocks[lp[1]]['label']} ({lp[1] - lp[0] + 1} blocks)" if lp else "-"
            print(f"{b['label']:>12} {len(b['ins']):5d} {ss:4d} {sl:4d} {wl:3d} {rl:3d} {gl:3d} {dr:3d} {dw:3d} {depth(i):5d} {lps}")
    print("total", tot, "blocks", len(blocks), "loops", len(loops))

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static attribution of a kernel's step shell from hipcc's --save-temps assembly (round-4 verdict, item 4).

    tools/isa_attribution.py FILE.s KERNEL_SUBSTRING [--min 25]

Parses the kernel's basic blocks (labels, fall-through `%bb.N` blocks, LLVM's loop annotations), builds the control-flow
graph from the branch instructions and reports, for the OUTERMOST loop that holds most of the code (the env-step loop of
k_rollout / k_ppo_fused) and for every loop nested in it (the RK45 attempt loop, the probe ...):
  * the blocks that lie on EVERY path from the loop's header to its back edge -- they run in every iteration, whatever the
    lanes do -- with their instruction mix summed;
  * the conditional blocks (skipped when no lane of the wave needs them: `s_cbranch_execz` regions, scalar branches), listed
    one by one when they are large, so that a block full of lane moves can be told apart from one that runs every step.
Instruction classes: valu = vector ALU without the moves below; moves = v_readlane / v_writelane / v_readfirstlane /
v_accvgpr_* (SGPR spills through VGPR lanes, AGPR copies); trans = v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos;
f64 = instructions with an _f64 suffix (subset of valu + trans); salu; smem = s_load / s_buffer_load; vmem = global / buffer /
scratch; lds = ds_*.
"""
import collections
import re
import sys


def classify(ins):
    op = ins.split()[0]
    c = collections.Counter()
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr")):
        c["moves"] += 1
    elif op.startswith("v_mfma"):
        c["mfma"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op):
            c["trans"] += 1
        if "_f64" in op:
            c["f64"] += 1
    elif op.startswith(("s_load", "s_buffer_load")):
        c["smem"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        c["vmem"] += 1
        if op.startswith("scratch_"):
            c["scratch"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    return c


class Block:
    def __init__(self, name):
        self.name, self.ins, self.hint = name, [], ""
        self.header_of = None      # depth if this block is a loop header
        self.in_loop = None        # innermost loop header name (for non-headers)
        self.parent = None         # for headers: the parent loop's header
        self.succ = []

    def mix(self):
        c = collections.Counter()
        for i in self.ins:
            c += classify(i)
        c["n"] = len(self.ins)
        return c


def parse(path, kernel):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kernel in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    name = lines[start].split(":")[0]
    blocks, cur = [], Block("entry")
    blocks.append(cur)
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", t)
        m2 = re.match(r"^; %bb\.(\d+):(.*)$", t)
        if m or m2:
            cur = Block(m.group(1)[2:] if m else f"bb.{m2.group(1)}")
            blocks.append(cur)
            rest = (m.group(2) if m else m2.group(2))
            t = ";" + rest.split(";", 1)[1] if ";" in rest else ""
        if t.startswith(";"):
            mh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", t)
            if mh:
                cur.header_of = int(mh.group(1))
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", t)
            if mi:
                cur.in_loop = mi.group(1)
            mp = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", t)
            if mp and cur.parent is None:
                cur.parent = mp.group(1)   # the first Parent line is the outermost; keep overwriting below for the innermost
            if mp:
                cur.parent = mp.group(1)
            mn = re.search(r"%([\w.]+)", t)
            if mn and not cur.hint and not cur.ins:
                cur.hint = mn.group(1)[:70]
            continue
        if not t or t.startswith("."):
            continue
        cur.ins.append(t)
    by = {b.name: b for b in blocks}
    for k, b in enumerate(blocks):
        nxt = blocks[k + 1].name if k + 1 < len(blocks) else None
        fall = True
        for i in b.ins:
            op = i.split()[0]
            if op == "s_branch":
                b.succ.append(i.split()[1].lstrip(".L")); fall = False
            elif op.startswith("s_cbranch"):
                b.succ.append(i.split()[1].lstrip(".L"))
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and nxt:
            b.succ.append(nxt)
    return name, blocks, by


def loop_chain(b, by):
    """Headers of the loops a block belongs to, innermost first."""
    chain, h = [], (b.name if b.header_of else b.in_loop)
    while h and h in by and len(chain) < 8:
        chain.append(h)
        h = by[h].parent
    return chain


def every_iteration(header, members, by):
    """Members on every path header -> back edge (within the loop, inner loops collapsed as ordinary nodes)."""
    mem = set(members)
    latches = [m for m in mem if header in by[m].succ]

    def reach(skip):
        seen, st = set(), [header]
        while st:
            x = st.pop()
            if x in seen or x == skip:
                continue
            seen.add(x)
            for s in by[x].succ:
                if s in mem and s != header:
                    st.append(s)
        return seen
    must = set()
    for m in mem:
        if m == header:
            must.add(m); continue
        r = reach(m)
        if not any(l in r for l in latches if l != m):
            must.add(m)
    return must


def fmt(c):
    return " ".join(f"{k}={c[k]}" for k in ("n", "valu", "moves", "trans", "f64", "salu", "smem", "vmem", "lds", "mfma", "scratch") if c[k])


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 25
    name, blocks, by = parse(path, kernel)
    print(f"kernel {name}")
    tot = collections.Counter()
    for b in blocks:
        tot += b.mix()
    print(f"  static total: {fmt(tot)}")
    chains = {b.name: loop_chain(b, by) for b in blocks}
    headers = [b for b in blocks if b.header_of]
    size = {h.name: sum(by[x].mix()["n"] for x in chains if h.name in chains[x]) for h in headers}
    tops = [h for h in headers if h.header_of == 1]
    if not tops:
        print("  no loop"); return
    top = max(tops, key=lambda h: size[h.name])

    def report(h, indent):
        members = [x for x in chains if h.name in chains[x]]
        direct = [x for x in members if chains[x][0] == h.name]          # not inside a nested loop
        inner = [x for x in headers if x.parent == h.name and x.name in members]
        # collapse: for the must-analysis every member counts as a node (inner-loop blocks included)
        must = every_iteration(h.name, members, by)
        ev, cond = collections.Counter(), collections.Counter()
        for x in direct:
            (ev if x in must else cond).update(by[x].mix())
        pad = " " * indent
        print(f"{pad}loop {h.name} (depth {h.header_of}, {size[h.name]} instructions, {len(members)} blocks)")
        print(f"{pad}  every iteration (outside nested loops): {fmt(ev)}")
        print(f"{pad}  conditional     (outside nested loops): {fmt(cond)}")
        big = sorted([x for x in direct if by[x].mix()["n"] >= minsz], key=lambda x: -by[x].mix()["n"])
        for x in big:
            tag = "EVERY" if x in must else "cond "
            print(f"{pad}    {tag} {x:>10s} {fmt(by[x].mix())}  {by[x].hint}")
        for ih in inner:
            report(ih, indent + 4)
    report(top, 2)
    # code outside the top loop
    out = collections.Counter()
    for b in blocks:
        if top.name not in chains[b.name]:
            out += b.mix()
    print(f"  outside the step loop: {fmt(out)}")


if __name__ == "__main__":
    main()

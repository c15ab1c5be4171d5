#!/usr/bin/env python3
"""Static instruction mix of the main loop of a kernel in a --save-temps .s file (development aid).

    python tools/isa_loop_count.py FILE.s KERNEL_SUBSTRING

Finds the kernel whose (mangled) name contains KERNEL_SUBSTRING, takes the depth-1 loop that holds the tally's ds_add_f32 and counts VALU / SALU /
LDS / VMEM instructions of the basic blocks that belong to it, split at the first v_fract_f32 (swap arm | step arm)."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or (i > start and re.match(r"^_Z\w*:", lines[i])))
    body = lines[start:end]
    # blocks: label line or '; %bb.N:' comment starts a block; membership from the 'in Loop: Header=X' / 'Parent Loop X' / 'This Loop Header' notes
    blocks, cur = [], None
    for l in body:
        if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", l):
            cur = {"head": l, "notes": l, "ins": []}
            blocks.append(cur)
        elif cur is not None:
            if l.strip().startswith(";"):
                cur["notes"] += " " + l
            elif l.strip() and not l.strip().startswith("."):
                cur["ins"].append(l.strip().split()[0])
    loops = {}
    for b in blocks:
        m = re.search(r"=>\s*This Loop Header: Depth=1", b["notes"])
        if m:
            name = re.match(r"^\.(LBB\w+):", b["head"]).group(1)
            loops[name] = [b]
    for b in blocks:
        for name in loops:
            if re.search(r"(Header=|Parent Loop )" + name.replace("LBB", "BB") + r"\b", b["notes"]):
                loops[name].append(b)
    # the walk: the loop that holds the LDS float atomic of the tally
    cands = [n for n in loops if any(i.startswith("ds_add_f32") for b in loops[n] for i in b["ins"])] or list(loops)
    name = max(cands, key=lambda n: sum(len(b["ins"]) for b in loops[n]))
    order = [b for b in blocks if b in loops[name]]
    seen_fract = False
    part = {"swap": [], "step": []}
    for b in order:
        if any(i.startswith("v_fract") for i in b["ins"]) or any(i.startswith("v_trunc") for i in b["ins"]):
            seen_fract = True
        part["step" if seen_fract else "swap"] += b["ins"]
    for k, ins in part.items():
        c = lambda p: sum(1 for i in ins if re.match(p, i))
        print("%-5s VALU %4d  SALU %4d  LDS %3d  VMEM %3d   (v_mov %d, v_cndmask %d, v_cmp %d)" % (
            k, c(r"v_"), c(r"s_"), c(r"ds_"), c(r"(global|flat|buffer)_"), c(r"v_mov"), c(r"v_cndmask"), c(r"v_cmp")))


if __name__ == "__main__":
    main()

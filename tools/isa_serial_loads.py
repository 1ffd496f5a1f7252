#!/usr/bin/env python3
"""Find chains of dependent-looking loads in a gfx950 assembly listing: runs of N or more loads (LDS, global, scratch, scalar) each
followed within a few instructions by a full wait (`s_waitcnt lgkmcnt(0)` / `vmcnt(0)`) before the next load is issued. That is what hipcc
emits when it reuses one register for every load of an unrolled loop (seen at the register ceiling of a kernel, or through `volatile` LDS
pointers): N memory round trips instead of one. Round 3 found the 32-long chain in the pose optimiser's block reduction this way
(8 960 cycles per reduction); no other kernel of the library has a chain of four or more.

usage:  hipcc --offload-arch=gfx950 -O3 ... -c kernel.hip -o /tmp/k.o -save-temps=obj
        tools/isa_serial_loads.py /tmp/k-hip-amdgcn-amd-amdhsa-gfx950.s [min_chain=4]
prints, per kernel, (chain length, first line of the chain in the listing)."""
import re
import sys

LOADS = ("ds_read", "global_load", "scratch_load", "s_load", "buffer_load", "flat_load")


def chains(ins, min_chain):
    out, cur, k = [], [], 0
    while k < len(ins) - 1:
        line_no, text = ins[k]
        if text.startswith(LOADS):
            waited, j = False, k + 1
            while j < len(ins) and j <= k + 4:
                nxt = ins[j][1]
                if nxt.startswith("s_waitcnt") and ("lgkmcnt(0)" in nxt or "vmcnt(0)" in nxt):
                    waited = True
                    break
                if nxt.startswith(LOADS):
                    break
                j += 1
            if waited:
                cur.append(line_no)
            else:
                if len(cur) >= min_chain:
                    out.append(cur)
                cur = []
        elif text.startswith(("s_barrier", "s_cbranch", "s_branch", "s_endpgm")):
            if len(cur) >= min_chain:
                out.append(cur)
            cur = []
        k += 1
    if len(cur) >= min_chain:
        out.append(cur)
    return out


def main():
    path = sys.argv[1]
    min_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    name, ins, found = None, [], 0
    for i, raw in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            if name:
                for c in chains(ins, min_chain):
                    print("%s: %d loads, each waited for, from line %d" % (name[:100], len(c), c[0]))
                    found += 1
            name, ins = m.group(1), []
            continue
        t = raw.strip()
        if t and not t.startswith(";") and not t.endswith(":") and not t.startswith("."):
            ins.append((i, t))
    if name:
        for c in chains(ins, min_chain):
            print("%s: %d loads, each waited for, from line %d" % (name[:100], len(c), c[0]))
            found += 1
    print("# %d chain(s) of >= %d serialised loads" % (found, min_chain))


if __name__ == "__main__":
    main()

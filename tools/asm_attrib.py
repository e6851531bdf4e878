#!/usr/bin/env python3
"""Static instruction counts of one kernel, attributed to source lines.

usage: asm_attrib.py file.s kernel_mangled_name [--by func|line]

file.s = `hipcc --cuda-device-only -gline-tables-only -S` output.  Every instruction is charged to
the innermost .loc in force (file, line); lines are then grouped into the source regions listed in
REGIONS (rs_physics_body.inc / rs_math.hpp / rs_kernels.hip line ranges) so that one can see where
the vector, scalar, memory and branch instructions of the time loop come from.  Static counts: a
block that runs once per step counts once, the boundary-layer loop body counts once too.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
        return "branch"
    if op in ("s_waitcnt", "s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    by = "line" if "--by" in sys.argv and sys.argv[sys.argv.index("--by") + 1] == "line" else "file"
    files = {}
    cur = (0, 0)
    inside = False
    counts = collections.defaultdict(lambda: collections.Counter())
    for ln in open(path):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", ln)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if ln.startswith(kernel + ":"):
            inside = True
            continue
        if not inside:
            continue
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+([a-z_0-9]+)", ln)
        if not m or ln.lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        cls = classify(op)
        if cls == "other":
            continue
        key = (files.get(cur[0], "?"), cur[1]) if by == "line" else files.get(cur[0], "?")
        counts[key][cls] += 1
    tot = collections.Counter()
    keys = sorted(counts, key=lambda k: -sum(counts[k].values()))
    print(f"{'where':50s} {'valu':>6s} {'salu':>6s} {'smem':>6s} {'branch':>6s} {'wait':>6s} {'lds':>5s} {'vmem':>5s}")
    for k in keys:
        c = counts[k]
        tot.update(c)
        name = f"{k[0]}:{k[1]}" if by == "line" else k
        print(f"{name:50s} {c['valu']:6d} {c['salu']:6d} {c['smem']:6d} {c['branch']:6d} {c['wait']:6d} {c['lds']:5d} {c['vmem']:5d}")
    print(f"{'TOTAL':50s} {tot['valu']:6d} {tot['salu']:6d} {tot['smem']:6d} {tot['branch']:6d} {tot['wait']:6d} {tot['lds']:5d} {tot['vmem']:5d}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing: per basic block, the number of vector-ALU,
transcendental, matrix, scalar, LDS and vector-memory instructions, with the back edges (loops) marked.

    hipcc --offload-arch=gfx950 -O3 ... --offload-device-only -S ac_fast.hip -o /tmp/ac_fast.s
    python tools/asm_blocks.py /tmp/ac_fast.s 'k_fwd_multiILi2ELi0ELi4ELi0ELb0ELb1E'

Used to see where a kernel's issue slots go before spending GPU time on counters (profiles/r4/*census*.txt).
"""
import re
import sys
from collections import OrderedDict

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]):
            if re.match(r"^_Z\S+:", l):
                start = i
                break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = {"n": OrderedDict(), "succ": [], "line": start}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = {"n": OrderedDict(), "succ": [], "line": i}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        b = blocks[cur]
        b["n"][c] = b["n"].get(c, 0) + 1
        if c == "branch":
            t = s.split()[-1]
            b["succ"].append(t)
        if op.startswith("scratch_"):
            b["n"]["scratch"] = b["n"].get("scratch", 0) + 1
    order = list(blocks)
    pos = {k: i for i, k in enumerate(order)}
    cats = ["valu", "trans", "mfma", "salu", "lds", "vmem", "smem", "wait", "branch", "scratch"]
    print("%-12s %6s " % ("block", "line") + " ".join("%6s" % c for c in cats) + "  back-edges")
    tot = {c: 0 for c in cats}
    for k in order:
        b = blocks[k]
        back = [t for t in b["succ"] if t in pos and pos[t] <= pos[k]]
        n = sum(b["n"].values())
        if n == 0:
            continue
        for c in cats:
            tot[c] += b["n"].get(c, 0)
        print("%-12s %6d " % (k, b["line"] + 1) + " ".join("%6d" % b["n"].get(c, 0) for c in cats) + ("  -> " + ",".join(back) if back else ""))
    print("%-12s %6s " % ("total", "") + " ".join("%6d" % tot[c] for c in cats))


if __name__ == "__main__":
    main()

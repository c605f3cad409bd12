#!/usr/bin/env python3
"""Static instruction mix of one kernel of the gfx950 assembly, split at its workgroup barriers (build host, no GPU needed).

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only csrc/hsqp_capi.hip -o capi.s
    python tools/isa_phases.py capi.s 'k_lqILb1' [--all]

The LQ kernels are issue-bound sequences of barrier-separated phases whose item loops are mostly unrolled, so the static count of a
segment is close to what one wave issues per pass through it; segments that contain a backward branch (a loop) are flagged with the
number of instructions inside the loop body.  Used to pick what to cut before spending GPU time."""
import re
import sys


def kernel_lines(path, pat):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\S*" + pat + r"\S*:", l):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            return lines[start:i]
    raise SystemExit("kernel not found")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_") and "f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    body = kernel_lines(path, pat)
    segs, cur, labels = [], {"n": 0, "loops": []}, {}
    pos = 0
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\S+):", t)
            if m:
                labels[m.group(1)] = pos
            continue
        if re.match(r"^\S+:", t):
            continue
        op = t.split()[0]
        pos += 1
        if op == "s_barrier":
            segs.append(cur)
            cur = {"n": 0, "loops": []}
            continue
        k = classify(op)
        cur[k] = cur.get(k, 0) + 1
        cur["n"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels:   # backward branch: a loop of (pos - labels[tgt]) instructions
                cur["loops"].append(pos - labels[tgt])
    segs.append(cur)
    keys = ["f64", "valu", "mfma", "salu", "smem", "lds", "vmem", "scratch"]
    tot = {k: 0 for k in keys + ["n"]}
    print(f"{'seg':>4} {'instr':>6} " + " ".join(f"{k:>7}" for k in keys) + "  loops(body sizes)")
    for i, s in enumerate(segs):
        for k in keys + ["n"]:
            tot[k] += s.get(k, 0)
        print(f"{i:4d} {s['n']:6d} " + " ".join(f"{s.get(k, 0):7d}" for k in keys) + ("  " + str(s["loops"]) if s["loops"] else ""))
    print(f"{'sum':>4} {tot['n']:6d} " + " ".join(f"{tot[k]:7d}" for k in keys))


if __name__ == "__main__":
    main()

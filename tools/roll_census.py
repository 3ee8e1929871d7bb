#!/usr/bin/env python3
"""Static instruction census of creff_roll_kernel's five role loops (consumer kh 0 / kh 1, producers KV / Q / AUX) from a hipcc -S listing:
for every depth-2 loop (the iteration loop of a role) the instructions between consecutive s_barrier, by class.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only ar-seg_amd/csrc/creff_roll.hip -o /tmp/creff_roll.s
    python tools/roll_census.py /tmp/creff_roll.s creff_roll_kernelILi1E

With --marks (listing built with -DROLL_MARK): the instructions between the "; ROLLMARK i" comments the RT(i) sites leave -- every role,
also where hipcc peeled or unswitched the iteration loop (the LAST copy of a mark sequence 3 -> 0 -> 1 -> 2 -> 3 of a role is its steady-state
loop body): segment "3->0" = H1, "1->2" = H2.
"""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_census import classify  # noqa: E402


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start:end]
    if "--marks" in sys.argv:
        marks = [(i, int(l.split("ROLLMARK")[1].split()[0])) for i, l in enumerate(body) if "ROLLMARK" in l]
        for (i0, m0), (i1, m1) in zip(marks, marks[1:]):
            if (m0, m1) not in ((3, 0), (1, 2)):
                continue
            sg = collections.Counter()
            for l in body[i0:i1]:
                t = l.strip()
                if not t or t.startswith((";", ".")) or t.endswith(":"):
                    continue
                cls, cost = classify(t.split()[0], t)
                sg[cls] += 1
                sg["_cycles"] += cost
            valu = sum(v for c, v in sg.items() if c.startswith("valu"))
            print(f"line {start + i0:6d} {'H1' if m0 == 3 else 'H2'}: valu {valu:4d}  mfma {sg['mfma']:3d}  ds_read {sg['ds_read']:3d}  ds_write {sg['ds_write']:3d}  "
                  f"vmem {sg['vmem_load'] + sg['vmem_store']:3d}  salu {sg['salu']:3d}  issue~{sg['_cycles']:.0f}   " +
                  " ".join(f"{c[5:]}={v}" for c, v in sorted(sg.items()) if c.startswith("valu_")))
        return
    # loop headers at depth 2
    heads = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l]
    for h in heads:
        # label is on the previous line(s)
        j = h
        while not body[j].startswith(".LBB"):
            j -= 1
        label = body[j].split(":")[0]
        # the loop ends at the last branch back to the label
        last = max(i for i, l in enumerate(body) if re.search(r"s_cbranch\w*\s+" + re.escape(label) + r"\b|s_branch\s+" + re.escape(label) + r"\b", l))
        segs, cur = [], collections.Counter()
        for l in body[j:last + 1]:
            t = l.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            op = t.split()[0]
            if op == "s_barrier":
                segs.append(cur)
                cur = collections.Counter()
                continue
            cls, cost = classify(op, t)
            cur[cls] += 1
            cur["_cycles"] += cost
        segs.append(cur)
        print(f"loop {label}: {len(segs) - 1} barriers")
        for k, sg in enumerate(segs):
            valu = sum(v for c, v in sg.items() if c.startswith("valu"))
            print(f"   seg {k}: valu {valu:4d}  mfma {sg['mfma']:3d}  ds_read {sg['ds_read']:3d}  ds_write {sg['ds_write']:3d}  vmem {sg['vmem_load'] + sg['vmem_store']:3d}  "
                  f"salu {sg['salu']:3d}  issue~{sg['_cycles']:.0f}   " + " ".join(f"{c[5:]}={v}" for c, v in sorted(sg.items()) if c.startswith("valu_")))


if __name__ == "__main__":
    main()

"""Generates tests/golden/heap_ref.json by driving the REFERENCE's intrusive heap
(smpl/include/smpl/intrusive_heap.h, compiled in place into oracle/_ref/heap_ref by oracle/Makefile)
with seeded operation sequences.  Run in the build container (where /root/reference exists):

    make -C oracle ref && python tests/golden/make_heap_golden.py

The fixture holds inputs (ops) and the reference's outputs (element on top after every op)."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "oracle", "_ref", "heap_ref")


def make_ops(seed, n, nprio):
    rng = np.random.default_rng(seed)
    ops = []
    prios = []          # current priority per element
    alive = []
    for _ in range(n):
        r = rng.random()
        if r < 0.45 or not alive:
            p = int(rng.integers(0, nprio))
            ops.append((0, p)); prios.append(p); alive.append(len(prios) - 1)
        elif r < 0.70:
            ops.append((1, 0))      # pop: which element leaves is the heap's business; track lazily below
        elif r < 0.82:
            e = int(rng.choice(alive)); p = int(rng.integers(0, prios[e] + 1))
            ops.append((2, (e << 20) | p)); prios[e] = p
        elif r < 0.90:
            e = int(rng.choice(alive)); p = int(rng.integers(prios[e], max(nprio, prios[e] + 1)))
            ops.append((5, (e << 20) | p)); prios[e] = p
        elif r < 0.97:
            e = int(rng.choice(alive)); ops.append((3, e))
        else:
            ops.append((4, 0)); prios = [(p * 7919 + 13) % 1000 for p in prios]
    return ops


def run_ref(ops):
    text = f"{len(ops)}\n" + "\n".join(f"{c} {k}" for c, k in ops) + "\n"
    out = subprocess.run([BIN], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split()
    return [int(x) for x in out]


def main():
    cases = []
    # ties are the interesting part: few distinct priorities
    for seed, n, nprio in [(1, 400, 4), (2, 400, 1000), (3, 2000, 16), (4, 2000, 3), (5, 3000, 1000)]:
        ops = make_ops(seed, n, nprio)
        cases.append({"seed": seed, "ops": ops, "top_after_each_op": run_ref(ops)})
    # the sequence SURVEY.md section 8c quotes: {5,3,3,9,1} pops element indices 4,1,2,0,3
    ops = [(0, 5), (0, 3), (0, 3), (0, 9), (0, 1)] + [(1, 0)] * 5
    cases.append({"seed": "survey", "ops": ops, "top_after_each_op": run_ref(ops)})
    with open(os.path.join(ROOT, "tests", "golden", "heap_ref.json"), "w") as f:
        json.dump({"source": "smpl/include/smpl/intrusive_heap.h via oracle/_ref/heap_ref", "cases": cases}, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()

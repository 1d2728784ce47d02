"""Generates tests/golden/shortcut_ref.json by driving the REFERENCE's path shortcutting loop
(smpl/include/smpl/geometry/shortcut.h + detail/shortcut.hpp, compiled in place into oracle/_ref/shortcut_ref by
oracle/Makefile) with seeded tables.  Run in the build container (where /root/reference exists):

    make -C oracle ref && python tests/golden/make_shortcut_golden.py

The fixture holds inputs (P, the cost table, the validity table) and the reference's outputs (indices of the shortcut
path).  Costs are dyadic rationals so that the text round trip is exact."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "oracle", "_ref", "shortcut_ref")


def make_case(seed):
    rng = np.random.default_rng(seed)
    P = int(rng.integers(0, 29)) if seed % 7 else int(seed % 3)          # includes P = 0, 1, 2
    kind = seed % 4
    if kind == 0:      # metric costs: L1 distance between random lattice points -> shortcuts never cost more
        pts = rng.integers(-64, 64, size=(max(P, 1), 3)) / 16.0
        cost = np.abs(pts[:P, None, :] - pts[None, :P, :]).sum(-1)
    elif kind == 1:    # points on a line with equal steps: direct cost == accumulated cost exactly (the "<=" case)
        x = np.arange(max(P, 1)) / 8.0
        cost = np.abs(x[:P, None] - x[None, :P])
    elif kind == 2:    # arbitrary costs: a shortcut may be valid and still lose on cost
        cost = rng.integers(1, 64, size=(P, P)) / 8.0
    else:              # metric costs with a detour penalty on long hops
        pts = rng.integers(-64, 64, size=(max(P, 1), 2)) / 16.0
        cost = np.abs(pts[:P, None, :] - pts[None, :P, :]).sum(-1)
        hop = np.abs(np.arange(P)[:, None] - np.arange(P)[None, :])
        cost = cost + (hop > 3) * 0.5
    pv = [0.95, 0.7, 0.4, 0.15][(seed // 4) % 4]
    valid = (rng.random((P, P)) < pv).astype(int)
    return P, np.asarray(cost, float).reshape(P, P), valid


def run(P, cost, valid):
    text = f"{P}\n" + " ".join(repr(float(c)) for c in cost.ravel()) + "\n" + " ".join(str(int(v)) for v in valid.ravel()) + "\n"
    out = subprocess.run([BIN], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split()
    ok, n = int(out[0]), int(out[1])
    return ok, [int(x) for x in out[2:2 + n]]


def main():
    cases = []
    for seed in range(1, 65):
        P, cost, valid = make_case(seed)
        ok, idx = run(P, cost, valid)
        cases.append({"seed": seed, "P": P, "cost": cost.ravel().tolist(), "valid": valid.ravel().tolist(), "ok": ok, "out": idx})
    with open(os.path.join(ROOT, "tests", "golden", "shortcut_ref.json"), "w") as f:
        json.dump({"source": "smpl/include/smpl/geometry/detail/shortcut.hpp:110-286 compiled in place (oracle/_ref/shortcut_ref)",
                   "cases": cases}, f)
    print(len(cases), "cases,", sum(len(c["out"]) for c in cases), "output points")


if __name__ == "__main__":
    main()

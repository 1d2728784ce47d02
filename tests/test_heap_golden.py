"""The oracle's IntrusiveHeap against golden vectors produced by the REFERENCE's own header
(smpl/include/smpl/intrusive_heap.h compiled in place: oracle/_ref/heap_ref, tests/golden/make_heap_golden.py).
This is the one piece of the path the reference's code pins directly."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle_binding import ORACLE_DIR, Oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "heap_ref.json")


@pytest.fixture(scope="module")
def oracle(small_cfg):
    return Oracle(small_cfg)


def test_heap_matches_reference_golden(oracle):
    cases = json.load(open(GOLD))["cases"]
    assert len(cases) >= 6
    for c in cases:
        got = oracle.heap_run(np.array(c["ops"], np.int32))
        assert got.tolist() == c["top_after_each_op"], f"case seed={c['seed']}"


def test_survey_tie_order(oracle):
    # priorities {5,3,3,9,1} pop in element order 4,1,2,0,3 (ties: the newer element sifts above)
    ops = [(0, 5), (0, 3), (0, 3), (0, 9), (0, 1)] + [(1, 0)] * 5
    got = oracle.heap_run(np.array(ops, np.int32)).tolist()
    assert got[4:] == [4, 1, 2, 0, 3, -1]


def test_live_reference_binary_agrees_when_present(oracle):
    """In the build container oracle/_ref/heap_ref exists: drive it live on a fresh random sequence."""
    binp = os.path.join(ORACLE_DIR, "_ref", "heap_ref")
    if not os.path.exists(binp):
        pytest.skip("oracle/_ref/heap_ref not built here (reference absent)")
    rng = np.random.default_rng(77)
    ops = []
    n = 0
    for _ in range(1500):
        r = rng.random()
        if r < 0.5 or n == 0:
            ops.append((0, int(rng.integers(0, 6)))); n += 1
        elif r < 0.85:
            ops.append((1, 0))
        else:
            ops.append((3, int(rng.integers(0, n))))
    text = f"{len(ops)}\n" + "\n".join(f"{c} {k}" for c, k in ops) + "\n"
    ref = [int(x) for x in subprocess.run([binp], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.split()]
    assert oracle.heap_run(np.array(ops, np.int32)).tolist() == ref

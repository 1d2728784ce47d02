"""Readers of the reference's on-disk formats (host side)."""
import os

import numpy as np
import pytest

from oracle_binding import Oracle
from smpl_amd import formats, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_YAML = "/root/reference/sbpl_collision_checking_test/config/collision_model_pr2.yaml"


def test_env_file_of_the_reference_scene():
    boxes = formats.parse_env(open(os.path.join(GOLDEN, "tabletop.env")).read())
    # the header says ONE object: the second row of the file is never read (call_planner.cpp:183-204)
    assert boxes == [("tabletop", (0.55, 0.0, 0.6), (0.4, 1.5, 0.02))]
    assert (boxes[0][1], boxes[0][2]) == scenes.TABLETOP
    assert formats.parse_env("") == [] and formats.parse_env("2\na 1 2 3 4 5 6\nb 1 2") == [
        ("a", (1.0, 2.0, 3.0), (4.0, 5.0, 6.0)), ("b", (1.0, 2.0, 0.0), (0.0, 0.0, 0.0))]


def test_reference_mprim_file_loads_as_upstream_rows(small_cfg):
    import copy
    text = open(os.path.join(GOLDEN, "pr2.mprim")).read()
    cfg = copy.copy(small_cfg)
    cfg.mprim = text
    o = Oracle(cfg)
    assert o.M == 25                       # 3 adaptive slots + (4 long + 7 short) x 2
    o2 = Oracle(small_cfg)                 # the generated text of the test scenes is the same table
    o.set_goal_joint(cfg.goal, cfg.goal_tol); o2.set_goal_joint(cfg.goal, cfg.goal_tol)
    a, b = o.eval_state(np.array(cfg.start)), o2.eval_state(np.array(cfg.start))
    assert np.array_equal(a["flags"], b["flags"]) and np.array_equal(a["q"], b["q"])


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="the reference tree is only present in the build container")
def test_sphere_lines_from_the_reference_collision_yaml():
    lines = formats.sphere_lines_from_collision_yaml(open(REF_YAML).read(), links=["r_upper_arm_roll_link", "r_shoulder_pan_link"],
                                                     rename={"r_upper_arm_roll_link": "upper_arm_link"})
    assert lines[0] == "sphere r_shoulder_pan_link rsh0 0.13 0.0 -0.04 0.16 5"
    assert len(lines) == 5 and lines[1].startswith("sphere upper_arm_link rua0 0.18 0.0 -0.015 0.11 4")
    everything = formats.sphere_lines_from_collision_yaml(open(REF_YAML).read())
    assert len(everything) > 60

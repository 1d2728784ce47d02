"""Readers of the reference's on-disk formats (host side)."""
import os

import numpy as np
import pytest

from oracle_binding import Oracle
from smpl_amd import formats, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_YAML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collision_model_pr2.yaml")   # fixture copy of the reference's data file


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


def test_sphere_lines_from_the_reference_collision_yaml():
    lines = formats.sphere_lines_from_collision_yaml(open(REF_YAML).read(), links=["r_upper_arm_roll_link", "r_shoulder_pan_link"],
                                                     rename={"r_upper_arm_roll_link": "upper_arm_link"})
    assert lines[0] == "sphere r_shoulder_pan_link rsh0 0.13 0.0 -0.04 0.16 5"
    assert len(lines) == 5 and lines[1].startswith("sphere upper_arm_link rua0 0.18 0.0 -0.015 0.11 4")
    everything = formats.sphere_lines_from_collision_yaml(open(REF_YAML).read())
    assert len(everything) > 60


ARM_URDF = """<?xml version="1.0"?>
<robot name="arm3">
  <link name="base_link"/><link name="l1"/><link name="l2"/><link name="l3"/><link name="tool"/>
  <link name="head"/><link name="cam"/>
  <joint name="j1" type="revolute"><parent link="base_link"/><child link="l1"/>
    <origin xyz="0 0 0.3" rpy="0 0 0.5"/><axis xyz="0 0 1"/><limit lower="-1.5" upper="1.5" effort="1" velocity="1"/></joint>
  <joint name="j2" type="continuous"><parent link="l1"/><child link="l2"/><origin xyz="0.25 0 0"/><axis xyz="0 1 0"/></joint>
  <joint name="j3" type="prismatic"><parent link="l2"/><child link="l3"/><origin xyz="0.2 0 0" rpy="0 1.5707963267948966 0"/>
    <axis xyz="0 0 1"/><limit lower="0" upper="0.2"/></joint>
  <joint name="jt" type="fixed"><parent link="l3"/><child link="tool"/><origin xyz="0 0 0.1"/></joint>
  <joint name="neck" type="revolute"><parent link="base_link"/><child link="head"/><origin xyz="0 0 1"/><axis xyz="0 0 1"/>
    <limit lower="-1" upper="1"/></joint>
  <joint name="camj" type="fixed"><parent link="head"/><child link="cam"/></joint>
</robot>
"""


def test_urdf_subset_reader_produces_a_model_the_host_compiler_accepts():
    from smpl_amd import capi
    spheres = ["sphere l1 a 0.1 0 0 0.08 2", "sphere l2 b 0.1 0 0 0.06 2", "sphere l3 c 0 0 0.05 0.05 1", "sphere head h 0 0 0 0.1 1"]
    text = formats.urdf_to_robot_text(ARM_URDF, "arm", ["l1", "l2", "l3"], ["j1", "j2", "j3"], "tool", spheres, [("l1", "l3")])
    lines = text.splitlines()
    assert lines[0] == "robot arm3" and lines[1] == "link base_link"
    assert "link head" not in lines and not any(l.startswith("joint neck") for l in lines)   # off the group's sub-tree
    assert any(l.startswith("joint j1 revolute base_link l1  0.0 0.0 0.3  0.0 0.0 0.5  0.0 0.0 1.0  -1.5 1.5") for l in lines)
    assert any(l.startswith("joint j2 continuous l1 l2") for l in lines)
    assert not any("sphere head" in l for l in lines) and "acm l1 l3" in lines
    m = capi.Model(text)                                  # the host model compiler (no GPU needed)
    assert (m.nvars, m.ntrees) == (3, 3)
    # a non-planning movable joint on the kept chain is frozen at zero; a non-zero hold must be folded in by the caller
    t2 = formats.urdf_to_robot_text(ARM_URDF, "arm", ["l1", "l2", "l3"], ["j1", "j3"], "tool", spheres)
    assert any(l.startswith("joint j2 fixed l1 l2") for l in t2.splitlines())
    with pytest.raises(ValueError):
        formats.urdf_to_robot_text(ARM_URDF, "arm", ["l1", "l2", "l3"], ["j1", "j3"], "tool", spheres, held_at={"j2": 0.3})
    with pytest.raises(ValueError):
        formats.urdf_to_robot_text(ARM_URDF.replace('type="fixed"><parent link="head"', 'type="floating"><parent link="head"'),
                                   "arm", ["l1"], ["j1"], "l1")


def test_group_links_from_the_reference_collision_yaml():
    links = formats.group_links_from_collision_yaml(open(REF_YAML).read(), "right_gripper")
    assert links == ["r_gripper_palm_link", "r_gripper_r_finger_link", "r_gripper_r_finger_tip_link", "r_gripper_l_finger_link",
                     "r_gripper_l_finger_tip_link"]
    assert formats.group_links_from_collision_yaml(open(REF_YAML).read(), "head") == ["head_pan_link", "head_tilt_link"]
    with pytest.raises(ValueError):          # right_arm is a chain r_shoulder_pan_link -> r_wrist_roll_link: needs the URDF
        formats.group_links_from_collision_yaml(open(REF_YAML).read(), "right_arm")


def test_group_chains_resolve_through_the_urdf():
    y = """
robot_collision_model:
  collision_groups:
    - name: tip_group
      links:
        - name: tool
    - name: arm
      groups: [ tip_group ]
      chains:
        - base: l1
          tip: l3
"""
    assert formats.group_links_from_collision_yaml(y, "arm", ARM_URDF) == ["tool", "l1", "l2", "l3"]
    with pytest.raises(ValueError):
        formats.group_links_from_collision_yaml(y.replace("base: l1", "base: head"), "arm", ARM_URDF)


def test_pr2_right_arm_as_data_compiles_to_the_surveyed_model(cfg3_pr2):
    """cfg 3's robot from data files: 7 planning variables, 8 sphere trees with 1, 4, 7, 3, 2, 2, 2, 2 leaves (23 leaves:
    SURVEY section 2, K2 row), the limits of the URDF subset, and only the five link pairs the demo's allowed-collision
    matrix leaves checked (r_shoulder_pan_link against the wrist-roll link, the fingers and the finger tips)."""
    from smpl_amd import capi
    text = cfg3_pr2.robot_text
    lines = text.splitlines()
    spheres = [l.split() for l in lines if l.startswith("sphere ")]
    per_link = {}
    for sp in spheres:
        per_link[sp[1]] = per_link.get(sp[1], 0) + 1
    assert len(spheres) == 23 and sorted(per_link.values()) == [1, 2, 2, 2, 2, 3, 4, 7]
    assert per_link["r_forearm_roll_link"] == 7 and per_link["r_upper_arm_roll_link"] == 4 and per_link["r_shoulder_pan_link"] == 1
    assert any(l.startswith("joint r_elbow_flex_joint revolute r_upper_arm_link r_elbow_flex_link  0.4 0.0 0.0") and l.endswith("-2.1213 -0.15")
               for l in lines)
    assert any(l.startswith("joint r_forearm_roll_joint continuous") for l in lines)
    assert sum(l.startswith("acm ") for l in lines) == 83
    m = capi.Model(text)
    assert (m.nvars, m.ntrees) == (7, 8)
    assert m.npairs == 5


def test_pr2_right_arm_kinematics_known_answers(cfg3_pr2):
    """Known answers for the URDF subset (hand arithmetic on the joint origins): with every joint at zero the arm points
    along +x from the shoulder, so r_gripper_palm_link sits at torso + (0, -0.188, 0) + (0.1 + 0.4 + 0.321, 0, 0); with the
    shoulder pan at +90 degrees the same chain points along +y."""
    from oracle_binding import Oracle
    o = Oracle(cfg3_pr2)
    p0 = np.array(o.planning_fk([0.0] * 7))
    assert np.allclose(p0, [0.821, -0.188, 0.8], atol=1e-12)
    p1 = np.array(o.planning_fk([np.pi / 2, 0, 0, 0, 0, 0, 0]))
    assert np.allclose(p1, [0.0, -0.188 + 0.821, 0.8], atol=1e-12)
    # elbow bent by -90 degrees about y: the forearm points along +z from the elbow at x = 0.5
    p2 = np.array(o.planning_fk([0, 0, 0, -np.pi / 2, 0, 0, 0]))
    assert np.allclose(p2, [0.5, -0.188, 0.8 + 0.321], atol=1e-12)

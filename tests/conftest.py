import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_cfg():
    from smpl_amd import scenes
    return scenes.config_small()


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="session")
def cfg3_pr2():
    """SURVEY cfg 3 with the PR2 right arm built from data files: the reference's collision-model YAML (fixture copy), a
    hand-transcribed URDF subset of the arm, and the right-arm rows of the demo's allowed-collision matrix."""
    import json
    from smpl_amd import scenes
    y = open(os.path.join(GOLDEN, "collision_model_pr2.yaml")).read()
    u = open(os.path.join(GOLDEN, "pr2_right_arm.urdf")).read()
    acm = json.load(open(os.path.join(GOLDEN, "pr2_right_arm_acm.json")))["allowed_pairs"]
    return scenes.config3_pr2(y, u, acm)

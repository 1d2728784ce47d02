"""Writes tests/golden/pr2_right_arm_acm.json: the pairs of `right_arm` group links that the reference demo's allowed-
collision matrix marks as never colliding (the 1 081 rows of smpl_test/src/call_planner.cpp:441-1526, "copied from the
srdf for the pr2"), restricted to the 14 links of the group.  Run in the build container (needs /root/reference); the
JSON is data -- link-name pairs -- and travels with the tests."""
import itertools
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smpl_amd import formats   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rows = re.findall(r'acm\.setEntry\("([^"]+)", "([^"]+)", true\)', open("/root/reference/smpl_test/src/call_planner.cpp").read())
allowed = {frozenset(r) for r in rows}
links = formats.group_links_from_collision_yaml(open(os.path.join(HERE, "collision_model_pr2.yaml")).read(), "right_arm",
                                                open(os.path.join(HERE, "pr2_right_arm.urdf")).read())
pairs = [[a, b] for a, b in itertools.combinations(links, 2) if frozenset((a, b)) in allowed]
json.dump({"group": "right_arm", "links": links, "allowed_pairs": pairs, "rows_in_reference_matrix": len(rows)},
          open(os.path.join(HERE, "pr2_right_arm_acm.json"), "w"), indent=0)
print(len(pairs), "allowed pairs among", len(links), "links")

"""Readers for the on-disk formats either side of the path (host side only).

* `.env` object files as `smpl_test/src/call_planner.cpp:158-207` reads them: a count, then `count` rows of
  `name x y z dim_x dim_y dim_z` (whitespace separated); anything after the counted rows is ignored.
* the `spheres_models` section of an `sbpl_collision_checking` collision-model YAML
  (`sbpl_collision_checking/src/collision_model_config.cpp`: per link a list of `{name, x, y, z, radius, priority}`),
  turned into the `sphere` lines of this repo's plain-text robot model (`DESIGN.md` section 4).
* `.mprim` files need no reader here: the C-ABI takes their text as it is (`smplx_space_create`; both the upstream and
  the fork's row format, `manip_lattice_action_space.cpp:103-195`).
"""
from __future__ import annotations


def parse_env(text: str):
    """-> [(name, (x, y, z), (dim_x, dim_y, dim_z))]; boxes are centred at (x, y, z) (`GetCollisionCube`)."""
    tok = text.split()
    if not tok:
        return []
    n = int(tok[0])             # atoi of the first token (:183)
    out = []
    k = 1
    for _ in range(n):
        if k >= len(tok):
            break
        name = tok[k]
        k += 1
        vals = []
        for _ in range(6):
            vals.append(float(tok[k]) if k < len(tok) else 0.0)   # missing numbers stay 0 as in the reference's loop
            k += 1
        out.append((name, tuple(vals[:3]), tuple(vals[3:])))
    return out


def sphere_lines_from_collision_yaml(text: str, links=None, rename=None):
    """`sphere <link> <name> x y z radius priority` lines for the links in `links` (default: every link that lists
    spheres).  `rename` maps YAML link names to the names used in the plain-text model."""
    import yaml
    doc = yaml.safe_load(text)
    models = (doc.get("robot_collision_model") or {}).get("spheres_models") or []
    lines = []
    for m in models:
        link = m.get("link_name")
        if links is not None and link not in links:
            continue
        for sp in m.get("spheres") or []:
            lines.append("sphere %s %s %r %r %r %r %d" % ((rename or {}).get(link, link), sp["name"], float(sp["x"]), float(sp["y"]),
                                                          float(sp["z"]), float(sp["radius"]), int(sp.get("priority", 1))))
    return lines

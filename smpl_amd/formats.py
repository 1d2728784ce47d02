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


def group_links_from_collision_yaml(text: str, group: str, urdf_xml: str | None = None):
    """Links of a collision group: nested groups first, then its chains (base -> tip, resolved through the URDF's
    parent links), then its own links, in file order (`collision_groups` of the YAML;
    sbpl_collision_checking/src/collision_model_config.cpp, robot_collision_model.cpp:517-623)."""
    import yaml
    doc = yaml.safe_load(text)
    groups = {g["name"]: g for g in ((doc.get("robot_collision_model") or {}).get("collision_groups") or [])}
    parent_of = None
    if urdf_xml is not None:
        import xml.etree.ElementTree as ET
        parent_of = {j.find("child").get("link"): j.find("parent").get("link") for j in ET.fromstring(urdf_xml).findall("joint")}

    def chain_links(base, tip):
        if parent_of is None:
            raise ValueError("group %s has a chain (%s -> %s): the URDF is needed to resolve it" % (group, base, tip))
        out, l = [], tip
        while True:
            out.append(l)
            if l == base:
                return list(reversed(out))
            if l not in parent_of:
                raise ValueError("chain tip %s is not below %s" % (tip, base))
            l = parent_of[l]

    def expand(name, seen):
        if name in seen or name not in groups:
            return []
        seen.add(name)
        g = groups[name]
        out = []
        for sub in g.get("groups") or []:
            out += expand(sub, seen)
        for c in g.get("chains") or []:
            out += chain_links(c["base"], c["tip"])
        out += [l["name"] for l in g.get("links") or []]
        return out
    links = []
    for l in expand(group, set()):
        if l not in links:
            links.append(l)
    return links


def urdf_to_robot_text(urdf_xml: str, group_name: str, group_links, planning_joints, planning_link: str,
                       sphere_lines=(), acm_pairs=(), root: str | None = None, held_at=None) -> str:
    """URDF subset -> this repo's plain-text robot model (what
    sbpl_collision_checking/src/robot_collision_model.cpp:117-286 takes from a urdf::ModelInterface):
    `<link name>`, `<joint name type>` with `<parent link>`, `<child link>`, `<origin xyz rpy>`, `<axis xyz>`,
    `<limit lower upper>`; joint types fixed / revolute / continuous / prismatic (floating and planar joints are not
    part of this path).  Only the part of the tree that leads to the group's links and the planning link is kept
    (everything else would be voxelised out-of-group geometry, which is out of scope).  Joints that are neither
    planning joints nor fixed are held at `held_at[name]` (default 0), as the reference holds the joints outside the
    planning group at their current value."""
    import xml.etree.ElementTree as ET
    rob = ET.fromstring(urdf_xml)
    if rob.tag != "robot":
        raise ValueError("not a URDF: root element is <%s>" % rob.tag)
    links = [l.get("name") for l in rob.findall("link")]
    joints = []
    for j in rob.findall("joint"):
        t = j.get("type")
        if t not in ("fixed", "revolute", "continuous", "prismatic"):
            raise ValueError("joint %s: type %s is not supported on this path" % (j.get("name"), t))
        o = j.find("origin")
        xyz = [float(x) for x in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
        rpy = [float(x) for x in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
        a = j.find("axis")
        axis = [float(x) for x in (a.get("xyz", "1 0 0") if a is not None else "1 0 0").split()]   # URDF default axis
        lim = j.find("limit")
        lo = float(lim.get("lower", "0")) if lim is not None else 0.0
        hi = float(lim.get("upper", "0")) if lim is not None else 0.0
        joints.append(dict(name=j.get("name"), type=t, parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                           xyz=xyz, rpy=rpy, axis=axis, lo=lo, hi=hi))
    children = {j["child"] for j in joints}
    if root is None:
        roots = [l for l in links if l not in children]
        if len(roots) != 1:
            raise ValueError("cannot tell the root link: %r" % roots)
        root = roots[0]
    by_child = {j["child"]: j for j in joints}
    keep_links, keep_joints = {root}, []
    for target in list(group_links) + [planning_link]:
        chain = []
        l = target
        while l != root:
            if l not in by_child:
                raise ValueError("link %s is not connected to the root %s" % (target, root))
            chain.append(by_child[l])
            l = by_child[l]["parent"]
        for j in reversed(chain):
            if j["name"] not in [k["name"] for k in keep_joints]:
                keep_joints.append(j)
            keep_links.add(j["child"])
    held = dict(held_at or {})
    out = ["robot " + rob.get("name", "robot"), "link " + root]
    out += ["link " + l for l in links if l in keep_links and l != root]
    for j in joints:                       # file order, restricted to the kept sub-tree
        if j not in keep_joints:
            continue
        t = j["type"]
        xyz, rpy = list(j["xyz"]), j["rpy"]
        if t != "fixed" and j["name"] not in planning_joints:
            # a joint outside the planning group: frozen.  Only a zero value keeps the origin as it is.
            v = float(held.get(j["name"], 0.0))
            if v != 0.0:
                raise ValueError("joint %s is outside the planning group and held at %r: fold it into the origin first" % (j["name"], v))
            t = "fixed"
        out.append("joint %s %s %s %s  %r %r %r  %r %r %r  %r %r %r  %r %r" % (
            j["name"], t, j["parent"], j["child"], xyz[0], xyz[1], xyz[2], rpy[0], rpy[1], rpy[2],
            j["axis"][0], j["axis"][1], j["axis"][2], j["lo"], j["hi"]))
    out += [s for s in sphere_lines if s.split()[1] in keep_links]
    out.append("group %s %s" % (group_name, " ".join(group_links)))
    out += ["acm %s %s" % (a, b) for (a, b) in acm_pairs if a in keep_links and b in keep_links]
    out.append("planning_joints " + " ".join(planning_joints))
    out.append("planning_link " + planning_link)
    return "\n".join(out) + "\n"

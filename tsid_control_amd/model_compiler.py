"""Offline model compiler: URDF + SRDF (TSID side) and MJCF + STL hulls (sim side) -> one flat blob.

The reference loads its robot through pinocchio's URDF/SRDF parsers (ctrl/WalkController.py:13-23)
and MuJoCo's MJCF compiler (main.py:50-52).  Neither library exists here, so this module restates
what those loaders produce for the asset files the reference config names (ctrl/conf.py:9-15):

* TSID side  - joint tree in pinocchio's order (children sorted by joint name, depth first; this
  reproduces the indices hard-coded in main.py:14-42), joint placements, body spatial inertias with
  fixed-joint children merged into their parent joint's body (the two 0.01 kg sole links,
  robot_mod.urdf:176-191), the two sole frames, effort/velocity limits and the SRDF "standing"
  configuration (robot.srdf:4-26).
* sim side   - MuJoCo body tree (robot.xml:55-241), inertias, joint armature/frictionloss
  (robot.xml:8), position-actuator kp and the kv MuJoCo derives from dampratio at compile time
  (robot.xml:9), dof/body inverse weights at qpos0 (used for constraint regularisation), the convex
  hulls of the 21 collision meshes pre-transformed into their body frames (with the hull vertex
  graph MuJoCo's plane<->mesh collider walks), contact excludes (robot.xml:18-52) and the
  TSID->MuJoCo joint permutation (main.py:11-44).

It is run in the build container (where /root/reference exists); the blob it writes
(tsid_control_amd/assets/op3_v1.tsidb) is committed and is the only thing that travels to the GPU
box.  STL files, URDF and MJCF never travel.

Blob layout (little endian): magic "TSIDBM01", u32 n_sections, u32 reserved, then n_sections table
entries {char name[24]; u32 dtype (0=f64, 1=i32); u32 count; u64 byte_offset}, then 8-byte aligned
payloads.  include/tsidb_model.h documents every section.
"""
from __future__ import annotations

import struct
import sys
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

MAGIC = b"TSIDBM01"


# --------------------------------------------------------------------------- small SO(3)/SE(3) kit
def rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def quat_wxyz_to_R(q):
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def sym6_to_mat(i6):
    """(xx, xy, xz, yy, yz, zz) -> 3x3."""
    xx, xy, xz, yy, yz, zz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


def mat_to_sym6(m):
    return np.array([m[0, 0], m[0, 1], m[0, 2], m[1, 1], m[1, 2], m[2, 2]], dtype=np.float64)


class Inertia:
    """Rigid-body inertia: mass, com (in the carrying frame), rotational inertia about the com."""

    def __init__(self, mass, com, I):
        self.mass = float(mass)
        self.com = np.asarray(com, dtype=np.float64).copy()
        self.I = np.asarray(I, dtype=np.float64).copy()

    def transformed(self, R, p):
        """Express in a parent frame given this frame's placement (R, p) in the parent."""
        return Inertia(self.mass, R @ self.com + p, R @ self.I @ R.T)

    def __add__(self, other):
        m = self.mass + other.mass
        c = (self.mass * self.com + other.mass * other.com) / m
        da, db = self.com - c, other.com - c
        I = (self.I - self.mass * skew(da) @ skew(da)) + (other.I - other.mass * skew(db) @ skew(db))
        return Inertia(m, c, I)

    def pack(self):
        return np.concatenate([[self.mass], self.com, mat_to_sym6(self.I)])


# --------------------------------------------------------------------------- TSID side (URDF/SRDF)
def _floats(s):
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def parse_urdf(urdf_path: Path):
    root = ET.parse(urdf_path).getroot()
    links = {}
    for ln in root.findall("link"):
        ine = ln.find("inertial")
        if ine is None:
            links[ln.get("name")] = None
            continue
        o = ine.find("origin")
        xyz = _floats(o.get("xyz", "0 0 0")) if o is not None else np.zeros(3)
        rpy = _floats(o.get("rpy", "0 0 0")) if o is not None else np.zeros(3)
        it = ine.find("inertia")
        I = sym6_to_mat([float(it.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")])
        Rin = rpy_to_R(rpy)
        links[ln.get("name")] = Inertia(float(ine.find("mass").get("value")), xyz, Rin @ I @ Rin.T)
    joints = []
    for j in root.findall("joint"):
        o = j.find("origin")
        lim = j.find("limit")
        joints.append(dict(
            name=j.get("name"), type=j.get("type"),
            parent=j.find("parent").get("link"), child=j.find("child").get("link"),
            xyz=_floats(o.get("xyz", "0 0 0")), rpy=_floats(o.get("rpy", "0 0 0")),
            axis=_floats(j.find("axis").get("xyz")) if j.find("axis") is not None else np.array([1.0, 0, 0]),
            effort=float(lim.get("effort")) if lim is not None else 0.0,
            velocity=float(lim.get("velocity")) if lim is not None else 0.0,
        ))
    return links, joints


def build_tsid_model(urdf_path: Path, srdf_path: Path, frame_names):
    """Pinocchio-ordered joint tree. Joint 0 is the free-flyer root carrying the root link."""
    links, joints = parse_urdf(urdf_path)
    children = {j["child"] for j in joints}
    roots = [l for l in links if l not in children]
    assert len(roots) == 1, roots
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)
    for v in by_parent.values():
        v.sort(key=lambda j: j["name"])  # urdfdom keeps child joints in a name-sorted map

    names = ["root_joint"]
    parent = [-1]
    place = [(np.eye(3), np.zeros(3))]
    inert = [links[roots[0]]]
    eff, vel = [], []
    frames = {}

    def walk(link, jidx, R_acc, p_acc):
        # (R_acc, p_acc): placement of `link`'s frame in joint jidx's frame (identity unless fixed)
        for j in by_parent.get(link, []):
            Rj, pj = rpy_to_R(j["rpy"]), j["xyz"]
            R = R_acc @ Rj
            p = R_acc @ pj + p_acc
            if j["type"] == "fixed":
                if links[j["child"]] is not None:
                    inert[jidx] = inert[jidx] + links[j["child"]].transformed(R, p)
                frames[j["name"]] = (jidx, R, p)
                walk(j["child"], jidx, R, p)
            elif j["type"] in ("revolute", "continuous"):
                assert np.allclose(j["axis"], [0, 0, 1]), "only z-axis hinges are supported"
                names.append(j["name"])
                parent.append(jidx)
                place.append((R, p))
                inert.append(links[j["child"]])
                eff.append(j["effort"])
                vel.append(j["velocity"])
                walk(j["child"], len(names) - 1, np.eye(3), np.zeros(3))
            else:
                raise ValueError(j["type"])

    walk(roots[0], 0, np.eye(3), np.zeros(3))

    # SRDF reference configuration
    sr = ET.parse(srdf_path).getroot()
    q0 = np.zeros(7 + len(names) - 1)
    q0[6] = 1.0
    for gs in sr.findall("group_state"):
        if gs.get("name") != "standing":
            continue
        for j in gs.findall("joint"):
            vals = _floats(j.get("value"))
            if j.get("name") == "root_joint":
                q0[:7] = vals
            else:
                q0[7 + names.index(j.get("name")) - 1] = vals[0]
    return dict(
        names=names, parent=np.array(parent, np.int32),
        place=np.array([np.concatenate([R.reshape(-1), p]) for R, p in place]),
        inertia=np.array([I.pack() for I in inert]),
        effort=np.array(eff), velocity=np.array(vel), q0=q0,
        frame_parent=np.array([frames[n][0] for n in frame_names], np.int32),
        frame_place=np.array([np.concatenate([frames[n][1].reshape(-1), frames[n][2]]) for n in frame_names]),
    )


# --------------------------------------------------------------------------- sim side (MJCF/STL)
def load_stl_vertices(path: Path):
    with open(path, "rb") as f:
        f.seek(80)
        n = struct.unpack("<I", f.read(4))[0]
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
        d = np.frombuffer(f.read(n * 50), dtype=rec)
    return np.unique(d["v"].reshape(-1, 3).astype(np.float64), axis=0)


def hull_volume_centroid(verts):
    """Centre of mass of the solid convex hull (what MuJoCo re-centres a mesh geom on, so what its convex
    collision functions use as the geom centre): volume-weighted mean of the tetrahedra (o, a, b, c) over the
    hull's triangles, o = an interior point."""
    from scipy.spatial import ConvexHull

    h = ConvexHull(verts)
    o = verts[h.vertices].mean(0)
    a, b, c = (verts[h.simplices[:, k]] - o for k in range(3))
    vol = np.abs(np.einsum("ij,ij->i", a, np.cross(b, c))) / 6.0
    return o + ((a + b + c) / 4.0 * vol[:, None]).sum(0) / vol.sum()


def convex_hull_graph(verts):
    """Hull vertices + neighbour lists (edges of the triangulated hull, as qhull 'Qt' gives MuJoCo)."""
    from scipy.spatial import ConvexHull

    h = ConvexHull(verts)
    ids = np.sort(h.vertices)
    remap = -np.ones(len(verts), np.int64)
    remap[ids] = np.arange(len(ids))
    nb = [set() for _ in ids]
    for tri in h.simplices:
        a, b, c = remap[tri]
        nb[a].update((b, c)); nb[b].update((a, c)); nb[c].update((a, b))
    adr = np.zeros(len(ids) + 1, np.int32)
    edges = []
    for i, s in enumerate(nb):
        e = sorted(s)
        edges.extend(e)
        adr[i + 1] = adr[i] + len(e)
    return verts[ids], adr, np.array(edges, np.int32)


CHUNK = 64  # hull vertices per spatial chunk (= one wavefront-wide load)


def kd_order(pts, leaf=CHUNK):
    """Permutation that groups points into spatially compact runs of `leaf` (recursive median split on
    the widest axis, left part sized to a multiple of `leaf`), so that each run has a tight box."""
    idx = np.arange(len(pts))

    def rec(ids):
        if len(ids) <= leaf:
            return [ids]
        ext = pts[ids].max(0) - pts[ids].min(0)
        ax = int(np.argmax(ext))
        order = ids[np.argsort(pts[ids, ax], kind="stable")]
        half = ((len(ids) // 2 + leaf - 1) // leaf) * leaf
        return rec(order[:half]) + rec(order[half:])

    return np.concatenate(rec(idx))


def build_sim_model(mjcf_robot: Path, mesh_dir: Path):
    root = ET.parse(mjcf_robot).getroot()
    dflt = root.find("default").find("default")
    jd = dflt.find("joint")
    pd = dflt.find("position")
    frictionloss, armature = float(jd.get("frictionloss")), float(jd.get("armature"))
    kp, dampratio = float(pd.get("kp")), float(pd.get("dampratio"))

    mesh_file = {}
    for m in root.find("asset").findall("mesh"):
        fn = m.get("file").replace("\\", "/")
        mesh_file[Path(fn).stem] = fn

    bodies = []

    def walk(elem, parent):
        for b in elem.findall("body"):
            idx = len(bodies)
            ine = b.find("inertial")
            fi = _floats(ine.get("fullinertia"))  # xx yy zz xy xz yz
            I = sym6_to_mat([fi[0], fi[3], fi[4], fi[1], fi[5], fi[2]])
            jn = b.find("joint")
            fj = b.find("freejoint")
            geoms = [g for g in b.findall("geom") if g.get("class") == "collision"]
            assert len(geoms) == 1
            g = geoms[0]
            bodies.append(dict(
                name=b.get("name"), parent=parent,
                pos=_floats(b.get("pos", "0 0 0")), quat=_floats(b.get("quat", "1 0 0 0")),
                inertia=Inertia(float(ine.get("mass")), _floats(ine.get("pos")), I),
                joint=("free" if fj is not None else jn.get("name")),
                gpos=_floats(g.get("pos", "0 0 0")), gquat=_floats(g.get("quat", "1 0 0 0")),
                mesh=g.get("mesh"),
            ))
            if jn is not None:
                assert np.allclose(_floats(jn.get("axis")), [0, 0, 1]) and jn.get("type") == "hinge"
            walk(b, idx)

    walk(root.find("worldbody"), -1)
    name_to_idx = {b["name"]: i for i, b in enumerate(bodies)}
    excl = [(name_to_idx[e.get("body1")], name_to_idx[e.get("body2")]) for e in root.find("contact").findall("exclude")]
    actuators = [p.get("joint") for p in root.find("actuator").findall("position")]

    hull_v, hull_adr, edge_adr, edges, rb = [], [0], [0], [], []
    chunk_adr, chunk_box, hull_center, hull_box = [0], [], [], []
    for b in bodies:
        v = load_stl_vertices(mesh_dir / mesh_file[b["mesh"]])
        hv, eadr, e = convex_hull_graph(v)
        Rg = quat_wxyz_to_R(b["gquat"])
        hv_body = hv @ Rg.T + b["gpos"]  # the XML geom frame is the STL frame (MuJoCo's own
        # re-centering of the mesh moves geom_pos/quat and the vertices by inverse amounts)
        # spatial order: runs of CHUNK vertices with a bounding box each (exact pruning of the support search)
        perm = kd_order(hv_body)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        hv_body = hv_body[perm]
        nb_lists = [sorted(int(inv[j]) for j in e[eadr[old]:eadr[old + 1]]) for old in perm]
        eadr = np.concatenate([[0], np.cumsum([len(x) for x in nb_lists])]).astype(np.int32)
        e = np.array([j for x in nb_lists for j in x], np.int32)
        for c0 in range(0, len(hv_body), CHUNK):
            blk = hv_body[c0:c0 + CHUNK]
            lo, hi = blk.min(0), blk.max(0)
            chunk_box.append(np.concatenate([0.5 * (lo + hi), 0.5 * (hi - lo)]))
        chunk_adr.append(len(chunk_box))
        assert chunk_adr[-1] - chunk_adr[-2] <= 64, "a hull may have at most 64 chunks (4096 vertices)"
        c = 0.5 * (hv_body.min(0) + hv_body.max(0))
        rb.append(np.concatenate([c, [np.linalg.norm(hv_body - c, axis=1).max()]]))
        hull_center.append(hull_volume_centroid(hv_body))
        hull_box.append(np.concatenate([c, 0.5 * (hv_body.max(0) - hv_body.min(0))]))
        hull_v.append(hv_body)
        edge_adr.extend((eadr[1:] + len(edges)).tolist())
        edges.extend(e.tolist())
        hull_adr.append(hull_adr[-1] + len(hv_body))
    return dict(
        bodies=bodies, excl=np.array(excl, np.int32), actuators=actuators,
        frictionloss=frictionloss, armature=armature, kp=kp, dampratio=dampratio,
        hull_v=np.concatenate(hull_v), hull_adr=np.array(hull_adr, np.int32),
        edge_adr=np.array(edge_adr, np.int32), edges=np.array(edges, np.int32), rbound=np.array(rb),
        chunk_adr=np.array(chunk_adr, np.int32), chunk_box=np.array(chunk_box),
        hull_center=np.array(hull_center), hull_box=np.array(hull_box),
    )


def sim_constants_at_qpos0(sim):
    """dof_M0, dof_invweight0, body_invweight0, actuator kv: what MuJoCo's compiler derives at qpos0.

    Velocity convention of the free joint: qvel[0:3] world-frame linear velocity of the body origin,
    qvel[3:6] body-frame angular velocity; each hinge dof follows in document order.
    """
    bodies = sim["bodies"]
    nb = len(bodies)
    nv = 6 + nb - 1
    Rw, pw = [None] * nb, [None] * nb
    for i, b in enumerate(bodies):
        R, p = quat_wxyz_to_R(b["quat"]), b["pos"]
        if b["parent"] < 0:
            Rw[i], pw[i] = R, p
        else:
            Rw[i], pw[i] = Rw[b["parent"]] @ R, Rw[b["parent"]] @ p + pw[b["parent"]]
    dof_body = [0] * 6 + list(range(1, nb))  # dof -> body it moves

    def jac(point, body):
        """6 x nv Jacobian (linear; angular) of a point fixed to `body`, world frame."""
        J = np.zeros((6, nv))
        chain = []
        k = body
        while k >= 0:
            chain.append(k)
            k = bodies[k]["parent"]
        for k in chain:
            if k == 0:
                J[:3, 0:3] = np.eye(3)
                J[3:, 3:6] = Rw[0]
                J[:3, 3:6] = -skew(point - pw[0]) @ Rw[0]
            else:
                d = 6 + k - 1
                ax = Rw[k][:, 2]
                J[3:, d] = ax
                J[:3, d] = np.cross(ax, point - pw[k])
        return J

    M = np.zeros((nv, nv))
    coms = []
    for i, b in enumerate(bodies):
        c = Rw[i] @ b["inertia"].com + pw[i]
        coms.append(c)
        J = jac(c, i)
        Iw = Rw[i] @ b["inertia"].I @ Rw[i].T
        M += b["inertia"].mass * J[:3].T @ J[:3] + J[3:].T @ Iw @ J[3:]
    arm = np.array([0.0] * 6 + [sim["armature"]] * (nv - 6))
    M += np.diag(arm)
    Minv = np.linalg.inv(M)
    dof_M0 = np.diag(M).copy()
    dinv = np.diag(Minv).copy()
    dof_invweight0 = dinv.copy()
    dof_invweight0[0:3] = dinv[0:3].mean()
    dof_invweight0[3:6] = dinv[3:6].mean()
    body_invweight0 = np.zeros((nb, 2))
    for i in range(nb):
        J = jac(coms[i], i)
        A = J @ Minv @ J.T
        body_invweight0[i] = [np.trace(A[:3, :3]) / 3, np.trace(A[3:, 3:]) / 3]
    joint_names = [b["joint"] for b in bodies]
    act_dof = np.array([6 + joint_names.index(a) - 1 for a in sim["actuators"]], np.int32)
    kv = sim["dampratio"] * 2.0 * np.sqrt(sim["kp"] * dof_M0[act_dof])
    return dict(M0=M, dof_M0=dof_M0, dof_invweight0=dof_invweight0, body_invweight0=body_invweight0,
                act_dof=act_dof, act_kv=kv, arm=arm)


# --------------------------------------------------------------------------- blob writer
def write_blob(path: Path, sections: dict):
    table, payload = [], b""
    hdr = 16 + 40 * len(sections)
    for name, arr in sections.items():
        a = np.ascontiguousarray(arr)
        if a.dtype.kind == "f":
            a, code = a.astype("<f8"), 0
        else:
            a, code = a.astype("<i4"), 1
        pad = (-(hdr + len(payload))) % 8
        payload += b"\0" * pad
        table.append((name.encode(), code, a.size, hdr + len(payload)))
        payload += a.tobytes()
    out = MAGIC + struct.pack("<II", len(sections), 0)
    for nm, code, cnt, off in table:
        assert len(nm) < 24
        out += nm.ljust(24, b"\0") + struct.pack("<IIQ", code, cnt, off)
    out += payload
    path.write_bytes(out)


def compile_model(ref_root: Path, out: Path):
    v1 = ref_root / "robot" / "v1"
    frame_names = ["left_sole_joint_fixed", "right_sole_joint_fixed"]  # ctrl/conf.py:17-18
    ts = build_tsid_model(v1 / "urdf" / "robot_mod.urdf", v1 / "urdf" / "robot.srdf", frame_names)
    sim = build_sim_model(v1 / "mujoco" / "robot.xml", v1 / "mujoco" / "assets")
    return assemble_blob(ts, sim, out)


def assemble_blob(ts, sim, out: Path):
    """TSID model + sim model -> blob sections (both robots)."""
    c0 = sim_constants_at_qpos0(sim)
    bodies = sim["bodies"]
    nb = len(bodies)

    # TSID q index feeding each sim actuator (main.py:11-44), derived by joint name
    ctrl_qidx = np.array([7 + ts["names"].index(a) - 1 for a in sim["actuators"]], np.int32)
    # sim joint (document order) -> TSID joint index (for closed-loop coupling)
    sim_joint_names = [b["joint"] for b in bodies[1:]]
    sim2tsid = np.array([ts["names"].index(n) - 1 for n in sim_joint_names], np.int32)

    # collision geoms: one per body in the v1 robot (robot.xml:13-15), several in the v0 robot
    geom_body = np.array([g["body"] for g in sim["geoms"]], np.int32) if "geoms" in sim else np.arange(nb, dtype=np.int32)
    ng = len(geom_body)
    # candidate robot<->robot geom pairs: different bodies, minus parent-child, minus excludes (body pairs)
    exc = {tuple(sorted(map(int, e))) for e in sim["excl"]}
    pairs = [(i, j) for i in range(ng) for j in range(i + 1, ng)
             if geom_body[i] != geom_body[j] and bodies[geom_body[j]]["parent"] != geom_body[i]
             and bodies[geom_body[i]]["parent"] != geom_body[j] and tuple(sorted((int(geom_body[i]), int(geom_body[j])))) not in exc]
    condim = int(sim.get("condim", 3))
    friction = np.asarray(sim.get("friction", [1.0, 0.005, 0.0001]), dtype=np.float64)
    solref = np.asarray(sim.get("solref", [0.02, 1.0]), dtype=np.float64)
    damping = float(sim.get("damping", 0.0))
    nv = 6 + nb - 1
    big = 1e300
    cr, fr = sim.get("ctrlrange", (-big, big)), sim.get("forcerange", (-big, big))

    sections = {
        # NJ NQ NV NA NB has_sim | NG condim eulerdamp: which libtsidb build the blob is for
        "model_dims": np.array([len(ts["names"]), len(ts["names"]) + 6, len(ts["names"]) + 5, len(ts["names"]) - 1, nb, 1,
                                ng, condim, 1 if damping > 0 else 0], np.int32),
        # ---- TSID side
        "pin_parent": ts["parent"], "pin_place": ts["place"], "pin_inertia": ts["inertia"],
        "pin_frame_parent": ts["frame_parent"], "pin_frame_place": ts["frame_place"],
        "pin_effort": ts["effort"], "pin_velocity": ts["velocity"], "pin_q0": ts["q0"],
        # ---- sim side
        "mj_parent": np.array([b["parent"] for b in bodies], np.int32),
        "mj_pos": np.array([b["pos"] for b in bodies]),
        "mj_quat": np.array([b["quat"] / np.linalg.norm(b["quat"]) for b in bodies]),
        "mj_inertia": np.array([b["inertia"].pack() for b in bodies]),
        "mj_armature": c0["arm"],
        "mj_frictionloss": np.array([0.0] * 6 + [sim["frictionloss"]] * (nb - 1)),
        "mj_damping": np.array([0.0] * 6 + [damping] * (nb - 1)),
        "mj_geom_body": geom_body,
        "mj_act_range": np.tile(np.array([cr[0], cr[1], fr[0], fr[1]], dtype=np.float64), (len(c0["act_dof"]), 1)),
        "mj_dof_M0": c0["dof_M0"], "mj_dof_invw0": c0["dof_invweight0"],
        "mj_body_invw0": c0["body_invweight0"],
        "mj_act_dof": c0["act_dof"], "mj_act_kp": np.full(len(c0["act_dof"]), sim["kp"]),
        "mj_act_kv": c0["act_kv"],
        "mj_ctrl_qidx": ctrl_qidx, "mj_sim2tsid": sim2tsid,
        "mj_hull_adr": sim["hull_adr"], "mj_hull_vert": sim["hull_v"],
        "mj_hull_eadr": sim["edge_adr"], "mj_hull_edge": sim["edges"],
        "mj_rbound": sim["rbound"], "mj_pairs": np.array(pairs, np.int32),
        "mj_chunk_adr": sim["chunk_adr"], "mj_chunk_box": sim["chunk_box"],
        "mj_hull_center": sim["hull_center"], "mj_hull_box": sim["hull_box"],
        # MuJoCo option / contact defaults (no <option>, no geom contact attrs in the MJCF)
        "mj_opt": np.array([0.002, -9.81, 1e-8, 100, 50, 0.01, 1.0]),  # dt gz tol iters ls_iters ls_tol impratio
        # mu (sliding), solref[2], solimp[5] (MuJoCo defaults), condim, torsional friction, margin, spare
        "mj_contact": np.array([friction[0], solref[0], solref[1], 0.9, 0.95, 0.001, 0.5, 2.0, condim, friction[1],
                                float(sim.get("margin", 0.0)), 0.0]),
    }
    write_blob(out, sections)
    return ts, sim, c0, sections


def emit_topology_header(mj_parent, path: Path, nj=None, na=None, has_sim=True, ng=None, condim=3, eulerdamp=False, npair=0):
    """Compile-time facts of ONE robot for libtsidb: the dimensions (TSID joints incl. the free-flyer, nq, nv,
    actuated joints, sim bodies) and the sim tree's dof ancestry for the register Cholesky (tree-sparse elimination
    without run-time branches).  One library is built per robot (`-DTSIDB_TOPOLOGY_HEADER=...`); tsidb_create refuses
    a blob whose dimensions or tree differ.  has_sim=False: a TSID-only robot (no MJCF compiled): the sim tree is a
    placeholder and the library is built without the sim stage."""
    nb = len(mj_parent)
    nv = 6 + nb - 1
    nj = nb if nj is None else nj
    na = nv - 6 if na is None else na
    anc_body = []
    for b in range(nb):
        a, m = b, 0
        while a >= 0:
            m |= 1 << a
            a = int(mj_parent[a])
        anc_body.append(m)
    dofanc = []
    for k in range(nv):
        if k < 6:
            dofanc.append((1 << k) - 1)                       # root block is dense
        else:
            m = 0
            for a in range(nb):
                if a != k - 5 and (anc_body[k - 5] >> a) & 1:
                    m |= 0x3F if a == 0 else 1 << (5 + a)
            dofanc.append(m)
    txt = ("// GENERATED by tsid_control_amd/model_compiler.py from the compiled model blob.\n"
           "// Dimensions of the robot this library is built for, and MJ_DOFANC[k]: bitmask of the sim dofs that are\n"
           "// strict ancestors of dof k (lower indices).\n"
           "#pragma once\nnamespace tsidb {\n"
           f"constexpr int TOPO_NJ = {nj}, TOPO_NQ = {nj + 6}, TOPO_NV = {nj + 5}, TOPO_NA = {na};\n"
           f"constexpr int TOPO_HAS_SIM = {1 if has_sim else 0};\n"
           f"constexpr int TOPO_NB = {nb};\n"
           "// sim stage: collision geoms (convex hulls; several per body are allowed), contact dimension (3: pyramid of 4\n"
           "// rows, 4: + torsional friction, 6 rows), joint damping integrated implicitly (MuJoCo's Euler), room for the\n"
           "// candidate geom pairs (a multiple of 64)\n"
           f"constexpr int TOPO_NG = {nb if ng is None else ng}, TOPO_CONDIM = {condim}, TOPO_EULERDAMP = {1 if eulerdamp else 0}, "
           f"TOPO_MAXPAIR = {max(64, (npair + 63) // 64 * 64)};\n"
           "constexpr int TOPO_PARENT[] = {" + ", ".join(str(int(x)) for x in mj_parent) + "};\n"
           "constexpr unsigned MJ_DOFANC[] = {" + ", ".join(hex(x) + "u" for x in dofanc) + "};\n"
           "} // namespace tsidb\n")
    path.write_text(txt)


def compile_tsid_only(urdf: Path, srdf: Path, frame_names, out: Path):
    """A robot with only its TSID side compiled (URDF + SRDF): the blob carries the pin_* sections and model_dims;
    the library built for it has no sim stage.  Used for the v0 robot (robot/v0/robot.urdf, robot.srdf,
    legacy/op3_conf.py:56-62), whose MJCF uses features outside the built sim subset."""
    ts = build_tsid_model(urdf, srdf, frame_names)
    nj = len(ts["names"])
    sections = {
        "model_dims": np.array([nj, nj + 6, nj + 5, nj - 1, nj, 0], np.int32),  # NJ NQ NV NA NB has_sim
        "pin_parent": ts["parent"], "pin_place": ts["place"], "pin_inertia": ts["inertia"],
        "pin_frame_parent": ts["frame_parent"], "pin_frame_place": ts["frame_place"],
        "pin_effort": ts["effort"], "pin_velocity": ts["velocity"], "pin_q0": ts["q0"],
    }
    write_blob(out, sections)
    return ts, sections


# --------------------------------------------------------------------------- general MJCF (the v0 robot)
def _quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def R_to_quat_wxyz(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def build_sim_model_mjcf(mjcf: Path, mesh_dir: Path):
    """Sim side of a robot from a general single-file MJCF (robot/v0/robot.xml): top-level <default> for joints /
    geoms / position actuators, bodies with several mesh geoms each (every geom whose contype is not 0 collides - in
    this file that includes the "visual" ones: only the default class itself carries class="visualgeom"), a static
    child body welded into its parent (root -> torso), hinge axes +-z (a -z hinge is the +z hinge of the body frame
    turned by pi about x), diaginertia + quat inertials, position actuators with control / force ranges, the ground
    plane.  Returns the dict build_sim_model returns, plus `geoms` (body, hull) and the contact / actuator extras."""
    root = ET.parse(mjcf).getroot()
    dflt = root.find("default")
    jd, gd, pd = dflt.find("joint"), dflt.find("geom"), dflt.find("position")
    damping, armature, frictionloss = (float(jd.get(k, "0")) for k in ("damping", "armature", "frictionloss"))
    g_condim, g_contype = int(gd.get("condim", "3")), int(gd.get("contype", "1"))
    g_friction = _floats(gd.get("friction", "1 0.005 0.0001"))
    g_solref = _floats(gd.get("solref", "0.02 1"))
    g_margin = float(gd.get("margin", "0"))
    kp = float(pd.get("kp", "1"))
    ctrlrange, forcerange = _floats(pd.get("ctrlrange", "-1e300 1e300")), _floats(pd.get("forcerange", "-1e300 1e300"))
    mesh_file = {m.get("name"): m.get("file") for m in root.find("asset").findall("mesh")}
    Rx = np.diag([1.0, -1.0, -1.0])

    bodies, geoms = [], []

    def inertial_of(b):
        ine = b.find("inertial")
        if ine is None:
            return None
        Ri = quat_wxyz_to_R(_floats(ine.get("quat", "1 0 0 0")))
        if ine.get("fullinertia") is not None:
            fi = _floats(ine.get("fullinertia"))
            I = sym6_to_mat([fi[0], fi[3], fi[4], fi[1], fi[5], fi[2]])
        else:
            I = Ri @ np.diag(_floats(ine.get("diaginertia"))) @ Ri.T
        return Inertia(float(ine.get("mass")), _floats(ine.get("pos", "0 0 0")), I)

    def walk(elem, parent, Rp, pp):
        """(Rp, pp): placement, in the carrying body `parent`, of the frame `elem`'s children are expressed in"""
        for b in elem.findall("body"):
            R = Rp @ quat_wxyz_to_R(_floats(b.get("quat", "1 0 0 0")))
            p = Rp @ _floats(b.get("pos", "0 0 0")) + pp
            jn, fj = b.find("joint"), b.find("freejoint")
            if jn is None and fj is None:          # static child: welded into the carrying body
                carrier, Rb, pb = parent, R, p
            else:
                flip = False
                if jn is not None:
                    ax = _floats(jn.get("axis", "0 0 1"))
                    assert jn.get("type", "hinge") == "hinge" and np.allclose(np.abs(ax), [0, 0, 1]), "z hinges only"
                    flip = ax[2] < 0
                if flip:
                    R = R @ Rx
                carrier = len(bodies)
                bodies.append(dict(name=b.get("name"), parent=parent, pos=p, quat=R_to_quat_wxyz(R), inertia=None,
                                   joint=("free" if fj is not None else jn.get("name"))))
                Rb, pb = (Rx if flip else np.eye(3)), np.zeros(3)   # the XML body frame seen from the (turned) body frame
            ine = inertial_of(b)
            if ine is not None:
                ine = ine.transformed(Rb, pb)
                cur = bodies[carrier]["inertia"]
                bodies[carrier]["inertia"] = ine if cur is None else cur + ine
            for g in b.findall("geom"):
                if g.get("type", "sphere") != "mesh" or g.get("class") == "visualgeom" or int(g.get("contype", g_contype)) == 0:
                    continue
                Rg = Rb @ quat_wxyz_to_R(_floats(g.get("quat", "1 0 0 0")))
                pg = Rb @ _floats(g.get("pos", "0 0 0")) + pb
                geoms.append(dict(name=g.get("name"), body=carrier, R=Rg, p=pg, mesh=g.get("mesh")))
            walk(b, carrier, Rb, pb)

    walk(root.find("worldbody"), -1, np.eye(3), np.zeros(3))
    for b in bodies:
        assert b["inertia"] is not None and b["inertia"].mass > 0, f"body {b['name']} has no mass"
    actuators = [p.get("joint") for p in root.find("actuator").findall("position")]

    # hulls per geom, vertices in the BODY frame, k-d ordered chunks (as build_sim_model does per body)
    hull_cache = {}
    hull_v, hull_adr, edge_adr, edges, rb, chunk_adr, chunk_box, hull_center, hull_box = [], [0], [0], [], [], [0], [], [], []
    for g in geoms:
        if g["mesh"] not in hull_cache:
            hull_cache[g["mesh"]] = convex_hull_graph(load_stl_vertices(mesh_dir / mesh_file[g["mesh"]]))
        hv, eadr, e = hull_cache[g["mesh"]]
        hv_body = hv @ g["R"].T + g["p"]
        perm = kd_order(hv_body)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        hv_body = hv_body[perm]
        nb_lists = [sorted(int(inv[j]) for j in e[eadr[old]:eadr[old + 1]]) for old in perm]
        eadr2 = np.concatenate([[0], np.cumsum([len(x) for x in nb_lists])]).astype(np.int32)
        e2 = np.array([j for x in nb_lists for j in x], np.int32)
        for c0 in range(0, len(hv_body), CHUNK):
            blk = hv_body[c0:c0 + CHUNK]
            lo, hi = blk.min(0), blk.max(0)
            chunk_box.append(np.concatenate([0.5 * (lo + hi), 0.5 * (hi - lo)]))
        chunk_adr.append(len(chunk_box))
        c = 0.5 * (hv_body.min(0) + hv_body.max(0))
        rb.append(np.concatenate([c, [np.linalg.norm(hv_body - c, axis=1).max()]]))
        hull_center.append(hull_volume_centroid(hv_body))
        hull_box.append(np.concatenate([c, 0.5 * (hv_body.max(0) - hv_body.min(0))]))
        hull_v.append(hv_body)
        edge_adr.extend((eadr2[1:] + len(edges)).tolist())
        edges.extend(e2.tolist())
        hull_adr.append(hull_adr[-1] + len(hv_body))
    return dict(
        bodies=bodies, geoms=geoms, excl=np.zeros((0, 2), np.int32), actuators=actuators,
        frictionloss=frictionloss, armature=armature, damping=damping, kp=kp, dampratio=0.0,
        ctrlrange=ctrlrange, forcerange=forcerange,
        condim=max(g_condim, 3), friction=g_friction, solref=g_solref, margin=g_margin,
        hull_v=np.concatenate(hull_v), hull_adr=np.array(hull_adr, np.int32),
        edge_adr=np.array(edge_adr, np.int32), edges=np.array(edges, np.int32), rbound=np.array(rb),
        chunk_adr=np.array(chunk_adr, np.int32), chunk_box=np.array(chunk_box),
        hull_center=np.array(hull_center), hull_box=np.array(hull_box),
    )


if __name__ == "__main__":
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    out = Path(sys.argv[2] if len(sys.argv) > 2 else Path(__file__).parent / "assets" / "op3_v1.tsidb")
    ts, sim, c0, sec = compile_model(ref, out)
    emit_topology_header(sec["mj_parent"], Path(__file__).parent / "csrc" / "tsidb_topology.hpp",
                         ng=len(sec["mj_geom_body"]), npair=len(sec["mj_pairs"]))
    # the v0 robot (legacy/op3_conf.py:53-62): robot/v0/robot.urdf + robot.srdf, robot/v0/robot.xml + meshes/*.stl
    v0 = ref / "robot" / "v0"
    ts0 = build_tsid_model(v0 / "robot.urdf", v0 / "robot.srdf", ["leg_left_sole_joint_fixed", "leg_right_sole_joint_fixed"])
    sim0 = build_sim_model_mjcf(v0 / "robot.xml", v0 / "meshes")
    _, _, _, sec0 = assemble_blob(ts0, sim0, out.parent / "op3_v0.tsidb")
    emit_topology_header(sec0["mj_parent"], Path(__file__).parent / "csrc" / "tsidb_topology_v0.hpp",
                         ng=len(sec0["mj_geom_body"]), condim=int(sec0["mj_contact"][8]), eulerdamp=bool(sec0["mj_damping"].any()),
                         npair=len(sec0["mj_pairs"]))
    print("v0 joints:", ts0["names"], "mass", ts0["inertia"][:, 0].sum(), "geoms", len(sec0["mj_geom_body"]), "pairs", len(sec0["mj_pairs"]))
    print("joints:", ts["names"])
    print("total mass (TSID):", ts["inertia"][:, 0].sum(), " (sim):", sum(b["inertia"].mass for b in sim["bodies"]))
    print("hull verts:", len(sim["hull_v"]), "edges:", len(sim["edges"]), "blob bytes:", out.stat().st_size)
    print("act kv:", c0["act_kv"])
    print("ctrl_qidx:", sec["mj_ctrl_qidx"])

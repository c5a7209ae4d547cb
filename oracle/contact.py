"""oracle/contact.py — CPU restatement of STARK's contact detection and contact/friction table construction.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/, __graft_entry__.smoke() and nothing else.

Follows, with numpy over all candidate pairs at once (brute force, small cases):
  * narrow-phase classification       TriangleMeshCollisionDetection/src/ipc_toolkit_geometry_functions.cpp:38-330
  * pair enumeration and exclusions   TriangleMeshCollisionDetection/src/BroadPhasePTEEBase.cpp:162-270 (brute force path; the
                                      octree path yields the same set), ProximityDetection.cpp:75-190
  * edge-triangle intersection        ipc_toolkit_geometry_functions.cpp:565-585, BroadPhaseET.cpp:161-165
  * routing into the 21 barrier tables   stark/src/models/interactions/EnergyFrictionalContact.cpp:368-530
  * the 14 lagged friction tables        EnergyFrictionalContact.cpp:531-773, friction_geometry.cpp
  * binding recipes of the 35 potentials EnergyFrictionalContact.cpp:829-1218, :1358-1423
Pinned against the reference's own tables (tests/golden/contact*.npz) by tests/test_oracle_contact.py.
"""
from dataclasses import dataclass, field

import numpy as np

EE_PARALLEL_CUTOFF = 1e-30  # EnergyFrictionalContact.h:71 (edge_edge_cross_norm_sq_cutoff)

# PointTriangleDistanceType / EdgeEdgeDistanceType (ipc_toolkit_geometry_functions.h:12-40)
P_T0, P_T1, P_T2, P_E0, P_E1, P_E2, P_T = range(7)
EA0_EB0, EA0_EB1, EA1_EB0, EA1_EB1, EA_EB0, EA_EB1, EA0_EB, EA1_EB, EA_EB = range(9)


def _dot(a, b):
    return np.einsum("...i,...i->...", a, b)


def _sq(a):
    return _dot(a, a)


# ---- narrow phase --------------------------------------------------------------------------------------------------------
def _edge_param(p, e0, e1, n):
    """point_triangle_unrolled_edge_parametrization (:196-241): coordinates of p in the basis (e1-e0, (e1-e0) x n)."""
    b0 = e1 - e0
    b1 = np.cross(b0, n)
    d = p - e0
    with np.errstate(divide="ignore", invalid="ignore"):
        return _dot(b0, d) / _sq(b0), _dot(b1, d) / _sq(b1)


def point_triangle_type(p, t0, t1, t2):
    """point_triangle_distance_type (:242-273). Arrays [...,3] -> int array."""
    n = np.cross(t1 - t0, t2 - t0)
    a0, b0 = _edge_param(p, t0, t1, n)
    a1, b1 = _edge_param(p, t1, t2, n)
    a2, b2 = _edge_param(p, t2, t0, n)
    out = np.full(a0.shape, P_T, dtype=np.int32)
    done = np.zeros(a0.shape, dtype=bool)

    def take(cond, val):
        nonlocal done
        sel = cond & ~done
        out[sel] = val
        done |= sel

    take((a0 > 0.0) & (a0 < 1.0) & (b0 >= 0.0), P_E0)
    take((a1 > 0.0) & (a1 < 1.0) & (b1 >= 0.0), P_E1)
    take((a2 > 0.0) & (a2 < 1.0) & (b2 >= 0.0), P_E2)
    take((a0 <= 0.0) & (a2 >= 1.0), P_T0)
    take((a1 <= 0.0) & (a0 >= 1.0), P_T1)
    take((a2 <= 0.0) & (a1 >= 1.0), P_T2)
    return out


def _point_line_sq(p, e0, e1):
    return _sq(np.cross(e0 - p, e1 - p)) / _sq(e1 - e0)


def point_triangle_sq_distance(p, t0, t1, t2):
    """point_triangle_sq_distance (:274-303) -> (type, d2)."""
    ty = point_triangle_type(p, t0, t1, t2)
    n = np.cross(t1 - t0, t2 - t0)
    with np.errstate(divide="ignore", invalid="ignore"):
        cand = [
            _sq(t0 - p), _sq(t1 - p), _sq(t2 - p),
            _point_line_sq(p, t0, t1), _point_line_sq(p, t1, t2), _point_line_sq(p, t2, t0),
            _dot(p - t0, n) ** 2 / _sq(n),
        ]
    d2 = np.choose(ty, cand)
    return ty, d2


def edge_edge_type(ea0, ea1, eb0, eb1):
    """edge_edge_distance_type (:79-168) for NON-parallel edges (|u x v|^2 >= cutoff); parallel pairs are dropped by the
    caller exactly as ProximityDetection.cpp:152-155 does, whatever their type."""
    u = ea1 - ea0
    v = eb1 - eb0
    w = ea0 - eb0
    a = _sq(u)
    b = _dot(u, v)
    c = _sq(v)
    d = _dot(u, w)
    e = _dot(v, w)
    D = a * c - b * b
    sN = b * e - c * d
    tN = np.where(sN <= 0.0, e, np.where(sN >= D, e + b, a * e - b * d))
    tD = np.where((sN <= 0.0) | (sN >= D), c, D)
    default = np.where(sN <= 0.0, EA0_EB, np.where(sN >= D, EA1_EB, EA_EB)).astype(np.int32)
    out = default.copy()
    lo = tN <= 0.0
    hi = (tN >= tD) & ~lo
    out[lo] = np.where(-d <= 0.0, EA0_EB0, np.where(-d >= a, EA1_EB0, EA_EB0))[lo]
    out[hi] = np.where((-d + b) <= 0.0, EA0_EB1, np.where((-d + b) >= a, EA1_EB1, EA_EB1))[hi]
    return out


def edge_edge_sq_distance(ea0, ea1, eb0, eb1):
    ty = edge_edge_type(ea0, ea1, eb0, eb1)
    nrm = np.cross(ea1 - ea0, eb1 - eb0)
    with np.errstate(divide="ignore", invalid="ignore"):
        cand = [
            _sq(eb0 - ea0), _sq(eb1 - ea0), _sq(eb0 - ea1), _sq(eb1 - ea1),
            _point_line_sq(eb0, ea0, ea1), _point_line_sq(eb1, ea0, ea1), _point_line_sq(ea0, eb0, eb1), _point_line_sq(ea1, eb0, eb1),
            _dot(eb0 - ea0, nrm) ** 2 / _sq(nrm),
        ]
    return ty, np.choose(ty, cand)


def edge_intersects_triangle(q1, q2, a, b, c):
    """is_edge_intersecting_triangle (:565-585); coplanar configurations report False."""
    e1 = b - a
    e2 = c - a
    n = np.cross(e1, e2)
    dr = q2 - q1
    det = -_dot(dr, n)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        ao = q1 - a
        dao = np.cross(ao, dr)
        u = _dot(e2, dao) * inv
        v = -_dot(e1, dao) * inv
        t = _dot(ao, n) * inv
        return (np.abs(det) >= 1e-14) & (t >= 0.0) & (t <= 1.0) & (u >= 0.0) & (v >= 0.0) & ((u + v) <= 1.0)


# ---- friction geometry (friction_geometry.cpp) ---------------------------------------------------------------------------------
def _normalized(v):
    """Eigen's normalized(): zero vectors are returned unchanged."""
    n = np.sqrt(_sq(v))[..., None]
    return np.where(n > 0, v / np.where(n > 0, n, 1.0), v)


def barycentric_point_triangle(p, a, b, c):
    v0, v1, v2 = b - a, c - a, p - a
    d00, d01, d11, d20, d21 = _dot(v0, v0), _dot(v0, v1), _dot(v1, v1), _dot(v2, v0), _dot(v2, v1)
    inv = 1.0 / (d00 * d11 - d01 * d01)
    v = (d11 * d20 - d01 * d21) * inv
    w = (d00 * d21 - d01 * d20) * inv
    return np.stack([1.0 - v - w, v, w], axis=-1)


def barycentric_point_edge(p, a, b):
    ab = b - a
    alpha = _dot(p - a, ab) / _sq(ab)
    return np.stack([1.0 - alpha, alpha], axis=-1)


def barycentric_edge_edge(A, B, P, Q):
    da, db, r = B - A, Q - P, A - P
    a, e, f, b, c = _dot(da, da), _dot(db, db), _dot(db, r), _dot(da, db), _dot(da, r)
    denom = a * e - b * b
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (b * f - c * e) / denom
        t = (b * s + f) / e
    par = denom < 1e-16
    return np.stack([np.where(par, 0.5, s), np.where(par, 0.5, t)], axis=-1)


def _proj(u, v):
    return np.concatenate([u, v], axis=-1)


def projection_matrix_triangle(a, b, c):
    v01, v02 = a - c, b - c
    u = _normalized(v01)
    return _proj(u, _normalized(np.cross(np.cross(v01, v02), u)))


def projection_matrix_edge_edge(a, b, p, q):
    u = _normalized(b - a)
    n = np.cross(u, q - p)
    return _proj(u, _normalized(np.cross(u, n)))


def projection_matrix_point_point(p, a):
    n = _normalized(p - a)
    e = np.where((n[..., 2] < 0.99)[..., None], np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    u = _normalized(np.cross(e, n))
    return _proj(u, _normalized(np.cross(u, n)))


def projection_matrix_point_edge(p, a, b):
    u = _normalized(b - a)
    return _proj(u, _normalized(np.cross(u, p - a)))


# ---- binding recipes of the 35 potentials ----------------------------------------------------------------------------------------
# A recipe is the list of (role, stride, conn column) in the order of the reference's mws.make_* calls; role names the host
# array: v1 x0 X dt k thick | rb_xloc rb_v1 rb_w1 rb_t0 rb_q0 | T mu fn bary epsv
def _d_x1(cols):
    return [("v1", 3, c) for c in cols] + [("x0", 3, c) for c in cols] + [("dt", 1, -1)]


def _d_X(cols):
    return [("X", 3, c) for c in cols]


def _d_v1(cols):
    return [("v1", 3, c) for c in cols]


def _rb(rb, cols):  # _get_rb_x1 and _get_rb_v1 bind the same arrays (:1358-1369)
    return [("dt", 1, -1)] + [("rb_xloc", 3, c) for c in cols] + [("rb_v1", 3, rb), ("rb_w1", 3, rb), ("rb_t0", 3, rb), ("rb_q0", 4, rb)]


def _rb_X(cols):
    return [("rb_xloc", 3, c) for c in cols]


_TAIL_PT = [("thick", 1, 0), ("thick", 1, 1), ("k", 1, -1)]
_TAIL_EE = [("k", 1, -1), ("thick", 1, 0), ("thick", 1, 1)]


def _fr_tail(nbary):
    return ([("bary", nbary, 0)] if nbary else []) + [("T", 6, 0), ("mu", 1, 0), ("fn", 1, 0), ("epsv", 1, -1), ("dt", 1, -1)]


def _build_recipes():
    R = {}
    # conn layouts (contact_and_friction_data.h): d_d: ga gb | ...; rb_rb: ga gb rba rbb | ...; rb_d: ga gb rb | ...
    for fam, A, B, base in (("d_d", "d", "d", 2), ("rb_rb", "rb", "rb", 4), ("rb_d", "rb", "d", 3)):
        rba, rbb = 2, 3

        def pos(kind, rb, cols):
            return _d_x1(cols) if kind == "d" else _rb(rb, cols)

        def rest(kind, cols):
            return _d_X(cols) if kind == "d" else _rb_X(cols)

        def c(n, start):
            return list(range(start, start + n))

        # point-triangle family: (suffix, KA, KB)
        pts = [("pt_pp", 1, 1), ("pt_pe", 1, 2), ("pt_pt", 1, 3)]
        if fam == "rb_d":
            pts += [("pt_ep", 2, 1), ("pt_tp", 3, 1)]
        for suf, ka, kb in pts:
            ca, cb = c(ka, base), c(kb, base + ka)
            R["contact_%s_%s_cubic" % (fam, suf)] = (base + ka + kb, pos(A, rba, ca) + pos(B, rbb if fam == "rb_rb" else rba, cb) + _TAIL_PT)
        # edge-edge family: (suffix, point on A, point on B)
        ees = [("ee_pp", True, True), ("ee_pe", True, False), ("ee_ee", False, False)]
        if fam == "rb_d":
            ees += [("ee_ep", False, True)]
        for suf, pa, pb in ees:
            col = base
            seq = []
            for kind, rb, has_p in ((A, rba, pa), (B, rbb if fam == "rb_rb" else rba, pb)):
                e = c(2, col)
                col += 2
                seq += pos(kind, rb, e) + rest(kind, e)
                if has_p:
                    seq += pos(kind, rb, [col])
                    col += 1
            R["contact_%s_%s_cubic" % (fam, suf)] = (col, seq + _TAIL_EE)
    # friction: conn = idx | [rba rbb | rb] | points of A | points of B
    for fam, A, B, base in (("d_d", "d", "d", 1), ("rb_rb", "rb", "rb", 3), ("rb_d", "rb", "d", 2)):
        kinds = [("pp", 1, 1, 0), ("pe", 1, 2, 2), ("pt", 1, 3, 3), ("ee", 2, 2, 2)]
        if fam == "rb_d":
            kinds += [("ep", 2, 1, 2), ("tp", 3, 1, 3)]
        for suf, ka, kb, nb in kinds:
            ca = list(range(base, base + ka))
            cb = list(range(base + ka, base + ka + kb))
            sa = _d_v1(ca) if A == "d" else _rb(1, ca)
            sb = _d_v1(cb) if B == "d" else _rb(2 if fam == "rb_rb" else 1, cb)
            R["friction_%s_%s_C0" % (fam, suf)] = (base + ka + kb, sa + sb + _fr_tail(nb))
    return R


RECIPES = _build_recipes()


# ---- collision meshes ----------------------------------------------------------------------------------------------------------------
@dataclass
class Mesh:
    kind: str            # "d" | "rb"
    idx_in_ps: int
    verts: np.ndarray    # per collision vertex: index into the physical system's vertex array
    tris: np.ndarray     # [nt,3] local
    edges: np.ndarray    # [ne,2] local
    thickness: float


@dataclass
class ContactScene:
    meshes: list
    friction: dict = field(default_factory=dict)   # (min mesh, max mesh) -> mu
    disabled: set = field(default_factory=set)     # (min mesh, max mesh)

    def mu(self, a, b):
        return self.friction.get((min(a, b), max(a, b)), 0.0)

    def is_disabled(self, a, b):
        return (min(a, b), max(a, b)) in self.disabled


def scene_from_fixture(man, z):
    c = man["contact"]
    meshes = []
    for k, m in enumerate(c["meshes"]):
        meshes.append(Mesh(m["kind"], m["idx_in_ps"], z["cm%d_verts" % k].astype(np.int64), z["cm%d_tris" % k].reshape(-1, 3).astype(np.int64),
                           z["cm%d_edges" % k].reshape(-1, 2).astype(np.int64), m["thickness"]))
    sc = ContactScene(meshes)
    for a, b, mu in c["friction"]:
        sc.friction[(min(a, b), max(a, b))] = mu
    for k, m in enumerate(meshes):
        if m.kind == "rb":
            sc.disabled.add((k, k))   # _add_rigid_body disables rigid self collision (:208-209)
    return sc


def quat_to_R(q):
    """Eigen::Quaterniond::toRotationMatrix; q = (w, x, y, z) as stored in q0_ (RigidBodyDynamics.h:23)."""
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_time_integration(q0, w, dt):
    """rigidbody_transformations.cpp:30-37: q1 = normalize(q0 + dt/2 (0, w) * q0), (w, x, y, z) storage."""
    e, f, g, h = q0
    wx, wy, wz = w
    p = np.array([-wx * f - wy * g - wz * h, wx * e + wy * h - wz * g, -wx * h + wy * e + wz * f, wx * g - wy * f + wz * e])
    q = np.asarray(q0, dtype=np.float64) + 0.5 * dt * p
    return q / np.sqrt((q * q).sum())


def mesh_vertices(scene, state, dt):
    """EnergyFrictionalContact::_update_vertices (:219-250). state: dict of role -> array (x0, v1, rb_t0, rb_q0, rb_v1, rb_w1, rb_xloc)."""
    out = []
    for m in scene.meshes:
        if m.kind == "d":
            out.append(state["x0"][m.verts] + dt * state["v1"][m.verts])
        else:
            i = m.idx_in_ps
            t1 = state["rb_t0"][i] + dt * state["rb_v1"][i]
            R1 = quat_to_R(quat_time_integration(state["rb_q0"][i], state["rb_w1"][i], dt))
            out.append(state["rb_xloc"][m.verts] @ R1.T + t1)
    return out


# ---- detection --------------------------------------------------------------------------------------------------------------------------
def detect(scene, X, enlargement):
    """ProximityDetection::run (ProximityDetection.cpp:75-190) by brute force. Returns the six proximity lists as dicts of
    arrays; primitives are (mesh, local index) with local vertex connectivity."""
    M = scene.meshes
    pts = [(k, i) for k, m in enumerate(M) for i in range(len(m.verts))]
    tris = [(k, i) for k, m in enumerate(M) for i in range(len(m.tris))]
    eds = [(k, i) for k, m in enumerate(M) for i in range(len(m.edges))]
    res = {}
    enl2 = enlargement * enlargement
    # ---- point - triangle
    if pts and tris:
        pm = np.array([p[0] for p in pts]); pi = np.array([p[1] for p in pts])
        tm = np.array([t[0] for t in tris]); ti = np.array([t[1] for t in tris])
        tv = np.array([M[k].tris[i] for k, i in tris])
        P = np.array([X[k][i] for k, i in pts])
        T = np.array([[X[k][v] for v in M[k].tris[i]] for k, i in tris])
        ip, it = np.meshgrid(np.arange(len(pts)), np.arange(len(tris)), indexing="ij")
        ip, it = ip.ravel(), it.ravel()
        same = pm[ip] == tm[it]
        keep = ~(same & ((pi[ip] == tv[it, 0]) | (pi[ip] == tv[it, 1]) | (pi[ip] == tv[it, 2])))
        dis = np.array([[scene.is_disabled(a, b) for b in range(len(M))] for a in range(len(M))])
        keep &= ~dis[pm[ip], tm[it]]
        ip, it = ip[keep], it[keep]
        ty, d2 = point_triangle_sq_distance(P[ip], T[it, 0], T[it, 1], T[it, 2])
        hit = d2 < enl2
        res["pt"] = dict(pm=pm[ip][hit], pi=pi[ip][hit], tm=tm[it][hit], ti=ti[it][hit], tv=tv[it][hit], ty=ty[hit], d=np.sqrt(d2[hit]))
    # ---- edge - edge (edge_i < edge_j in the global edge order)
    if len(eds) > 1:
        em = np.array([e[0] for e in eds]); ei = np.array([e[1] for e in eds])
        ev = np.array([M[k].edges[i] for k, i in eds])
        E = np.array([[X[k][v] for v in M[k].edges[i]] for k, i in eds])
        ia, ib = np.triu_indices(len(eds), 1)
        same = em[ia] == em[ib]
        keep = ~(same & ((ev[ia, 0] == ev[ib, 0]) | (ev[ia, 0] == ev[ib, 1]) | (ev[ia, 1] == ev[ib, 0]) | (ev[ia, 1] == ev[ib, 1])))
        dis = np.array([[scene.is_disabled(a, b) for b in range(len(M))] for a in range(len(M))])
        keep &= ~dis[em[ia], em[ib]]
        ia, ib = ia[keep], ib[keep]
        cross2 = _sq(np.cross(E[ia, 1] - E[ia, 0], E[ib, 1] - E[ib, 0]))
        ok = cross2 > EE_PARALLEL_CUTOFF
        ia, ib = ia[ok], ib[ok]
        ty, d2 = edge_edge_sq_distance(E[ia, 0], E[ia, 1], E[ib, 0], E[ib, 1])
        hit = d2 < enl2
        res["ee"] = dict(am=em[ia][hit], ai=ei[ia][hit], av=ev[ia][hit], bm=em[ib][hit], bi=ei[ib][hit], bv=ev[ib][hit], ty=ty[hit], d=np.sqrt(d2[hit]))
    return res


def broad_phase(scene, X, enlargement):
    """The candidate pairs of ProximityDetection::run by brute force: boxes as AABBs.cpp:7-45 builds them (vertices cast to float, minimum /
    maximum, minus / plus `(float)enlargement + epsilon` in float arithmetic), overlap = `<=` both ways on every axis, minus points of their own
    triangle, edges sharing a vertex (same mesh) and blacklisted mesh pairs (BroadPhasePTEEBase.cpp:176-262, the Bruteforce strategy's loops).
    Returns (point_triangle, edge_edge): sorted arrays of rows (first.set, first.idx, second.set, second.idx); edge pairs with the lower
    global edge first. No fixture holds the reference's own list (nothing in stark/src reads it); it is pinned indirectly: every pair of
    the narrow-phase lists, which ARE pinned (tests/test_oracle_contact.py), must be among these candidates (tests/test_gpu_contact.py)."""
    M = scene.meshes
    extra = np.float32(np.float32(enlargement) + np.finfo(np.float32).eps)

    def boxes(prims):  # prims: list of (mesh, local vertex indices)
        lo = np.empty((len(prims), 3), dtype=np.float32)
        hi = np.empty((len(prims), 3), dtype=np.float32)
        for n, (k, vs) in enumerate(prims):
            x = np.asarray(X[k], dtype=np.float64)[list(vs)].astype(np.float32)
            lo[n] = x.min(axis=0) - extra
            hi[n] = x.max(axis=0) + extra
        return lo, hi

    pts = [(k, i) for k, m in enumerate(M) for i in range(len(m.verts))]
    tris = [(k, i) for k, m in enumerate(M) for i in range(len(m.tris))]
    eds = [(k, i) for k, m in enumerate(M) for i in range(len(m.edges))]
    pt_rows, ee_rows = [], []
    if pts and tris:
        plo, phi = boxes([(k, [i]) for k, i in pts])
        tlo, thi = boxes([(k, M[k].tris[i]) for k, i in tris])
        ov = ((plo[:, None, :] <= thi[None, :, :]) & (tlo[None, :, :] <= phi[:, None, :])).all(axis=2)
        for ip, it in zip(*np.nonzero(ov)):
            (pm, pi), (tm, ti) = pts[ip], tris[it]
            if scene.is_disabled(pm, tm) or (pm == tm and pi in M[tm].tris[ti]):
                continue
            pt_rows.append((pm, pi, tm, ti))
    if len(eds) > 1:
        elo, ehi = boxes([(k, M[k].edges[i]) for k, i in eds])
        ov = ((elo[:, None, :] <= ehi[None, :, :]) & (elo[None, :, :] <= ehi[:, None, :])).all(axis=2)
        for ia, ib in zip(*np.nonzero(np.triu(ov, 1))):
            (am, ai), (bm, bi) = eds[ia], eds[ib]
            if scene.is_disabled(am, bm) or (am == bm and set(M[am].edges[ai]) & set(M[bm].edges[bi])):
                continue
            ee_rows.append((am, ai, bm, bi))
    return (np.array(sorted(pt_rows), dtype=np.int64).reshape(-1, 4), np.array(sorted(ee_rows), dtype=np.int64).reshape(-1, 4))


def has_intersections(scene, X):
    """IntersectionDetection::run: any edge-triangle pair (not sharing a vertex, not disabled) that intersects."""
    M = scene.meshes
    for ka, ma in enumerate(M):
        for kb, mb in enumerate(M):
            if scene.is_disabled(ka, kb) or len(ma.edges) == 0 or len(mb.tris) == 0:
                continue
            ie, it = np.meshgrid(np.arange(len(ma.edges)), np.arange(len(mb.tris)), indexing="ij")
            ie, it = ie.ravel(), it.ravel()
            e, t = ma.edges[ie], mb.tris[it]
            if ka == kb:
                share = (e[:, 0, None] == t).any(1) | (e[:, 1, None] == t).any(1)
                ie, it, e, t = ie[~share], it[~share], e[~share], t[~share]
            if edge_intersects_triangle(X[ka][e[:, 0]], X[ka][e[:, 1]], X[kb][t[:, 0]], X[kb][t[:, 1]], X[kb][t[:, 2]]).any():
                return True
    return False


# ---- routing ----------------------------------------------------------------------------------------------------------------------------
def _gv(scene, mesh, local):
    """_local_to_ps_global_indices (:274-301)."""
    return int(scene.meshes[mesh].verts[local])


def _classified_pairs(scene, prox):
    """The six result lists of ProximityDetection in the reference's (first, second) roles:
    yields (family, d, A, B) with A/B = dict(mesh, verts(local), edge(local) | None)."""
    out = []
    if "pt" in prox:
        r = prox["pt"]
        for n in range(len(r["ty"])):
            ty, t = int(r["ty"][n]), r["tv"][n]
            A = dict(mesh=int(r["pm"][n]), verts=[int(r["pi"][n])])
            tm = int(r["tm"][n])
            if ty <= P_T2:
                out.append(("pt_pp", r["d"][n], A, dict(mesh=tm, verts=[int(t[ty])])))
            elif ty <= P_E2:
                k = ty - P_E0
                out.append(("pt_pe", r["d"][n], A, dict(mesh=tm, verts=[int(t[k]), int(t[(k + 1) % 3])])))
            else:
                out.append(("pt_pt", r["d"][n], A, dict(mesh=tm, verts=[int(v) for v in t])))
    if "ee" in prox:
        r = prox["ee"]
        for n in range(len(r["ty"])):
            ty = int(r["ty"][n])
            ea, eb = [int(v) for v in r["av"][n]], [int(v) for v in r["bv"][n]]
            am, bm = int(r["am"][n]), int(r["bm"][n])
            EA, EB = dict(mesh=am, verts=ea, edge=ea), dict(mesh=bm, verts=eb, edge=eb)

            def ep(mesh, edge, v):
                return dict(mesh=mesh, verts=[v], edge=edge)

            d = r["d"][n]
            if ty == EA0_EB0: out.append(("ee_pp", d, ep(am, ea, ea[0]), ep(bm, eb, eb[0])))
            elif ty == EA0_EB1: out.append(("ee_pp", d, ep(am, ea, ea[0]), ep(bm, eb, eb[1])))
            elif ty == EA1_EB0: out.append(("ee_pp", d, ep(am, ea, ea[1]), ep(bm, eb, eb[0])))
            elif ty == EA1_EB1: out.append(("ee_pp", d, ep(am, ea, ea[1]), ep(bm, eb, eb[1])))
            elif ty == EA_EB0: out.append(("ee_pe", d, ep(bm, eb, eb[0]), EA))
            elif ty == EA_EB1: out.append(("ee_pe", d, ep(bm, eb, eb[1]), EA))
            elif ty == EA0_EB: out.append(("ee_pe", d, ep(am, ea, ea[0]), EB))
            elif ty == EA1_EB: out.append(("ee_pe", d, ep(am, ea, ea[1]), EB))
            else: out.append(("ee_ee", d, EA, EB))
    return out


def contact_tables(scene, prox):
    """_before_energy_evaluation__update_contacts (:368-530): name -> int rows."""
    T = {name: [] for name in RECIPES if name.startswith("contact_")}
    M = scene.meshes
    for fam, d, A, B in _classified_pairs(scene, prox):
        ga, gb = A["mesh"], B["mesh"]
        if d > M[ga].thickness + M[gb].thickness:
            continue
        ka, kb = M[ga].kind, M[gb].kind
        va = [_gv(scene, ga, v) for v in A["verts"]]
        vb = [_gv(scene, gb, v) for v in B["verts"]]
        ea = [_gv(scene, ga, v) for v in A["edge"]] if A.get("edge") else None
        eb = [_gv(scene, gb, v) for v in B["edge"]] if B.get("edge") else None
        ia, ib = M[ga].idx_in_ps, M[gb].idx_in_ps
        if fam.startswith("pt"):
            pa, pb = va, vb
        else:  # edge-edge families list the edge first, then the point (if any)
            pa = (ea + va) if fam == "ee_pp" or (fam == "ee_pe") else ea
            pb = (eb + vb) if fam == "ee_pp" else eb
        if ka == "d" and kb == "d":
            T["contact_d_d_%s_cubic" % fam].append([ga, gb] + pa + pb)
        elif ka == "rb" and kb == "rb":
            T["contact_rb_rb_%s_cubic" % fam].append([ga, gb, ia, ib] + pa + pb)
        elif ka == "rb":
            T["contact_rb_d_%s_cubic" % fam].append([ga, gb, ia] + pa + pb)
        else:  # deformable first: the rigid side is listed first and asymmetric families switch table
            sw = {"pt_pp": "pt_pp", "pt_pe": "pt_ep", "pt_pt": "pt_tp", "ee_pp": "ee_pp", "ee_pe": "ee_ep", "ee_ee": "ee_ee"}[fam]
            T["contact_rb_d_%s_cubic" % sw].append([gb, ga, ib] + pb + pa)
    return {k: np.array(v, dtype=np.int32).reshape(len(v), RECIPES[k][0]) for k, v in T.items()}


def friction_tables(scene, prox, X, k_stiffness):
    """_before_time_step__update_friction_contacts (:531-773): name -> (conn rows, dict(T, mu, fn[, bary]))."""
    M = scene.meshes
    out = {name: dict(conn=[], T=[], mu=[], fn=[], bary=[]) for name in RECIPES if name.startswith("friction_")}
    for fam, d, A, B in _classified_pairs(scene, prox):
        ga, gb = A["mesh"], B["mesh"]
        dhat = M[ga].thickness + M[gb].thickness
        if d > dhat:
            continue
        mu = scene.mu(ga, gb)
        if mu == 0.0:
            continue
        ka, kb = M[ga].kind, M[gb].kind
        va = [_gv(scene, ga, v) for v in A["verts"]]
        vb = [_gv(scene, gb, v) for v in B["verts"]]
        ia, ib = M[ga].idx_in_ps, M[gb].idx_in_ps
        xa = [X[ga][v] for v in A["verts"]]
        xb = [X[gb][v] for v in B["verts"]]
        kind = {"pt_pp": "pp", "pt_pe": "pe", "pt_pt": "pt", "ee_pp": "pp", "ee_pe": "pe", "ee_ee": "ee"}[fam]
        if kind == "pp":
            bary, Tm = None, projection_matrix_point_point(xa[0], xb[0])
        elif kind == "pe":
            bary, Tm = barycentric_point_edge(xa[0], xb[0], xb[1]), projection_matrix_point_edge(xa[0], xb[0], xb[1])
        elif kind == "pt":
            bary, Tm = barycentric_point_triangle(xa[0], xb[0], xb[1], xb[2]), projection_matrix_triangle(xb[0], xb[1], xb[2])
        else:
            bary, Tm = barycentric_edge_edge(xa[0], xa[1], xb[0], xb[1]), projection_matrix_edge_edge(xa[0], xa[1], xb[0], xb[1])
        if ka == "d" and kb == "d":
            name, row = "friction_d_d_%s_C0" % kind, va + vb
        elif ka == "rb" and kb == "rb":
            name, row = "friction_rb_rb_%s_C0" % kind, [ia, ib] + va + vb
        elif ka == "rb":
            name, row = "friction_rb_d_%s_C0" % kind, [ia] + va + vb
        else:
            sw = {"pp": "pp", "pe": "ep", "pt": "tp", "ee": "ee"}[kind]
            name, row = "friction_rb_d_%s_C0" % sw, [ib] + vb + va
        t = out[name]
        t["conn"].append([len(t["conn"])] + row)
        t["T"].append(Tm)
        t["mu"].append(mu)
        t["fn"].append(k_stiffness * (dhat - d) ** 2)   # _barrier_force, cubic barrier (:1238-1242)
        if bary is not None:
            t["bary"].append(bary)
    res = {}
    for name, t in out.items():
        n = len(t["conn"])
        data = dict(T=np.array(t["T"]).reshape(n, 6), mu=np.array(t["mu"]), fn=np.array(t["fn"]))
        if t["bary"]:
            data["bary"] = np.array(t["bary"])
        res[name] = (np.array(t["conn"], dtype=np.int32).reshape(n, RECIPES[name][0]), data)
    return res


def max_thickness(scene):
    return max(m.thickness for m in scene.meshes)

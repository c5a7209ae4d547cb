"""GPU parity of the device contact detector (stark_amd/csrc/contact.hip) against the reference's own contact and friction
tables recorded in the contact fixtures, and of the full evaluation through the module-registered potentials."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import contact as oc  # noqa: E402
from oracle import evaluator as ev  # noqa: E402
from contact_util import friction_order, sorted_rows, state_from_fixture  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
CONTACT_FIXTURES = ["contactmix_t0", "contactmix_t1", "contactcorners_t0", "contactrods_t0", "contactrods_t1"]


def is_contact(name):
    return name.startswith("contact_") or name.startswith("friction_")


def build(name):
    """Engine with the fixture's non-contact potentials registered from the fixture and the 35 contact/friction potentials
    registered by the device contact module; collision meshes from the fixture's detector inputs."""
    import copy

    from gpu_util import engine_from_problem

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    base = copy.copy(prob)
    base.potentials = [copy.copy(p) for p in prob.potentials]
    for p in base.potentials:
        if is_contact(p.name):
            p.conn = p.conn[:0]   # (skipped by engine_from_problem)
    eng = engine_from_problem(base, man)
    st, roles = state_from_fixture(prob, man)
    ids = {}
    for role, aid in roles.items():
        stride = {"rb_q0": 4}.get(role, 3 if role in ("v1", "x0", "X", "rb_xloc", "rb_v1", "rb_w1", "rb_t0") else 1)
        key = (aid, stride)
        if key not in eng.array_ids:
            eng.array_ids[key] = eng.array(eng.host_arrays[aid].reshape(-1, stride), stride)
        ids["thickness" if role == "thick" else role] = eng.array_ids[key]
    eng.contact_init(**ids)
    scene = oc.scene_from_fixture(man, z)
    for m in scene.meshes:
        eng.contact_add_mesh(m.kind, m.idx_in_ps, m.verts, m.tris, m.edges)
    for (a, b), mu in scene.friction.items():
        eng.contact_set_friction(a, b, mu)
    return eng, prob, man, z, scene, st


@pytest.mark.parametrize("broad_phase", ["sweep_and_prune", "all_pairs", "own_sorts"])
@pytest.mark.parametrize("name", CONTACT_FIXTURES)
def test_device_tables_match_reference(name, broad_phase):
    """own_sorts: option seg_sort — the box list through k_bp_hist / k_bp_scatter / k_seg_sort and the contact keys through k_rank_sort_keys instead
    of the sorting library (measured slower, off by default: a second implementation of the sorted lists the sweep walks)"""
    from stark_amd import capi

    eng, prob, man, z, scene, st = build(name)
    eng.contact_set_broad_phase(broad_phase == "all_pairs")
    if broad_phase == "own_sorts":
        assert capi.lib().mistark_set_option(eng.h, b"seg_sort", 1) == 0
    dt = float(np.asarray(st["dt"]).ravel()[0])
    n_fr = eng.contact_update_friction()
    n_ct = eng.contact_update(dt)
    # collision vertices
    X = np.concatenate(oc.mesh_vertices(scene, st, dt))
    assert np.abs(eng.contact_vertices() - X).max() <= 1e-14 * max(1.0, np.abs(X).max())
    assert eng.contact_count_intersections(dt) == 0
    tot_c = tot_f = 0
    for pi, p in enumerate(man["potentials"]):
        if not is_contact(p["name"]):
            continue
        ref = prob.potentials[pi].conn
        ours = eng.contact_table(p["name"])
        assert ours.shape == ref.shape, (p["name"], ours.shape, ref.shape)
        if p["name"].startswith("contact_"):
            # bit-exact contact-pair indexing: the same rows in every barrier table (row order: ours is sorted by pair id)
            assert (sorted_rows(ours) == sorted_rows(ref)).all(), p["name"]
            tot_c += len(ref)
            continue
        tot_f += len(ref)
        if len(ref) == 0:
            continue
        assert (ours[:, 0] == np.arange(len(ours))).all()
        rec = oc.RECIPES[p["name"]][1]
        ref_data = {}
        for (role, stride, _), b in zip(rec, p["bindings"]):
            if role in ("T", "mu", "fn", "bary"):
                ref_data[role] = np.asarray(prob.arrays[b["array"]]).reshape(-1, stride)
        data = eng.contact_friction_data(p["name"], len(ours))
        data = {k: v.reshape(len(ours), -1) for k, v in data.items() if k in ref_data}
        o1, o2 = friction_order(ours, data), friction_order(ref, ref_data)
        assert (ours[o1, 1:] == ref[o2, 1:]).all(), p["name"]
        for role, rd in ref_data.items():
            scale = max(np.abs(rd).max(), 1e-300)
            assert np.abs(data[role][ours[o1, 0]] - rd[ref[o2, 0]]).max() <= 1e-9 * scale, (p["name"], role)
    assert tot_c == n_ct and tot_f == n_fr and tot_c > 0 and tot_f > 0

    # the whole evaluation through the device-written tables: energy, gradient, assembled matrix, linear solve
    E, grad = eng.eval(capi.EVAL_P_G_H)
    scale = sum(abs(p.get("E", 0.0)) for p in man["potentials"])
    assert abs(E - man["E"]) <= 1e-11 * max(1.0, scale)
    assert np.abs(grad - z["grad"]).max() <= 1e-9 * np.abs(z["grad"]).max()
    eng.assemble()
    y = eng.spmv(np.sin(0.37 * np.arange(eng.ndofs)))
    assert np.abs(y - z["spmv_y"]).max() <= 2e-6 * np.abs(z["spmv_y"]).max()
    # a second detection at the same state leaves the tables (and the dynamic matrix pattern) alone
    assert eng.contact_update(dt) == n_ct
    E2, _ = eng.eval(capi.EVAL_P)
    assert abs(E2 - E) <= 1e-13 * max(1.0, abs(E))
    eng.close()


def test_detects_intersections_and_moving_contacts():
    """Moves the cloth of the contact zoo through the fixed box: contact sets change with the DoFs, the intersection test fires."""
    from stark_amd import capi

    eng, prob, man, z, scene, st = build("contactmix_t0")
    dt = float(np.asarray(st["dt"]).ravel()[0])
    eng.contact_update_friction()
    n0 = eng.contact_update(dt)
    u = eng.get_dofs()
    # push every deformable node down by 5 cm within one step
    soft = man["dof_sets"][0]
    u2 = u.copy()
    u2[soft["offset"] + 2:soft["offset"] + soft["size"]:3] -= 0.05 / dt
    eng.set_dofs(u2)
    n1 = eng.contact_update(dt)
    st2 = dict(st)
    st2["v1"] = u2[soft["offset"]:soft["offset"] + soft["size"]].reshape(-1, 3)
    X = oc.mesh_vertices(scene, st2, dt)
    prox = oc.detect(scene, X, 2.0 * oc.max_thickness(scene))
    tables = oc.contact_tables(scene, prox)
    assert n1 == sum(len(t) for t in tables.values()) and n1 != n0
    for name, t in tables.items():
        assert (sorted_rows(eng.contact_table(name)) == sorted_rows(t)).all(), name
    assert oc.has_intersections(scene, X)
    assert eng.contact_count_intersections(dt) > 0
    # evaluation still works with the new tables (energies of penetrated pairs are just large)
    E, g = eng.eval(capi.EVAL_P_G_H)
    assert np.isfinite(E)
    eng.assemble()
    eng.close()


@pytest.mark.parametrize("cap", [64, 200])
def test_key_list_and_padded_sort_grow_on_demand(cap, monkeypatch):
    """A key list that starts far too short (64 or 200 entries for 297 + 292 contacts) and a padded sort that therefore cannot hold the
    first search: the searches repeat with grown buffers and end with the reference's tables."""
    monkeypatch.setenv("MISTARK_CONTACT_KEY_CAP", str(cap))
    eng, prob, man, z, scene, st = build("contactmix_t0")
    dt = float(np.asarray(st["dt"]).ravel()[0])
    eng.contact_update_friction()
    eng.contact_update(dt)
    for pi, p in enumerate(man["potentials"]):
        if not is_contact(p["name"]) or not p["name"].startswith("contact_"):
            continue
        ref = prob.potentials[pi].conn
        ours = eng.contact_table(p["name"])
        assert ours.shape == ref.shape and (sorted_rows(ours) == sorted_rows(ref)).all(), p["name"]
    eng.close()


def _oracle_lists(prox):
    """oracle/contact.py: detect()'s classified pairs as the rows of include/mistark_tmcd.h (ProximityDetection.cpp:113-129, :166-186)."""
    out = {k: [] for k in ("pt_point_point", "pt_point_edge", "pt_point_triangle", "ee_point_point", "ee_point_edge", "ee_edge_edge")}
    dist = {k: [] for k in out}
    pt = prox.get("pt")
    if pt is not None:
        for k in range(len(pt["ty"])):
            ty, tv = int(pt["ty"][k]), [int(v) for v in pt["tv"][k]]
            head = [int(pt["pm"][k]), int(pt["pi"][k]), int(pt["tm"][k]), int(pt["ti"][k])] + tv
            if ty <= 2:
                name, row = "pt_point_point", head + [tv[ty]]
            elif ty <= 5:
                name, row = "pt_point_edge", head + [tv[ty - 3], tv[(ty - 2) % 3]]
            else:
                name, row = "pt_point_triangle", head
            out[name].append(row)
            dist[name].append(float(pt["d"][k]))
    ee = prox.get("ee")
    if ee is not None:
        for k in range(len(ee["ty"])):
            ty = int(ee["ty"][k])
            A = [int(ee["am"][k]), int(ee["ai"][k]), int(ee["av"][k][0]), int(ee["av"][k][1])]
            B = [int(ee["bm"][k]), int(ee["bi"][k]), int(ee["bv"][k][0]), int(ee["bv"][k][1])]
            if ty <= 3:     # EA0_EB0, EA0_EB1, EA1_EB0, EA1_EB1
                name, row = "ee_point_point", A + [A[2 + ty // 2]] + B + [B[2 + ty % 2]]
            elif ty <= 5:   # EA_EB0, EA_EB1: the point lies on edge b and comes first
                name, row = "ee_point_edge", B + [B[2 + ty - 4]] + A
            elif ty <= 7:   # EA0_EB, EA1_EB
                name, row = "ee_point_edge", A + [A[2 + ty - 6]] + B
            else:
                name, row = "ee_edge_edge", A + B
            out[name].append(row)
            dist[name].append(float(ee["d"][k]))
    return out, dist


@pytest.mark.parametrize("name", ["contactmix_t0", "contactmix_t1", "contactrods_t0", "contactcorners_t0"])
def test_standalone_detector_returns_the_references_lists(name):
    """include/mistark_tmcd.h (the detector behind the reference's own tmcd::ProximityDetection / IntersectionDetection interface, host
    positions in, result lists out) against the oracle's brute-force restatement of ProximityDetection::run, which tests/test_oracle_contact.py
    pins to the reference's tables: the six lists as row sets, bit-exact, with their distances."""
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    st, _ = state_from_fixture(prob, man)
    scene = oc.scene_from_fixture(man, z)
    dt = float(np.asarray(st["dt"]).ravel()[0])
    X = [np.ascontiguousarray(x, dtype=np.float64) for x in oc.mesh_vertices(scene, st, dt)]
    cd = capi.CollisionDetector()
    for m, x in zip(scene.meshes, X):
        cd.add_mesh(x, m.tris, m.edges)
    for (a, b) in scene.disabled:
        cd.add_blacklist(a, b)
    enl = 2.0 * oc.max_thickness(scene)
    got = cd.run_proximity(enl)
    ref, ref_d = _oracle_lists(oc.detect(scene, X, enl))
    total = 0
    for lname, (rows, d) in got.items():
        r = np.array(ref[lname], dtype=np.int64).reshape(-1, rows.shape[1])
        assert rows.shape == r.shape, (lname, rows.shape, r.shape)
        if len(r) == 0:
            continue
        o1 = np.lexsort(rows.T[::-1])
        o2 = np.lexsort(r.T[::-1])
        assert (rows[o1] == r[o2]).all(), lname
        assert np.abs(d[o1] - np.array(ref_d[lname])[o2]).max() <= 1e-15 * max(1.0, d.max()), lname
        total += len(r)
    assert total > 10
    assert len(cd.run_intersection()) == 0
    # move one mesh through the others (positions are re-read at every run): intersections appear, and a second proximity run follows the move
    for x in X:
        if len(x) > 8:
            x[:, 2] -= 0.05
            break
    hits = cd.run_intersection()
    assert oc.has_intersections(scene, X) == (len(hits) > 0)
    got2 = cd.run_proximity(enl)
    ref2, _ = _oracle_lists(oc.detect(scene, X, enl))
    for lname, (rows, _) in got2.items():
        r = np.array(ref2[lname], dtype=np.int64).reshape(-1, rows.shape[1])
        assert rows.shape == r.shape, lname
        if len(r):
            assert (rows[np.lexsort(rows.T[::-1])] == r[np.lexsort(r.T[::-1])]).all(), lname
    if len(hits):
        e, t = hits[0, :4], hits[0, 4:]
        assert oc.edge_intersects_triangle(X[e[0]][e[2]][None], X[e[0]][e[3]][None], X[t[0]][t[2]][None], X[t[0]][t[3]][None], X[t[0]][t[4]][None]).all()
    cd.close()


@pytest.mark.parametrize("name", ["contactmix_t0", "contactmix_t1", "contactrods_t0", "contactcorners_t0"])
def test_standalone_detector_broad_phase_list(name):
    """mistark_cd_run_broad_phase (what tmcd::ProximityDetection::get_broad_phase_results() hands out, ProximityDetection.h:31): the candidate pairs
    as row sets against the oracle's brute force over boxes built the way the reference builds them (oracle/contact.py broad_phase) — bit-exact;
    every pair of the pinned narrow-phase lists is among them; and a proximity run behind it still returns its lists (the listing invalidates the
    detector's cache of the previous run)."""
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    st, _ = state_from_fixture(prob, man)
    scene = oc.scene_from_fixture(man, z)
    dt = float(np.asarray(st["dt"]).ravel()[0])
    X = [np.ascontiguousarray(x, dtype=np.float64) for x in oc.mesh_vertices(scene, st, dt)]
    cd = capi.CollisionDetector()
    for m, x in zip(scene.meshes, X):
        cd.add_mesh(x, m.tris, m.edges)
    for (a, b) in scene.disabled:
        cd.add_blacklist(a, b)
    enl = 2.0 * oc.max_thickness(scene)
    before = cd.run_proximity(enl)
    pt, ee = cd.run_broad_phase(enl)
    rpt, ree = oc.broad_phase(scene, X, enl)
    for got, ref in ((pt, rpt), (ee, ree)):
        assert got.shape == ref.shape, (got.shape, ref.shape)
        if len(ref):
            assert (got[np.lexsort(got.T[::-1])] == ref).all()
    assert len(rpt) + len(ree) > 20
    # the narrow phase only sees broad-phase candidates: every proximity pair is in the list
    spt = {tuple(r) for r in pt.tolist()}
    see = {tuple(r) for r in ee.tolist()}
    n_checked = 0
    for lname, (rows, _) in before.items():
        for r in rows.tolist():
            if lname.startswith("pt_"):
                assert (r[0], r[1], r[2], r[3]) in spt, (lname, r)
            else:
                second = r[5:7] if lname != "ee_edge_edge" else r[4:6]
                a, b = (r[0], r[1]), tuple(second)
                assert (a + b) in see or (b + a) in see, (lname, r)
            n_checked += 1
    assert n_checked > 10
    after = cd.run_proximity(enl)
    for lname in before:
        assert (before[lname][0] == after[lname][0]).all() and (before[lname][1] == after[lname][1]).all(), lname
    # only one family active: the other list stays empty
    cd.activate(point_triangle=True, edge_edge=False)
    pt2, ee2 = cd.run_broad_phase(enl)
    assert len(ee2) == 0 and pt2.shape == pt.shape
    cd.close()


def test_stand_in_get_broad_phase_results_equals_the_references_detector():
    """tmcd::ProximityDetection::get_broad_phase_results() of the stand-in header (shim/include_cd, on mistark_cd_run_broad_phase) against the
    reference's own detector: the same program (`shim_check tmcd_broad`) built once against each header; the reference's output is the
    committed fixture, the stand-in's comes from oracle/_ref/shim_check_cd on this GPU. And the same geometry straight through the C ABI."""
    import json
    import subprocess

    from contact_util import tmcd_broad_scene
    from stark_amd import capi

    z = np.load(os.path.join(GOLDEN, "tmcd_broad_phase_listing.npz"))
    ref = json.loads(bytes(z["listing_json"]).decode())
    scene, X, enl = tmcd_broad_scene()
    cd = capi.CollisionDetector()
    for m, x in zip(scene.meshes, X):
        cd.add_mesh(x, m.tris, m.edges)
    cd.add_blacklist(1, 1)
    pt, ee = cd.run_broad_phase(enl)
    cd.close()
    assert pt[np.lexsort(pt.T[::-1])].tolist() == ref["point_triangle"]
    assert ee[np.lexsort(ee.T[::-1])].tolist() == ref["edge_edge"]
    exe = os.path.join(ROOT, "oracle", "_ref", "shim_check_cd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_check_cd not built (needs /root/reference in the build container)")
    r = subprocess.run([exe, "tmcd_broad"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    got = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert got["narrow_pairs"] == ref["narrow_pairs"] > 0
    assert got["point_triangle"] == ref["point_triangle"] and got["edge_edge"] == ref["edge_edge"]


def test_standalone_detector_range_blacklists():
    """tmcd::ProximityDetection::add_blacklist_range_point_triangle / _edge_edge (ProximityDetection.h:24-25; BroadPhasePTEEBase.cpp:19-41,176-262:
    half-open intervals of local primitive indices, a pair is dropped when its triangle lies in the first and its point in the second interval —
    for edges: the lower edge in the first, the higher in the second): the lists with ranges registered are the lists without them minus
    exactly the pairs inside a range (filtered on the host from the unrestricted run), for point-triangle and edge-edge ranges between two
    meshes and inside one mesh."""
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "contactmix_t0.npz"))
    st, _ = state_from_fixture(prob, man)
    scene = oc.scene_from_fixture(man, z)
    dt = float(np.asarray(st["dt"]).ravel()[0])
    X = [np.ascontiguousarray(x, dtype=np.float64) for x in oc.mesh_vertices(scene, st, dt)]
    enl = 2.0 * oc.max_thickness(scene)

    def detector():
        cd = capi.CollisionDetector()
        for m, x in zip(scene.meshes, X):
            cd.add_mesh(x, m.tris, m.edges)
        for (a, b) in scene.disabled:
            cd.add_blacklist(a, b)
        return cd

    cd = detector()
    full = cd.run_proximity(enl)
    cd.close()
    # ranges chosen from the pairs that exist: half of the points / edges of the busiest mesh pairs
    pt = np.concatenate([full[n][0][:, [0, 1, 2, 3]] for n in ("pt_point_point", "pt_point_edge", "pt_point_triangle") if len(full[n][0])])
    ee = np.concatenate([full[n][0][:, [0, 1, 5 if n != "ee_edge_edge" else 4, 6 if n != "ee_edge_edge" else 5]] for n in ("ee_point_point", "ee_point_edge", "ee_edge_edge")
                         if len(full[n][0])])
    assert len(pt) > 5 and len(ee) > 5
    pm, tm = [int(v) for v in pt[np.argmax(np.bincount(pt[:, 0] * 64 + pt[:, 2]) [pt[:, 0] * 64 + pt[:, 2]])][[0, 2]]]
    p_mid = int(np.median(pt[(pt[:, 0] == pm) & (pt[:, 2] == tm), 1])) + 1
    r_pt = (pm, (0, p_mid), tm, (0, len(scene.meshes[tm].tris)))
    # edge pairs come as (lower global edge, higher global edge): take the busiest (first mesh, second mesh) with first <= second
    swap = ee[:, 0] > ee[:, 2]
    ee = np.where(swap[:, None], ee[:, [2, 3, 0, 1]], ee)
    key = ee[:, 0] * 64 + ee[:, 2]
    ea, eb = divmod(int(np.bincount(key).argmax()), 64)
    e_mid = int(np.median(ee[(ee[:, 0] == ea) & (ee[:, 2] == eb), 1])) + 1
    r_ee = (ea, (0, e_mid), eb, (e_mid if ea == eb else 0, len(scene.meshes[eb].edges)))
    cd = detector()
    cd.add_blacklist_range(False, r_pt[0], r_pt[1], r_pt[2], r_pt[3])
    cd.add_blacklist_range(True, r_ee[0], r_ee[1], r_ee[2], r_ee[3])
    got = cd.run_proximity(enl)
    cd.close()
    dropped = 0
    for name, (rows, d) in full.items():
        if name.startswith("pt_"):
            out = (rows[:, 0] == r_pt[0]) & (rows[:, 1] >= r_pt[1][0]) & (rows[:, 1] < r_pt[1][1]) & (rows[:, 2] == r_pt[2]) & (rows[:, 3] >= r_pt[3][0]) & (rows[:, 3] < r_pt[3][1])
        else:
            # (a row lists its two edges by ROLE — the one holding the closest point first —, the blacklist looks the pair up as (lower, higher)
            # edge of the detector's global numbering = meshes in registration order)
            j = 4 if name == "ee_edge_edge" else 5
            e_off = np.concatenate([[0], np.cumsum([len(m.edges) for m in scene.meshes])])
            ga, gb = e_off[rows[:, 0]] + rows[:, 1], e_off[rows[:, j]] + rows[:, j + 1]
            lo, hi = np.minimum(ga, gb), np.maximum(ga, gb)
            out = (lo >= e_off[r_ee[0]] + r_ee[1][0]) & (lo < e_off[r_ee[0]] + r_ee[1][1]) & (hi >= e_off[r_ee[2]] + r_ee[3][0]) & (hi < e_off[r_ee[2]] + r_ee[3][1])
        want = rows[~out]
        have = got[name][0]
        dropped += int(out.sum())
        assert have.shape == want.shape, (name, have.shape, want.shape)
        if len(want):
            assert (have[np.lexsort(have.T[::-1])] == want[np.lexsort(want.T[::-1])]).all(), name
    assert dropped > 2

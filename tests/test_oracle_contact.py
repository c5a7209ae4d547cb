"""Pins oracle/contact.py (detection, classification, table construction, friction geometry, binding recipes) against the
reference: known-answer vectors produced by the reference's own functions and the contact/friction tables the reference
built for the contact fixtures (tests/golden/make_fixtures.py)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import contact as oc  # noqa: E402
from oracle import evaluator as ev  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from contact_util import roles_from_manifest, sorted_rows, state_from_fixture  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CONTACT_FIXTURES = ["contactmix_t0", "contactmix_t1", "contactcorners_t0", "contactrods_t0", "contactrods_t1"]


def test_narrow_phase_known_answers():
    z = np.load(os.path.join(GOLDEN, "contact_geometry.npz"))
    x = z["pt_in"]
    ty, d2 = oc.point_triangle_sq_distance(x[:, 0:3], x[:, 3:6], x[:, 6:9], x[:, 9:12])
    assert (ty == z["pt_type"]).all()
    assert np.abs(d2 - z["pt_d2"]).max() <= 1e-13 * max(1.0, z["pt_d2"].max())
    assert len(set(ty)) == 7  # every region occurs
    y = z["ee_in"]
    cross2 = ((np.cross(y[:, 3:6] - y[:, 0:3], y[:, 9:12] - y[:, 6:9])) ** 2).sum(1)
    ok = cross2 >= 1e-30  # (parallel pairs never reach a table: ProximityDetection.cpp:152-155)
    ty, d2 = oc.edge_edge_sq_distance(y[:, 0:3], y[:, 3:6], y[:, 6:9], y[:, 9:12])
    assert ok.sum() > 500
    assert (ty[ok] == z["ee_type"][ok]).all()
    assert np.abs(d2[ok] - z["ee_d2"][ok]).max() <= 1e-12 * max(1.0, z["ee_d2"].max())
    assert len(set(ty[ok])) == 9
    w = z["et_in"]
    hit = oc.edge_intersects_triangle(w[:, 0:3], w[:, 3:6], w[:, 6:9], w[:, 9:12], w[:, 12:15])
    assert (hit.astype(np.int32) == z["et_hit"]).all()
    assert 0 < hit.sum() < len(hit)


def test_friction_geometry_known_answers():
    z = np.load(os.path.join(GOLDEN, "contact_geometry.npz"))
    x, y = z["pt_in"], z["ee_in"]
    p, a, b, c = x[:, 0:3], x[:, 3:6], x[:, 6:9], x[:, 9:12]
    tol = 1e-12
    assert np.abs(oc.barycentric_point_triangle(p, a, b, c) - z["fr_pt"][:, :3]).max() < 1e-10
    assert np.abs(oc.projection_matrix_triangle(a, b, c) - z["fr_pt"][:, 3:]).max() < tol
    assert np.abs(oc.barycentric_point_edge(p, a, b) - z["fr_pe"][:, :2]).max() < 1e-11
    assert np.abs(oc.projection_matrix_point_edge(p, a, b) - z["fr_pe"][:, 2:]).max() < tol
    assert np.abs(oc.projection_matrix_point_point(p, a) - z["fr_pp"]).max() < tol
    bee = oc.barycentric_edge_edge(y[:, 0:3], y[:, 3:6], y[:, 6:9], y[:, 9:12])
    assert np.abs(bee - z["fr_ee"][:, :2]).max() <= 1e-9 * np.abs(z["fr_ee"][:, :2]).max()
    assert np.abs(oc.projection_matrix_edge_edge(y[:, 0:3], y[:, 3:6], y[:, 6:9], y[:, 9:12]) - z["fr_ee"][:, 2:]).max() < 1e-9


def load(name):
    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    return prob, man, z


@pytest.mark.parametrize("name", CONTACT_FIXTURES)
def test_recipes_match_reference_bindings(name):
    _, man, _ = load(name)
    roles = roles_from_manifest(man)
    assert {"v1", "x0", "dt", "k", "thick"} <= set(roles) or name == "contactcorners_t0"


@pytest.mark.parametrize("name", CONTACT_FIXTURES)
def test_contact_tables_match_reference(name):
    """bit-exact contact-pair indexing: the same rows (as sets) in each of the 21 barrier tables."""
    prob, man, z = load(name)
    scene = oc.scene_from_fixture(man, z)
    st, _ = state_from_fixture(prob, man)
    dt = float(np.asarray(st["dt"]).ravel()[0])
    X = oc.mesh_vertices(scene, st, dt)
    prox = oc.detect(scene, X, 2.0 * oc.max_thickness(scene))
    tables = oc.contact_tables(scene, prox)
    total = 0
    for pi, p in enumerate(man["potentials"]):
        if not p["name"].startswith("contact_"):
            continue
        ref = prob.potentials[pi].conn
        ours = tables[p["name"]]
        assert ours.shape == ref.shape, (p["name"], ours.shape, ref.shape)
        assert (sorted_rows(ours) == sorted_rows(ref)).all(), p["name"]
        total += ref.shape[0]
    assert total > 0
    assert not oc.has_intersections(scene, X)


@pytest.mark.parametrize("name", CONTACT_FIXTURES)
def test_friction_tables_match_reference(name):
    prob, man, z = load(name)
    scene = oc.scene_from_fixture(man, z)
    st, roles = state_from_fixture(prob, man)
    X0 = oc.mesh_vertices(scene, st, 0.0)
    prox = oc.detect(scene, X0, 2.0 * oc.max_thickness(scene))
    k = float(np.asarray(st["k"]).ravel()[0])
    tables = oc.friction_tables(scene, prox, X0, k)
    total = 0
    for pi, p in enumerate(man["potentials"]):
        if not p["name"].startswith("friction_"):
            continue
        ref = prob.potentials[pi].conn
        conn, data = tables[p["name"]]
        assert conn.shape == ref.shape, (p["name"], conn.shape, ref.shape)
        if ref.shape[0] == 0:
            continue
        total += ref.shape[0]
        # rows without the running index, each with its data, compared in a canonical order
        rec = oc.RECIPES[p["name"]][1]
        ref_data = {}
        for (role, stride, _), b in zip(rec, p["bindings"]):
            if role in ("T", "mu", "fn", "bary"):
                ref_data[role] = np.asarray(prob.arrays[b["array"]]).reshape(-1, stride)
        # the same (p, q) pair can occur twice (once from a point-triangle and once from an edge-edge candidate, with the
        # roles of A and B exchanged and therefore another tangent basis): compare rows WITH their data as multisets
        def keyed(c, d):
            cols = [c[:, 1:].astype(np.float64)] + [np.round(d[r].reshape(len(c), -1)[c[:, 0]], 7) for r in sorted(d)]
            m = np.concatenate(cols, axis=1)
            return np.lexsort(m.T[::-1])
        o1, o2 = keyed(conn, data if "bary" in ref_data or "bary" not in data else {k: v for k, v in data.items() if k != "bary"}), keyed(ref, ref_data)
        assert (conn[o1, 1:] == ref[o2, 1:]).all(), p["name"]
        for role, rd in ref_data.items():
            od = data[role].reshape(len(o1), -1)[conn[o1, 0]]
            scale = max(np.abs(rd).max(), 1e-300)
            assert np.abs(od - rd[ref[o2, 0]]).max() <= 1e-9 * scale, (p["name"], role)
    assert total > 0


def test_broad_phase_restatement_equals_the_references_own_candidate_list():
    """oracle/contact.py broad_phase (boxes as AABBs.cpp builds them, overlap and exclusions of BroadPhasePTEEBase.cpp) against
    tmcd::ProximityDetection::get_broad_phase_results() of the reference's own detector on the same geometry (fixture tmcd_broad_phase_listing,
    made by `oracle/_ref/shim_check tmcd_broad`): the same pairs, row for row."""
    import json

    from contact_util import tmcd_broad_scene

    z = np.load(os.path.join(GOLDEN, "tmcd_broad_phase_listing.npz"))
    ref = json.loads(bytes(z["listing_json"]).decode())
    scene, X, enl = tmcd_broad_scene()
    pt, ee = oc.broad_phase(scene, X, enl)
    assert ref["narrow_pairs"] > 0 and len(ref["point_triangle"]) > 50 and len(ref["edge_edge"]) > 100
    assert pt.tolist() == ref["point_triangle"]
    assert ee.tolist() == ref["edge_edge"]

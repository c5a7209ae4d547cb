"""Shared helpers of the contact tests: array roles of a contact fixture, canonical row order."""
import numpy as np

from oracle import contact as oc


def roles_from_manifest(man):
    """role -> fixture array id, from every contact/friction potential that has elements; also checks that the oracle's
    recipes reproduce the reference's binding lists (stride and connectivity column of every mws.make_* call) for all 35."""
    roles = {}
    seen = 0
    for p in man["potentials"]:
        if p["name"] not in oc.RECIPES:
            continue
        seen += 1
        stride, rec = oc.RECIPES[p["name"]]
        assert p["conn_stride"] == stride, p["name"]
        assert [(b["stride"], b["conn"]) for b in p["bindings"]] == [(s, c) for _, s, c in rec], p["name"]
        for (role, _, _), b in zip(rec, p["bindings"]):
            if b["array"] >= 0 and role not in ("T", "mu", "fn", "bary"):
                assert roles.setdefault(role, b["array"]) == b["array"], (p["name"], role)
    assert seen == 35
    return roles


def state_from_fixture(prob, man):
    roles = roles_from_manifest(man)
    st = {r: np.asarray(prob.arrays[i]) for r, i in roles.items()}
    for r in ("x0", "v1", "X", "rb_xloc", "rb_v1", "rb_w1", "rb_t0"):
        if r in st:
            st[r] = st[r].reshape(-1, 3)
    if "rb_q0" in st:
        st["rb_q0"] = st["rb_q0"].reshape(-1, 4)
    return st, roles


def sorted_rows(a):
    a = np.asarray(a)
    if a.shape[0] == 0:
        return a
    return a[np.lexsort(a.T[::-1])]


def friction_order(conn, data):
    """Canonical order of friction rows WITH their data (the same pair can occur twice with another tangent basis)."""
    cols = [conn[:, 1:].astype(np.float64)] + [np.round(data[r].reshape(-1, data[r].shape[-1] if data[r].ndim > 1 else 1)[conn[:, 0]], 7) for r in sorted(data)]
    return np.lexsort(np.concatenate(cols, axis=1).T[::-1])

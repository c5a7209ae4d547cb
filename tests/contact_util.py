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


def tmcd_broad_scene():
    """The geometry of `shim_check tmcd_broad` (tests/shim/shim_check.cpp): a tilted 7 x 7 cloth patch over a box, as an oracle scene + positions."""
    import numpy as np

    from oracle import contact as oc

    n = 6
    cv = np.array([[-0.3 + 0.1 * i, -0.3 + 0.1 * j + 0.013 * i, 0.002 + 0.003 * i + 0.0011 * j * j] for j in range(n + 1) for i in range(n + 1)])
    ct = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            b, c = a + 1, a + n + 1
            d = c + 1
            ct += [[a, b, d], [a, d, c]]
    bv = np.array([[0.5 if k & 1 else -0.5, 0.5 if k & 2 else -0.5, 0.0 if k & 4 else -0.2] for k in range(8)])
    bt = []
    for q in [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]]:
        bt += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]

    def edges(T):
        return np.array(sorted({(min(t[k], t[(k + 1) % 3]), max(t[k], t[(k + 1) % 3])) for t in T for k in range(3)}), dtype=np.int32)

    class Scene:
        meshes = [oc.Mesh("d", 0, np.arange(len(cv)), np.array(ct, dtype=np.int32), edges(ct), 0.002),
                  oc.Mesh("rb", 0, np.arange(8), np.array(bt, dtype=np.int32), edges(bt), 0.002)]
        disabled = {(1, 1)}

        def is_disabled(self, a, b):
            return (min(a, b), max(a, b)) in self.disabled

    return Scene(), [np.ascontiguousarray(cv), np.ascontiguousarray(bv)], 0.004

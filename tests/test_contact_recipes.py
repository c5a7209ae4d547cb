"""Host-side check (no GPU): the binding recipes the device contact module registers its 35 potentials with
(mistark_contact_recipe, stark_amd/csrc/contact.hip) equal the reference's binding lists recorded in the contact fixtures."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import contact as oc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
ROLE_NAMES = ["v1", "x0", "X", "dt", "k", "thick", "epsv", "rb_xloc", "rb_v1", "rb_w1", "rb_t0", "rb_q0", "T", "mu", "fn", "bary"]


@pytest.mark.parametrize("fixture", ["contactmix_t0", "contactcorners_t0"])
def test_module_recipes_match_reference(fixture):
    from stark_amd import capi

    L = capi.lib()
    z = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    man = json.loads(bytes(z["manifest_json"]).decode())
    seen = 0
    for p in man["potentials"]:
        if not (p["name"].startswith("contact_") or p["name"].startswith("friction_")):
            continue
        seen += 1
        stride = C.c_int32()
        roles = (C.c_int32 * 64)(); strides = (C.c_int32 * 64)(); cols = (C.c_int32 * 64)()
        n = L.mistark_contact_recipe(p["name"].encode(), C.byref(stride), roles, strides, cols)
        assert n == len(p["bindings"]), p["name"]
        assert stride.value == p["conn_stride"], p["name"]
        assert [(strides[i], cols[i]) for i in range(n)] == [(b["stride"], b["conn"]) for b in p["bindings"]], p["name"]
        # and the same roles as the oracle's recipe (array identity)
        assert [ROLE_NAMES[roles[i]] for i in range(n)] == [r for r, _, _ in oc.RECIPES[p["name"]][1]], p["name"]
    assert seen == 35

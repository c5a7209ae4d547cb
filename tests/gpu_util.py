"""Test helper: build a stark_amd.Engine from a golden fixture / oracle Problem (generic registration through the C ABI)."""
import numpy as np

import stark_amd


def engine_from_problem(prob, man=None, custom_ops=None):
    """custom_ops: the fixture's npz; every potential is then registered through mistark_potential_custom with the reference's own op
    sequence (interpreted on the device) instead of its compiled kernel."""
    eng = stark_amd.Engine(0)
    host = [np.ascontiguousarray(a, dtype=np.float64) for a in prob.arrays]
    # DoF sets in registration order; empty sets are registered with size 0
    for s, (off, size) in enumerate(zip(prob.dof_offsets, prob.dof_sizes)):
        if s in prob.dof_arrays:
            eng.add_dof_set("set%d" % s, host[prob.dof_arrays[s]])
        else:
            eng.add_dof_set("set%d" % s, np.zeros(0))
    ids = {}
    pot_ids = {}
    for pi, pot in enumerate(prob.potentials):
        if pot.conn.shape[0] == 0:
            continue
        bs = []
        for b in pot.bindings:
            key = (b.array, b.stride)
            if key not in ids:
                ids[key] = eng.array(host[b.array].reshape(-1, b.stride), b.stride)
            bs.append((ids[key], b.stride, b.conn))
        if custom_ops is not None:
            z = custom_ops
            has_c = ("p%d_cops" % pi) in z
            pot_ids[pi] = eng.potential_custom(pot.name, pot.conn, bs, z["p%d_ops" % pi], z["p%d_opsc" % pi], sum(s for _, s, _ in bs),
                                               z["p%d_cops" % pi] if has_c else None, z["p%d_copsc" % pi] if has_c else None)
        else:
            pot_ids[pi] = eng.potential(pot.name, pot.conn, bs)
    eng.host_arrays = host
    eng.array_ids = ids
    eng.pot_ids = pot_ids
    return eng

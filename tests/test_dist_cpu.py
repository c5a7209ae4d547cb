"""CPU, world_size 2 over gloo: the arithmetic of the sharded path (stark_amd/csrc/shard.hip, pcg_sharded in kernels.hip) restated with the
oracle. The block rows are partitioned by the engine's own graph partition (mistark_partition_rows: host code, no GPU); each rank
evaluates ONLY the elements touching its rows, and then
  * the energies of the elements whose first block row a rank owns, all-gathered and summed in rank order, give the unsharded energy;
  * a rank's gradient rows and matrix rows are complete without any exchange (interface elements are evaluated by both sides);
  * the row-sharded block-Jacobi PCG — two exchanges per iteration: p.Ap, and (r.r, r.z) fused with the z of the interface rows, from
    which every rank advances the ghosts of the search direction itself; every rank reducing in rank order — stops at the unsharded
    solve's iteration (+-1) with the same solution, identical bits on both ranks.
The GPU kernels of the same path are covered by tests/test_gpu_sharded.py (2, 3 and 8 ranks)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def partition(L, nbr, world, outs, hub=None):
    tabs = [np.ascontiguousarray(o.block_rows, dtype=np.int32) for o in outs if len(o.E)]
    ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
    n_elem = (C.c_int64 * len(tabs))(*[t.shape[0] for t in tabs])
    nb = (C.c_int32 * len(tabs))(*[t.shape[1] for t in tabs])
    owner = np.zeros(nbr, dtype=np.int32)
    L.mistark_partition_rows.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    hub_p = None if hub is None else np.ascontiguousarray(hub, dtype=np.uint8).ctypes.data
    assert L.mistark_partition_rows(nbr, world, len(tabs), ptrs, n_elem, nb, hub_p, owner.ctypes.data) == 0
    return owner


def _rank_main(rank, world, port, name, result_dir):
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist

    from oracle import evaluator as ev
    from stark_amd import capi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = capi.lib()
    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    n = prob.ndofs
    nbr = n // 3
    full_E, full_g, full_outs = ev.evaluate_all(prob)
    A_full = ev.assemble(full_outs, n)

    def allgather(v):
        out = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(out, v)
        return out

    # ---- partition: the same owner map on every rank, every rank owns rows
    owner = partition(L, nbr, world, full_outs)
    maps = allgather(torch.tensor(owner))
    assert all((m.numpy() == owner).all() for m in maps)
    mine = owner == rank
    assert 0 < mine.sum() < nbr and set(np.unique(owner)) == set(range(world))

    # ---- this rank's problem: the elements touching its rows; an element's energy counts where its first block row lives
    local, E_mine = [], 0.0
    for o in full_outs:
        touch = mine[o.block_rows].any(axis=1)
        local.append(ev.ElementOutput(o.name, o.E[touch], o.g[touch], o.H[touch], o.block_rows[touch], o.active))
        E_mine += float(o.E[mine[o.block_rows[:, 0]]].sum())
    Es = allgather(torch.tensor([E_mine], dtype=torch.float64))
    E_sum = 0.0
    for r in range(world):
        E_sum += float(Es[r])                      # rank order: the same bits everywhere
    assert abs(E_sum - full_E) <= 1e-12 * max(1.0, abs(full_E))
    g_loc = np.zeros(n)
    for o in local:
        nb = o.block_rows.shape[1]
        for k in range(nb):
            np.add.at(g_loc.reshape(-1, 3), o.block_rows[:, k], o.g[:, 3 * k:3 * k + 3])
    own3 = np.repeat(mine, 3)
    assert np.abs(g_loc[own3] - full_g[own3]).max() <= 1e-12 * max(np.abs(full_g).max(), 1e-300)     # complete without exchange
    A_loc = ev.assemble(local, n).to_scipy().tocsr()
    S_full = A_full.to_scipy().tocsr()
    rows_own = np.nonzero(own3)[0]
    assert abs(A_loc[rows_own] - S_full[rows_own]).max() <= 4e-7 * abs(S_full).max()                    # my rows of the matrix, too

    # ---- local numbering: my rows (ascending), then the ghosts grouped by owner; send rows = my rows somebody else holds as ghost
    need = np.zeros((nbr, world), dtype=bool)
    for o in full_outs:
        m = np.zeros((len(o.E), world), dtype=bool)
        for k in range(o.block_rows.shape[1]):
            m[np.arange(len(o.E)), owner[o.block_rows[:, k]]] = True
        for k in range(o.block_rows.shape[1]):
            np.logical_or.at(need, o.block_rows[:, k], m)
    own_rows = np.nonzero(mine)[0]
    ghosts = np.concatenate([np.nonzero((owner == o) & need[:, rank])[0] for o in range(world) if o != rank]).astype(np.int64)
    grow = np.concatenate([own_rows, ghosts])
    lrow = -np.ones(nbr, dtype=np.int64)
    lrow[grow] = np.arange(len(grow))
    n_own, n_loc = len(own_rows), len(grow)
    send_of = [np.nonzero((owner == r) & (need & ~np.eye(world, dtype=bool)[owner]).any(axis=1))[0] for r in range(world)]
    stride = max(len(s) for s in send_of)
    pos_in_send = -np.ones(nbr, dtype=np.int64)
    for r in range(world):
        pos_in_send[send_of[r]] = np.arange(len(send_of[r]))
    ghost_src = owner[ghosts] * stride + pos_in_send[ghosts]
    assert (pos_in_send[ghosts] >= 0).all()

    def exchange2(partials, z_loc):
        """The second exchange of an iteration, as the engine packs it (k_fold_pack): this rank's two partial sums and the z of the rows
        other ranks hold as ghosts, in ONE all-gather. Returns the sums (reduced in rank order) and the ghosts' z."""
        buf = torch.zeros(2 + 3 * stride, dtype=torch.float64)
        buf[0], buf[1] = partials
        mine_send = z_loc.reshape(-1, 3)[lrow[send_of[rank]]].reshape(-1)
        buf[2:2 + len(mine_send)] = torch.tensor(mine_send)
        allb = torch.stack(allgather(buf)).numpy()
        sums = np.zeros(2)
        for r in range(world):
            sums += allb[r, :2]
        zg = allb[:, 2:].reshape(world * stride, 3)[ghost_src]
        return sums, zg

    # my rows of A in local columns
    Aown = A_loc[rows_own].tocoo()
    cols_l = 3 * lrow[Aown.col // 3] + Aown.col % 3
    assert (cols_l >= 0).all()                                   # every column of my rows is mine or a ghost
    A_l = sp.csr_matrix((Aown.data, (Aown.row, cols_l)), shape=(3 * n_own, 3 * n_loc))
    dinv = ev.block_diag_inverse(A_full)[own_rows]

    # ---- the row-sharded PCG (solve_pcg.h:83-232 with the three dot products exchanged)
    abs_tol, rel_tol = man["pcg"]["abs_tol"], 1e-4
    b = -full_g[own3]

    def gsum(vals):  # all-gather of this rank's partial sums, reduced in rank order
        parts = allgather(torch.tensor(vals, dtype=torch.float64))
        out = np.zeros(len(vals))
        for r in range(world):
            out += parts[r].numpy()
        return out

    x = np.zeros(3 * n_own)
    r = b.copy()
    zv = ev.apply_preconditioner(dinv, r)
    p = np.zeros(3 * n_loc)
    (bb, rz), zg = exchange2([float(r @ r), float(r @ zv)], zv)
    p[:3 * n_own] = zv
    p[3 * n_own:] = zg.reshape(-1)                            # p_0 = z_0, on the ghosts too
    its, converged = 0, bb < abs_tol * abs_tol
    while not converged and its < 10000:
        its += 1
        q = A_l @ p                                           # (the ghosts of p are current: no exchange of p itself)
        (pAp,) = gsum([float(p[:3 * n_own] @ q)])             # exchange 1: one number per rank
        if pAp <= 0.0:
            break
        alpha = rz / pAp
        x += alpha * p[:3 * n_own]
        r -= alpha * q
        zv = ev.apply_preconditioner(dinv, r)
        (rr, rz_new), zg = exchange2([float(r @ r), float(r @ zv)], zv)   # exchange 2: (r.r, r.z) and the boundary z, fused
        err = np.sqrt(rr / bb)
        if err < abs_tol or err < rel_tol:
            converged = True
            break
        beta = rz_new / rz
        p[:3 * n_own] = zv + beta * p[:3 * n_own]
        p[3 * n_own:] = zg.reshape(-1) + beta * p[3 * n_own:]  # the owner computes exactly this for the same row: no exchange needed
        rz = rz_new
    # gather the owned parts into the whole solution on every rank
    pad = max(int((owner == q_).sum()) for q_ in range(world))
    buf = torch.zeros(3 * pad, dtype=torch.float64)
    buf[:3 * n_own] = torch.tensor(x)
    parts = allgather(buf)
    x_full = np.zeros(n)
    for q_ in range(world):
        rows_q = np.nonzero(owner == q_)[0]
        x_full.reshape(-1, 3)[rows_q] = parts[q_].numpy()[:3 * len(rows_q)].reshape(-1, 3)
    xo, info_o = ev.solve_pcg(A_full, -full_g, abs_tol)
    assert converged == bool(info_o.converged) and abs(its - info_o.n_iterations) <= 1
    assert abs(its - man["pcg"]["iterations"]) <= 1               # ... and at the reference's
    assert np.abs(x_full - xo).max() <= 1e-4 * max(np.abs(xo).max(), 1e-300)
    same = allgather(torch.tensor(x_full))
    assert all((s.numpy() == x_full).all() for s in same)         # identical bits on every rank

    # ---- the same solve as the engine runs it between processes that exchange through windows (kernels.hip: pcg_sharded_fused): the
    # recurrences of Chronopoulos & Gear (u = M^-1 r, w = A u, s = A p by recurrence), so that ONE message per iteration carries the three
    # sums (gamma = r.u, rr = r.r, delta = w.u; p.Ap = delta - beta gamma / alpha_prev) and the halo of u travels on its own, hidden behind
    # the interior rows of the SpMV. Here: two all-gathers per iteration of which only the scalar one is a dependency of the next step.
    def halo_of(u_loc):
        buf = torch.zeros(3 * stride, dtype=torch.float64)
        mine_send = u_loc.reshape(-1, 3)[lrow[send_of[rank]]].reshape(-1)
        buf[:len(mine_send)] = torch.tensor(mine_send)
        allb = torch.stack(allgather(buf)).numpy()
        return allb.reshape(world * stride, 3)[ghost_src].reshape(-1)

    x2 = np.zeros(3 * n_own)
    r = b.copy()
    u = ev.apply_preconditioner(dinv, r)
    p_ = np.zeros(3 * n_own)
    s_ = np.zeros(3 * n_own)
    u_loc = np.concatenate([u, halo_of(u)])                       # message M1_0
    w = A_l @ u_loc
    gamma, rr, delta = gsum([float(r @ u), float(r @ r), float(w @ u)])   # message M2_0
    bb2 = rr
    its2, conv2 = 0, bb2 < abs_tol * abs_tol
    gamma_prev = alpha_prev = None
    while not conv2 and its2 < 10000:
        beta = 0.0 if gamma_prev is None else gamma / gamma_prev
        pAp = delta if gamma_prev is None else delta - beta * gamma / alpha_prev
        its2 += 1
        if pAp <= 0.0:
            break
        alpha = gamma / pAp
        p_ = u + beta * p_
        s_ = w + beta * s_
        x2 += alpha * p_
        r -= alpha * s_
        u = ev.apply_preconditioner(dinv, r)
        u_loc = np.concatenate([u, halo_of(u)])                   # M1_k
        w = A_l @ u_loc
        gamma_prev, alpha_prev = gamma, alpha
        gamma, rr, delta = gsum([float(r @ u), float(r @ r), float(w @ u)])   # M2_k
        err = np.sqrt(rr / bb2)
        if err < abs_tol or err < rel_tol:
            conv2 = True
    buf = torch.zeros(3 * pad, dtype=torch.float64)
    buf[:3 * n_own] = torch.tensor(x2)
    parts = allgather(buf)
    x2_full = np.zeros(n)
    for q_ in range(world):
        rows_q = np.nonzero(owner == q_)[0]
        x2_full.reshape(-1, 3)[rows_q] = parts[q_].numpy()[:3 * len(rows_q)].reshape(-1, 3)
    assert conv2 == bool(info_o.converged) and abs(its2 - info_o.n_iterations) <= 1, (its2, info_o.n_iterations)
    assert np.abs(x2_full - xo).max() <= 1e-4 * max(np.abs(xo).max(), 1e-300)
    same = allgather(torch.tensor(x2_full))
    assert all((s.numpy() == x2_full).all() for s in same)
    open(os.path.join(result_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tetbeam_full_4x1x1", "tetbeam_eo_4x1x1_big", "cloth_shells_6", "contactmix_t1", "rbchain", "attachzoo"])
def test_sharded_solve_equals_unsharded_gloo(name, tmp_path):
    import torch.multiprocessing as mp

    world = 2
    import socket

    with socket.socket() as sock:   # a port the OS considers free right now (a fixed one can sit in TIME_WAIT from the previous test)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_rank_main, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def test_partition_rows_host():
    """mistark_partition_rows on a chain of segments: contiguous pieces of (almost) equal size, hubs to the last rank."""
    from stark_amd import capi

    L = capi.lib()

    class O:
        pass

    n = 1000
    o = O()
    o.block_rows = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1).astype(np.int32)
    o.E = np.zeros(n - 1)
    for world in (2, 3, 8):
        owner = partition(L, n, world, [o])
        counts = np.bincount(owner, minlength=world)
        assert counts.min() >= n // world - 3 and counts.max() <= n // world + 3
        # pieces are contiguous along the chain (in one direction or the other)
        changes = np.count_nonzero(np.diff(owner))
        assert changes == world - 1
    hub = np.zeros(n, dtype=np.uint8)
    hub[:5] = 1
    owner = partition(L, n, 4, [o], hub)
    assert (owner[:5] == 3).all()

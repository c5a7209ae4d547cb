"""CPU, world_size 2 over gloo: the arithmetic of the multi-GPU path. Each rank takes the contiguous element ranges the engine
would take (mistark_shard_range, the C function the engine's prepare() uses — host code, no GPU), evaluates ONLY those elements
with the oracle, and the sums of energy, gradient and assembled matrix over the ranks (torch.distributed all_reduce, gloo) must
equal the unsharded result. The GPU kernels of the same path are covered by tests/test_gpu_sharded.py."""
import copy
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _rank_main(rank, world, port, name, result_dir):
    import torch
    import torch.distributed as dist

    from oracle import evaluator as ev
    from stark_amd import capi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = capi.lib()
    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    full_E, full_g, full_outs = ev.evaluate_all(prob)
    full_A = ev.assemble(full_outs, prob.ndofs).to_scipy() if hasattr(ev.BSR, "to_scipy") else None
    # this rank's problem: every potential cut to its contiguous range
    local = copy.copy(prob)
    local.potentials = []
    covered = 0
    for p in prob.potentials:
        n = p.conn.shape[0]
        b, e = C.c_int64(), C.c_int64()
        assert L.mistark_shard_range(n, rank, world, C.byref(b), C.byref(e)) == 0
        q = copy.copy(p)
        q.conn = p.conn[b.value:e.value]
        local.potentials.append(q)
        covered += e.value - b.value
        # ranges of all ranks tile [0, n) without gaps or overlap
        edges = []
        for r in range(world):
            bb, ee = C.c_int64(), C.c_int64()
            L.mistark_shard_range(n, r, world, C.byref(bb), C.byref(ee))
            edges.append((bb.value, ee.value))
        assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    E, g, outs = ev.evaluate_all(local)
    t = torch.tensor(np.concatenate([g, [E]]))
    dist.all_reduce(t)                                   # what ncclAllReduce does for (gradient | E) on the GPU path
    g_sum, E_sum = t[:-1].numpy(), float(t[-1])
    assert abs(E_sum - full_E) <= 1e-12 * max(1.0, abs(full_E))
    assert np.abs(g_sum - full_g).max() <= 1e-12 * max(np.abs(full_g).max(), 1e-300)
    # matrix: partial assembly in the GLOBAL pattern, summed over ranks
    import scipy.sparse as sp

    def to_csr(A):
        return sp.bsr_matrix((A.vals.astype(np.float64), A.cols, A.row_ptr), shape=(prob.ndofs, prob.ndofs)).tocsr()

    A_full = to_csr(ev.assemble(full_outs, prob.ndofs))
    A_loc = to_csr(ev.assemble(outs, prob.ndofs)) if outs else sp.csr_matrix((prob.ndofs, prob.ndofs))
    dense = torch.tensor(A_loc.toarray())
    dist.all_reduce(dense)
    assert np.abs(dense.numpy() - A_full.toarray()).max() <= 4e-7 * np.abs(A_full).max()
    # projection round (stark_amd/csrc/kernels.hip: exchange_projection_deltas): every rank projects ITS elements, the deltas travel as
    # (position, value) records in a common zero-filled float buffer (all-reduce = exchange), every rank sorts them by position (stable)
    # and adds them to its copy of the summed matrix: the result must equal the matrix assembled from all projected Hessians
    n = prob.ndofs
    recs = []
    for o in outs:
        Hp, changed = ev.project_to_pd(o.H)
        d = (Hp - o.H).astype(np.float32)
        nb = o.block_rows.shape[1]
        for e in np.nonzero(changed)[0]:
            for a in range(nb):
                for b in range(nb):
                    for i in range(3):
                        for j in range(3):
                            recs.append(((3 * o.block_rows[e, a] + i) * n + 3 * o.block_rows[e, b] + j, d[e, 3 * a + i, 3 * b + j]))
    cnt = torch.zeros(world, dtype=torch.float64)
    cnt[rank] = len(recs)
    dist.all_reduce(cnt)
    counts = [int(c) for c in cnt]
    total, offset = sum(counts), sum(counts[:rank])
    buf = torch.zeros(3 * max(total, 1), dtype=torch.float32)
    for k, (pos, v) in enumerate(recs):
        buf[3 * (offset + k)] = float(pos >> 16)      # positions as two exact floats, like the engine
        buf[3 * (offset + k) + 1] = float(pos & 0xffff)
        buf[3 * (offset + k) + 2] = float(v)
    dist.all_reduce(buf)
    b3 = buf.numpy().reshape(-1, 3)[:total]
    pos = (b3[:, 0].astype(np.int64) << 16) | b3[:, 1].astype(np.int64)
    patched = dense.numpy().astype(np.float32).reshape(-1).copy()
    for k in np.argsort(pos, kind="stable"):
        patched[pos[k]] += b3[k, 2]
    full_proj = []
    for o in full_outs:
        Hp, _ = ev.project_to_pd(o.H)
        full_proj.append(ev.ElementOutput(o.name, o.E, o.g, Hp, o.block_rows, o.active))
    A_proj = to_csr(ev.assemble(full_proj, prob.ndofs)).toarray()
    assert np.abs(patched.reshape(n, n) - A_proj).max() <= 4e-6 * np.abs(A_proj).max()
    count = torch.tensor([float(covered)])
    dist.all_reduce(count)
    assert int(count.item()) == sum(p.conn.shape[0] for p in prob.potentials)
    open(os.path.join(result_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tetbeam_full_4x1x1", "contactmix_t1"])
def test_sharded_sums_equal_unsharded_gloo(name, tmp_path):
    import torch.multiprocessing as mp

    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))

#!/usr/bin/env python3
"""The IPC-window transport by itself, one process per rank (include/mistark.h "IPC windows"):

    MISTARK_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 tools/ipc_selftest.py

Every rank creates its window, the handles travel over gloo, every rank maps the others' windows, and all-gathers of several sizes run with every
received value checked. Prints rank 0's average wall time per exchange (push kernel + polling kernel + stream synchronisation on an idle stream).
MISTARK_BENCH_DEVICE: all ranks on that device (a one-GPU box); default: rank r on device r."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from stark_amd import capi

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    device = int(os.environ.get("MISTARK_BENCH_DEVICE", local))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.cuda.set_device(device)
    dist.init_process_group(backend="gloo")

    def allgather_bytes(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    comm = capi.IpcComm(device, rank, world, 64 << 20, allgather_bytes)
    res = {}
    for n in (1, 16, 1024, 10 * 1024, 100 * 1024, 600 * 1024):
        dist.barrier()
        res[str(n)] = [round(v, 2) for v in comm.selftest(n, 50 if n <= 10240 else 5)]
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "one_device": "MISTARK_BENCH_DEVICE" in os.environ, "allgather_us_by_doubles [synchronised, in a train]": res}))
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

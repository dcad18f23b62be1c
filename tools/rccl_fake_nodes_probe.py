"""Can REAL RCCL run several ranks on ONE GPU?  RCCL refuses two ranks with the same (host hash, PCI bus id); NCCL_HOSTID
overrides the host hash, so ranks that claim different hosts pass the duplicate-device check and talk over the socket
transport on the loopback interface.  Slow (host staging + TCP), but it is RCCL's own bootstrap, group launches and
collective kernels at world > 1 -- what a 1-GPU box otherwise never runs.

  python tools/rccl_fake_nodes_probe.py [world]
"""
import os
import sys
import time


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_HOSTID="tfk-fake-node-%d" % rank,
                      NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    t0 = time.time()
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.full((1 << 20,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2
    assert float(x[0]) == want and float(x[-1]) == want, (float(x[0]), want)
    t1 = time.time()
    n = (1 << 22) // world * world
    src = (torch.arange(n, device="cuda") % 1000).float() * (rank + 1)  # (sums stay exact in fp32 in any order)
    out = torch.empty(n // world, device="cuda")
    dist.reduce_scatter_tensor(out, src)
    torch.cuda.synchronize()
    ref = (torch.arange(n) % 1000).float()[rank * (n // world):(rank + 1) * (n // world)] * want
    assert torch.equal(out.cpu(), ref)
    full = torch.empty(n, device="cuda")
    dist.all_gather_into_tensor(full, out)
    torch.cuda.synchronize()
    assert torch.equal(full.cpu(), (torch.arange(n) % 1000).float() * want)
    # grouped point-to-point: every rank sends a slice to every other rank
    ops, recv = [], {}
    for q in range(world):
        if q == rank:
            continue
        recv[q] = torch.empty(1024, device="cuda")
        ops.append(dist.P2POp(dist.isend, torch.full((1024,), float(rank * 100 + q), device="cuda"), q))
        ops.append(dist.P2POp(dist.irecv, recv[q], q))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    for q, t in recv.items():
        assert float(t[0]) == q * 100 + rank
    t2 = time.time()
    dist.barrier()
    if rank == 0:
        print("FAKE_NODES_OK world=%d init+allreduce %.1fs, rs+ag+p2p %.2fs" % (world, t1 - t0, t2 - t1), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)

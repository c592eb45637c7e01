"""How long does torch.distributed.barrier() hold the host on a world of one over RCCL, against an all_reduce of one element +
synchronize?  (bench.py's timed region starts behind one: the device idles meanwhile and its clock falls.)
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 scratch/t_barrier.py"""
import os, time, torch, torch.distributed as dist
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev) if "DEVID" in os.environ else dist.init_process_group("nccl")
x = torch.ones(1, device=dev)
def t(fn, n=6):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(round(1e3 * (time.perf_counter() - t0), 3))
    return out
print("barrier()            ", t(lambda: dist.barrier()))
print("barrier(device_ids)  ", t(lambda: dist.barrier(device_ids=[dev.index])))
print("all_reduce + sync    ", t(lambda: dist.all_reduce(x)))
y = torch.zeros(1 << 20, device=dev)
def busy():
    for _ in range(50): y.add_(1.0)
busy(); torch.cuda.synchronize()
print("after busy: barrier()", t(lambda: (busy(), dist.barrier())))
print("after busy: allreduce", t(lambda: (busy(), dist.all_reduce(x))))
dist.destroy_process_group()

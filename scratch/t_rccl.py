# experiment: does the RCCL path of bench.py (init, barrier, all_gather_into_tensor, all_reduce MAX) run on this box?  world = 1
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev)
from robot_camera_calibration_amd import dist as rdist
g = rdist.PoseGather(8, dev, 1, dist)
send = torch.arange(8 * 17, dtype=torch.float64, device=dev).reshape(8, 17)
recv = torch.zeros_like(send)
dist.all_gather_into_tensor(recv, send)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print("rccl ok", bool((recv == send).all()), float(t.item()))
dist.destroy_process_group()

"""One-rank RCCL probe: device and host cost of dist.all_reduce on the flat gradient buffer (what the collective's launch
path - stream hand-over to the RCCL stream and back - costs before any link traffic)."""
import os, time, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (3_600_000, 1_000_000, 1024):
    g = torch.zeros(n, device="cuda")
    for _ in range(5):
        dist.all_reduce(g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    t0 = time.perf_counter(); e0.record()
    for _ in range(K):
        dist.all_reduce(g)
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("n=%d  device %.1f us/call  host enqueue %.1f us/call" % (n, e0.elapsed_time(e1) * 1e3 / K, (t1 - t0) * 1e6 / K))
    # between two small kernels on the current stream
    x = torch.zeros(1024, device="cuda")
    e0.record()
    for _ in range(K):
        x.add_(1.0); dist.all_reduce(g); x.add_(1.0)
    e1.record(); torch.cuda.synchronize()
    a = e0.elapsed_time(e1) * 1e3 / K
    e0.record()
    for _ in range(K):
        x.add_(1.0); x.add_(1.0)
    e1.record(); torch.cuda.synchronize()
    print("      add, all_reduce, add: %.1f us   add, add: %.1f us" % (a, e0.elapsed_time(e1) * 1e3 / K))
dist.destroy_process_group()

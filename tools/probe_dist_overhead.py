"""Does a torch.distributed collective on the nccl (= RCCL) backend block the HOST until earlier GPU work has finished?
One rank.  Enqueues ~100 ms of GPU work, then times the host side of all_reduce / all_gather_into_tensor calls.  Diagnostic."""
import os, socket, time
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
x = torch.randn(8192, 8192, device=dev)
two = torch.zeros(2, device=dev); loc = torch.zeros(1024, device=dev); out = torch.empty(1024, device=dev)
for _ in range(3):
    dist.all_reduce(two, op=dist.ReduceOp.MAX); dist.all_gather_into_tensor(out, loc)
torch.cuda.synchronize()


def busy():
    for _ in range(12):
        (x @ x)


for name, fn in (("all_reduce(MAX)", lambda: dist.all_reduce(two, op=dist.ReduceOp.MAX)),
                 ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(out, loc)),
                 ("all_reduce(MAX, async_op=True)", lambda: dist.all_reduce(two, op=dist.ReduceOp.MAX, async_op=True)),
                 ("tensor.max() only", lambda: two.max())):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); busy(); t1 = time.perf_counter()
    fn()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{name}: enqueue of the GPU work {1e3 * (t1 - t0):.2f} ms, host time of the call behind it {1e3 * (t2 - t1):.3f} ms, "
          f"GPU drained after another {1e3 * (t3 - t2):.1f} ms", flush=True)
dist.destroy_process_group()

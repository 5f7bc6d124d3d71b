import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=torch.device("cuda", 0))
x = torch.arange(8, device="cuda", dtype=torch.float32).reshape(2, 4)
out = torch.empty(2, 4, device="cuda")
dist.all_gather_into_tensor(out, x); dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl world-1 ok:", torch.equal(out, x), float(t))
dist.destroy_process_group()

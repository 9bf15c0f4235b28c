import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(8, device="cuda", dtype=torch.float32)
dist.broadcast(t, src=0)
out = [torch.zeros_like(t)]
dist.all_gather(out, t)
g = [torch.zeros_like(t)]
dist.gather(t, g, dst=0)
m = torch.tensor([3.5], device="cuda", dtype=torch.float64)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("nccl world 1 ok", dist.get_backend(), out[0].tolist(), g[0].tolist(), m.item())
dist.destroy_process_group()

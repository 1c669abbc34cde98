import os, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
x = torch.ones(1024, device="cuda") * (rank + 1)
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", rank, "allreduce ok", float(x[0]), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): dist.all_reduce(x)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    dist.all_reduce(x)
g.replay(); torch.cuda.synchronize()
print("rank", rank, "graph allreduce ok", float(x[0]), flush=True)
dist.destroy_process_group()

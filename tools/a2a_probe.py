"""Probe: does all_to_all_single survive > 2 GiB messages on this RCCL/torch build?"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
for n in (50_000_000, 100_000_000, 300_000_000, 400_000_000):
    x = torch.arange(n, dtype=torch.int64, device="cuda") * 7 + 3
    y = torch.empty_like(x)
    dist.all_to_all_single(y, x, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad = int((x != y).sum().item())
    print(f"n={n} bytes={n*8/2**30:.2f} GiB mismatches={bad}", flush=True)
dist.destroy_process_group()

"""Host and device cost of one torch.distributed.all_to_all_single (RCCL) call at halo-record sizes,
measured with world_size 1 (self-exchange) -- a lower bound for the per-iteration exchange cost."""
import os, time
import torch, torch.distributed as dist
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29578")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for n in (1000, 10000, 60000):
    send = torch.zeros(n * 13, dtype=torch.float64, device="cuda")
    recv = torch.zeros_like(send)
    for async_op in (False, True):
        for _ in range(20):
            dist.all_to_all_single(recv, send, [n * 13], [n * 13])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for _ in range(K):
            if async_op:
                w = dist.all_to_all_single(recv, send, [n * 13], [n * 13], async_op=True)
                w.wait()
            else:
                dist.all_to_all_single(recv, send, [n * 13], [n * 13])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"slots={n} bytes={n*104} async={async_op}: host {1e6*(t1-t0)/K:.1f} us/call, total {1e6*(t2-t0)/K:.1f} us/call", flush=True)
dist.destroy_process_group()

"""hipMalloc cost vs size on the box (decides how the path pass sizes its DP-table arena)."""
import time

import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()


def t_alloc(gb):
    t0 = time.perf_counter()
    x = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    return x, 1e3 * (time.perf_counter() - t0)


for gb in (1, 2, 4, 6, 8, 10, 12, 16, 24):
    x, ms = t_alloc(gb)
    del x
    torch.cuda.empty_cache()
    print(f"single {gb} GB: {ms:.1f} ms", flush=True)
for chunk in (4, 8):
    held, tot = [], 0.0
    for i in range(int(160 / chunk)):
        x, ms = t_alloc(chunk)
        held.append(x)
        tot += ms
    print(f"{len(held)} x {chunk} GB chunks held together: {tot:.1f} ms total, last {ms:.1f} ms", flush=True)
    t0 = time.perf_counter()
    for x in held:
        x.fill_(1)
    torch.cuda.synchronize()
    print(f"  fill of all: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    del held, x
    torch.cuda.empty_cache()

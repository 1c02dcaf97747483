"""Device-to-device copy rate of the box (confirm the HBM figure the roofline is priced against).
A copy reads and writes every byte, so HBM traffic = 2 x bytes copied.  Uses hipMemcpyAsync D2D through torch
(plumbing only) and HIP events; prints one JSON line."""
import json

import torch

assert torch.cuda.is_available()
out = {}
for gib in (1, 4):
    n = gib << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n, dtype=torch.uint8, device="cuda")
    src.fill_(1)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[f"copy_{gib}GiB_GBs_read_plus_write"] = 2 * n / (ms * 1e-3) / 1e9
    del src, dst
print(json.dumps(out))

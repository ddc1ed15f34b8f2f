import torch
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n = 1 << 30
x = torch.empty(n, dtype=torch.uint8, device="cuda"); y = torch.empty(n, dtype=torch.uint8, device="cuda")
ms = t(lambda: y.copy_(x)); print(f"copy 1 GiB: {2*n/ms/1e6:.0f} GB/s (r+w)")
ms = t(lambda: y.zero_()); print(f"fill 1 GiB: {n/ms/1e6:.0f} GB/s (w)")
xf = x.view(torch.float32)
ms = t(lambda: xf.sum()); print(f"read-reduce 1 GiB: {n/ms/1e6:.0f} GB/s (r)")

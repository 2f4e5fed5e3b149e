"""Dev: practical HBM bandwidth of this GPU for plain streaming kernels (context for the HBM-bound conv layers)."""
import time, torch
dev = torch.device("cuda:0")
for mb in (100, 400, 1600):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.float16, device=dev).normal_()
    y = torch.empty_like(x)
    for name, fn, nbytes in (("copy (1R+1W)", lambda: y.copy_(x), 2 * n * 2), ("relu_ in place (1R+1W)", lambda: x.relu_(), 2 * n * 2),
                             ("sum (1R)", lambda: x.sum(), n * 2), ("fill (1W)", lambda: y.fill_(1.0), n * 2)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{mb:5d} MB  {name:24s} {nbytes / dt / 1e12:6.2f} TB/s")

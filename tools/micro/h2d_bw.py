import torch, time
x = torch.empty(1_000_000_000, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for n in (1_000_000_000, 64 << 20, 8 << 20):
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = max(1, 1_000_000_000 // n)
        for i in range(reps):
            d[i * n:(i + 1) * n].copy_(x[i * n:(i + 1) * n], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("chunk", n, "GB/s %.1f" % (reps * n / dt / 1e9), "ms per GB %.1f" % (dt * 1e3 * 1e9 / (reps * n)))

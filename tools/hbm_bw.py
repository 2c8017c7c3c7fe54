"""Practical HBM bandwidth of this box: fill (write-only), copy (read + write), reduce (read-only), via torch."""
import time, torch
n = 1 << 30
a = torch.empty(n // 4, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print(f"fill  (1 GiB written)        {n / t(lambda: a.fill_(1.0)) / 1e12:.2f} TB/s")
print(f"copy  (1 GiB read + written) {2 * n / t(lambda: b.copy_(a)) / 1e12:.2f} TB/s")
print(f"sum   (1 GiB read)           {n / t(lambda: a.sum()) / 1e12:.2f} TB/s")

"""What a plain streaming read reaches on this part (ceiling for the frame-embedding kernels): torch reductions / copies."""
import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for gb in (1, 4):
    x = torch.empty(gb * (1 << 30) // 4, device="cuda", dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    dt = t(lambda: x.sum()); print(f"{gb} GiB  sum (read only)     : {x.numel()*4/dt/1e12:.2f} TB/s")
    dt = t(lambda: torch.amax(x)); print(f"{gb} GiB  amax (read only)    : {x.numel()*4/dt/1e12:.2f} TB/s")
    dt = t(lambda: y.copy_(x)); print(f"{gb} GiB  copy (read + write) : {2*x.numel()*4/dt/1e12:.2f} TB/s")
    xb = x.view(torch.int32)
    dt = t(lambda: torch.bitwise_xor(xb, 1, out=y.view(torch.int32))); print(f"{gb} GiB  xor  (read + write) : {2*x.numel()*4/dt/1e12:.2f} TB/s")

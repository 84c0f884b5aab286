"""Measured HBM bandwidth of the box (report nominal and measured peak)."""
import time
import torch
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
for name, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 2 * n),
                              ("fill (write)", lambda: b.fill_(3), n),
                              ("sum (read)", lambda: a.view(torch.int64).sum(), n)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%-20s %6.2f ms  %7.1f GB/s" % (name, dt * 1e3, bytes_moved / dt / 1e9))
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory >> 30, "GiB")

"""Achievable HBM bandwidth on this box (torch copy / read-reduce), for the roofline denominators."""
import time
import torch
x = torch.empty(1 << 29, dtype=torch.float32, device="cuda")   # 2 GiB
y = torch.empty_like(x)
x.normal_()
for name, fn, bytes_ in (("copy", lambda: y.copy_(x), 2 * x.numel() * 4), ("read(sum)", lambda: x.sum(), x.numel() * 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print("%s: %.1f GB/s" % (name, bytes_ / dt / 1e9))
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")

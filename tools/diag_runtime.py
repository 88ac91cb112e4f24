import ctypes, os, sys
def maps(tag):
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(s in l for s in ("amdhip", "hsa-runtime", "libshodh", "amd_comgr", "rocprofiler"))})
    print(tag, libs, flush=True)
mode = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if mode == "torch_first":
    import torch
    maps("after import torch")
    print("is_available", torch.cuda.is_available(), flush=True)
    maps("after is_available")
    from shodh_memory_amd import _lib
    L = _lib.lib()
    maps("after lib load")
    print("shodh_device_count", L.shodh_device_count(), _lib.last_error(), flush=True)
    x = torch.zeros(4, device="cuda"); print("torch tensor ok", x.sum().item())
    print("shodh_device_count again", L.shodh_device_count(), _lib.last_error(), flush=True)
elif mode == "lib_first":
    from shodh_memory_amd import _lib
    L = _lib.lib()
    maps("after lib load")
    print("shodh_device_count", L.shodh_device_count(), _lib.last_error(), flush=True)
    import torch
    maps("after import torch")
    try:
        print("is_available", torch.cuda.is_available(), flush=True)
        x = torch.zeros(4, device="cuda"); print("torch tensor ok", x.sum().item())
    except Exception as e:
        print("torch failed:", e)
    print("shodh_device_count again", L.shodh_device_count(), _lib.last_error(), flush=True)
elif mode == "torch_init_first":
    import torch
    x = torch.zeros(4, device="cuda"); print("torch tensor ok", x.sum().item())
    from shodh_memory_amd import _lib
    L = _lib.lib()
    maps("after lib load")
    print("shodh_device_count", L.shodh_device_count(), _lib.last_error(), flush=True)
print("env", {k: v for k, v in os.environ.items() if "HIP" in k or "ROC" in k or "HSA" in k or "CUDA" in k or "LD_" in k})

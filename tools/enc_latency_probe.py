"""one text per call through the encoder (what encode() / encode_query() cost a remember / recall): python tools/enc_latency_probe.py [bf16|int8] [tokens]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import shodh_memory_amd as S
from shodh_memory_amd import _lib as L
dname = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
enc = S.MiniLMEmbedder(synthetic_seed=1234, dtype={"bf16": L.DTYPE_BF16, "int8": L.DTYPE_INT8, "fp32": L.DTYPE_FP32}[dname])
g = torch.Generator(device=dev).manual_seed(1)
ids = torch.zeros((1, 256), dtype=torch.int32, device=dev); mask = torch.zeros((1, 256), dtype=torch.uint8, device=dev)
ids[0, :ntok] = torch.randint(1000, 29000, (ntok,), generator=g, device=dev, dtype=torch.int32); mask[0, :ntok] = 1
out = torch.empty((1, 384), dtype=torch.float32, device=dev)
for _ in range(20): enc.encode_ids_device(ids, mask, out=out)
torch.cuda.synchronize()
lat = []
for _ in range(200):
    torch.cuda.synchronize(); t = time.perf_counter()
    enc.encode_ids_device(ids, mask, out=out)
    torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
lat.sort()
hid = ids.cpu().numpy(); hmask = mask.cpu().numpy()
hl = []
for _ in range(100):
    t = time.perf_counter(); enc.encode_ids(hid, hmask); hl.append(time.perf_counter() - t)
hl.sort()
print(json.dumps({"encoder": dname, "tokens": ntok, "device_ptr_p50_ms": round(lat[100] * 1e3, 4), "device_ptr_p95_ms": round(lat[190] * 1e3, 4), "host_ptr_p50_ms": round(hl[50] * 1e3, 4),
                  "device_us_last": enc.stage_timings_us()}))

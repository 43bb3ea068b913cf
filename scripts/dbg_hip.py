import ctypes as C, os, sys
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch
    print("torch cuda avail", torch.cuda.is_available(), torch.cuda.device_count())
from playaid_core_amd import _lib
lib = _lib.load()
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.split("\n") if "amdhip64" in l or "hsa-runtime" in l}))
for name in ["libamdhip64.so.7"]:
    h = C.CDLL(name)
    n = C.c_int(-1)
    rc = h.hipGetDeviceCount(C.byref(n))
    print(name, "hipGetDeviceCount rc", rc, "n", n.value)
print({k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "ROCR" in k})

"""Experiment: D2H of a 4 GB matrix into pageable numpy memory vs hipHostRegister'ed numpy memory."""
import ctypes, time
import numpy as np
import torch
hip = ctypes.CDLL("libamdhip64.so")
n = 50000 * 20000
src = torch.rand(n, device="cuda")
torch.cuda.synchronize()
dst = np.empty(n, np.float32)
dst[:] = 0                                   # touch pages
for label in ("pageable", "registered", "pageable", "registered"):
    t0 = time.perf_counter()
    if label == "registered":
        rc = hip.hipHostRegister(ctypes.c_void_p(dst.ctypes.data), ctypes.c_size_t(dst.nbytes), 0)
        t1 = time.perf_counter()
    else:
        rc, t1 = 0, t0
    hip.hipMemcpy(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(dst.nbytes), 2)
    t2 = time.perf_counter()
    if label == "registered":
        hip.hipHostUnregister(ctypes.c_void_p(dst.ctypes.data))
    t3 = time.perf_counter()
    print("%-10s rc=%d register %.3f s copy %.3f s (%.1f GB/s) unregister %.3f s total %.3f s" % (
        label, rc, t1 - t0, t2 - t1, dst.nbytes / (t2 - t1) / 1e9, t3 - t2, t3 - t0))
# fresh (untouched) destination, as np.empty gives it to brie_read
for label in ("pageable-fresh", "registered-fresh"):
    d2 = np.empty(n, np.float32)
    t0 = time.perf_counter()
    if label.startswith("registered"):
        hip.hipHostRegister(ctypes.c_void_p(d2.ctypes.data), ctypes.c_size_t(d2.nbytes), 0)
    hip.hipMemcpy(ctypes.c_void_p(d2.ctypes.data), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(d2.nbytes), 2)
    if label.startswith("registered"):
        hip.hipHostUnregister(ctypes.c_void_p(d2.ctypes.data))
    print("%-16s total %.3f s" % (label, time.perf_counter() - t0))

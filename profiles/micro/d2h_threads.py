"""Pageable device-to-host copies: one thread against four (one 4-GiB array each).  Is the 20 GB/s of the streamed
result matrices a per-thread limit (staging through the runtime's pinned buffers) or the link?"""
import json, threading, time
import numpy as np
import torch

dev = torch.device("cuda", 0)
src = [torch.randn(1 << 30, device=dev) for _ in range(4)]
dst = [np.empty(1 << 30, np.float32) for _ in range(4)]
for d in dst:
    d.fill(0)                                   # first touch
torch.cuda.synchronize()


def copy(i, stream):
    with torch.cuda.stream(stream):
        t = torch.from_numpy(dst[i])
        t.copy_(src[i], non_blocking=False)


streams = [torch.cuda.Stream() for _ in range(4)]
for rnd in range(3):
    t0 = time.perf_counter()
    for i in range(4):
        copy(i, streams[0])
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=copy, args=(i, streams[i])) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    par = time.perf_counter() - t0
    print(json.dumps({"one_thread_GBs": round(16 * 1.0737 / seq, 1), "four_threads_GBs": round(16 * 1.0737 / par, 1)}), flush=True)

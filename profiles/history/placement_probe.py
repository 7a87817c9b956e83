"""Is the streaming rate a property of WHERE an allocation landed?  40 separate 4 GiB device allocations; an in-place
elementwise pass (read + write) over each, three sweeps.  Reproducible per-buffer rates that differ between buffers mean the
physical backing of an allocation (fragment sizes / channel spread the driver happened to give it) sets the rate, which
would explain why one and the same kernel on the same virtual addresses runs at 8.0 or 9.8 ms after a re-allocation."""
import json, sys, time
import torch


def main():
    dev = torch.device("cuda", 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bufs = [torch.zeros(1 << 30, dtype=torch.float32, device=dev) for _ in range(n)]       # 4 GiB each
    torch.cuda.synchronize()
    rates = []
    for sweep in range(3):
        row = []
        for b in bufs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                b.add_(1.0)
            e1.record()
            torch.cuda.synchronize()
            row.append(round(3 * 2 * b.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9))
        rates.append(row)
        print(json.dumps({"sweep": sweep, "GBs": row}), flush=True)
    # three streams at once (two reads, one write) over neighbouring buffers: does a REGION stream slower?
    for sweep in range(2):
        row = []
        for i in range(n - 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                torch.add(bufs[i], bufs[i + 1], out=bufs[i + 2])
            e1.record()
            torch.cuda.synchronize()
            row.append(round(3 * 3 * bufs[i].numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9))
        print(json.dumps({"three_stream_sweep": sweep, "GBs": row}), flush=True)
    import numpy as np
    r = np.array(rates[1:], float)
    print(json.dumps({"per_buffer_mean_min": float(r.mean(0).min()), "per_buffer_mean_max": float(r.mean(0).max()),
                      "spread_between_buffers_pct": float((r.mean(0).max() / r.mean(0).min() - 1) * 100),
                      "mean_spread_within_buffer_pct": float(((r.max(0) / r.min(0)) - 1).mean() * 100),
                      "addresses_GiB": [round(b.data_ptr() / 2**30, 1) for b in bufs]}))


if __name__ == "__main__":
    main()

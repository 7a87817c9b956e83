"""In ONE process: create / time / destroy the headline-shape handle repeatedly and print where its arrays live.
coupled_bench.py showed step times alternating 8.5 / 10.2 ms between consecutive handles of the same kernel."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from brie_amd import _capi
    Nc, Ng, Kc = 50000, 20000, 3
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]
    Xc = torch.randn(Nc, Kc, generator=g, device=dev)
    hold = []
    for it in range(10):
        if it == 6:
            hold.append(torch.empty(3 << 30, dtype=torch.uint8, device=dev))       # shift the following allocations
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1)
        for l in range(2):
            sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        sh.upload(_capi.XC, Xc)
        sh.init_state()
        sh.step(3, 0.005, 1, trace=False)
        sh.synchronize()
        sh.profile_enable(True)
        sh.step(12, 0.005, 1, trace=False)
        ms, n = sh.profile_read()
        addrs = [sh.debug_address(w) for w in (0, 1, 8, 9, 20, 21, 22, 23)]
        print(json.dumps({"it": it, "kernel_ms": round(ms / n, 3), "addr": ["%x" % a for a in addrs],
                          "free_GB": round(_capi.device_memory(0)[0] / 2 ** 30, 1)}), flush=True)
        sh.close()


if __name__ == "__main__":
    main()

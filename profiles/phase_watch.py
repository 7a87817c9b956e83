"""Does the step time of the narrow kernel drift with the box, and do other kernels drift with it?

One process, ~70 s: every cycle times 8 steps of (a) the narrow model (Kc = 3), (b) the MFMA tile kernel with the smallest
wide design (Kc = 9), (c) the narrow model on fp32 count storage, and one pass of (d) the no-arithmetic stream_mix kernel
(8 reads / 6 writes), and samples the clocks and power rocm-smi reports.  Same counts for all handles.
"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"], capture_output=True, text=True,
                             timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if "level" in kl:
                continue
            for tag in ("sclk", "mclk", "fclk", "socclk", "power", "junction", "memory) (c"):
                if tag in kl:
                    keep[tag.replace(") (c", "_temp")] = str(v).strip("()")
        return keep
    except Exception as e:          # noqa
        return {"smi_error": str(e)[:80]}


def main():
    import torch
    from brie_amd import _capi
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]

    def make(Kc, storage=None):
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1)
        if storage == "f32":
            sh.set_count_storage(1)
        for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
        sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
        return sh

    def t(sh, n=8):
        t0 = time.perf_counter(); sh.step(n, 0.005, 1, trace=False); sh.synchronize()
        return round((time.perf_counter() - t0) / n * 1e3, 3)
    if os.environ.get("WATCH_ORDER") == "f32_first":       # does the order of creation (= where the arrays land) matter?
        f32 = make(3, "f32"); tile = make(9); narrow = make(3)
    else:
        narrow, tile, f32 = make(3), make(9), make(3, "f32")
    print(json.dumps({"storage": [narrow.count_storage, tile.count_storage, f32.count_storage]}), flush=True)
    t_end = time.time() + float(os.environ.get("WATCH_SECONDS", "70"))
    i = 0
    while time.time() < t_end:
        row = {"t": round(time.time() % 1000, 1), "narrow": t(narrow), "tile_kc9": t(tile), "narrow_f32": t(f32)}
        if i % 4 == 0:
            row["stream_8r6w_GBs"] = round(_capi.calibrate_stream(8, 6, 1 << 28, 3), 0)
            row.update(smi())
        print(json.dumps(row), flush=True)
        i += 1


if __name__ == "__main__":
    main()

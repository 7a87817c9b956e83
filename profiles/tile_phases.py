"""Where does a tile of elbo_adam_step_tile spend its cycles?  Needs a library built with -DBRIE_TILE_PROF=1:

    python -c "from brie_amd.build import compile_library, LIB_DIR; compile_library(out=LIB_DIR+'/variants/libbrie_amd_prof.so', defines=['BRIE_TILE_PROF=1'])"
    BRIE_AMD_LIB=brie_amd/lib/variants/libbrie_amd_prof.so python profiles/tile_phases.py     (GPU box)

Prints, per model, the average cycles per tile of wave 0 and of the waves 1-3 of a half in:
A forward MFMA | wait | B stream 8 rows | wait | C1 Xc^T.R | C2 R.Xg | wait.
"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from brie_amd import _capi
    lib = _capi.load_library()
    lib.brie_debug_tile_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]
    names = ["A_mfma", "wait1", "B_stream", "wait2", "C1_G", "C2_P", "wait3"]
    for (Kc, Kg, nh) in ((3, 64, 2), (3, 64, 1), (3, 32, 1), (3, 32, 2), (64, 0, 2), (32, 0, 1), (3, 16, 1)):
        os.environ["BRIE_TILE_HALVES"] = str(nh)
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1, Kg=Kg)
        for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
        if Kg: sh.upload(_capi.XG, (torch.randn(Ng, Kg, generator=g, device=dev) * 0.3).cpu().numpy())
        sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.brie_debug_tile_prof(buf, 1)
        t0 = time.perf_counter(); sh.step(6, 0.005, 1, trace=False); sh.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        lib.brie_debug_tile_prof(buf, 1)
        tiles = max(1, buf[7])
        w0 = {n: round(buf[i] / tiles) for i, n in enumerate(names)}
        w123 = {n: round(buf[8 + i] / tiles / 3) for i, n in enumerate(names)}
        print(json.dumps({"Kc": Kc, "Kg": Kg, "halves": nh, "ms_per_step": round(ms, 2), "tiles": int(tiles),
                          "cycles_per_tile_wave0": w0, "sum0": sum(w0.values()), "cycles_per_tile_waves123": w123}), flush=True)
        sh.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- ELBO iterations/s of the brie-quant hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one fused ELBO-gradient + Adam pass over every (cell, gene) element
of the workload (= one iteration of tfp.math.minimize in
/root/reference/brie/models/model_TFProb.py:239-241).  Workload = BASELINE.json
configs[2] (the headline): 50k cells x 20k genes, 2 count layers, 3 cell
covariates + gene intercept, MC_size=1, fp32; synthetic counts generated on the
device with the SURVEY 8(d) recipe and resident in HBM before the timed region.
With N>1 the 20k genes are sharded over the ranks (BASELINE configs[3]); there
is no collective inside the timed loop (genes are independent), so `value` is
total element-iterations of all ranks / max-over-ranks time.

One JSON line on stdout (rank 0) with `roofline` (HIP-event time of the
dominant kernel, algorithmic bytes 48+4L per element) and `cpu_baseline`
(the oracle's eager torch-CPU restatement in the reference's execution shape).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: Nc, Ng, Kc, L, theta, depth
    "c1": dict(Nc=200, Ng=500, Kc=0, L=2, theta=3.0, depth=2.0, desc="200x500, 2 isoforms (BASELINE configs[0])"),
    "c2": dict(Nc=10000, Ng=5000, Kc=1, L=3, theta=1.5, depth=2.0, desc="10k x 5k SE events, effLen, Kc=1 (configs[1])"),
    "c3": dict(Nc=50000, Ng=20000, Kc=3, L=2, theta=1.5, depth=2.0, desc="50k x 20k, Kc=3 + gene intercept (configs[2], headline)"),
    "c5": dict(Nc=100000, Ng=30000, Kc=5, L=2, theta=1.5, depth=1.0, desc="DMG 100k x 30k spliced/unspliced, Kc=5 (configs[4])"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GEN_CHUNK = 500                # genes per generation chunk (seeded per chunk => shard-invariant data)


def gen_chunk(torch, dev, cfg, Xc, size, c0, c1, seed):
    """Synthetic counts for genes [c0, c1) on the device (SURVEY 8d recipe)."""
    Nc, Kc, L = cfg["Nc"], cfg["Kc"], cfg["L"]
    g = torch.Generator(device=dev)
    g.manual_seed(seed * 1000003 + c0)
    n = c1 - c0
    W = torch.randn(Kc, n, generator=g, device=dev) * (torch.rand(Kc, n, generator=g, device=dev) < 0.2)
    b = torch.randn(n, generator=g, device=dev) * cfg["theta"]
    sig = torch.rand(n, generator=g, device=dev) * 1.5 + 0.5
    Z = b[None, :] + sig[None, :] * torch.randn(Nc, n, generator=g, device=dev)
    if Kc:
        Z = Z + Xc @ W
    psi = torch.sigmoid(Z.clamp_(-9, 9))
    lam = torch.exp(torch.randn(n, generator=g, device=dev)) * cfg["depth"]
    N = torch.poisson(size[:, None] * lam[None, :], generator=g)
    eff = None
    if L == 2:
        c_1 = torch.binomial(N, psi, generator=g)
        return [c_1, N - c_1], None
    l = torch.randint(50, 301, (3, n), generator=g, device=dev).float()
    rlen, eh, jh = 76.0, 10.0, 2.0
    eff = torch.zeros(n, 6, device=dev)
    eff[:, 0] = l[1] + rlen - 2 * jh
    eff[:, 4] = rlen - 2 * jh
    eff[:, 2] = l[0] + l[2] - 2 * eh + 2 * jh
    eff[:, 5] = eff[:, 2]
    p1, p2, p3 = psi * eff[:, 0], (1 - psi) * eff[:, 4], eff[:, 5][None, :].expand_as(psi)
    tot = p1 + p2 + p3
    c_1 = torch.binomial(N, p1 / tot, generator=g)
    c_2 = torch.binomial(N - c_1, (p2 / (p2 + p3)).clamp_(0, 1), generator=g)
    return [c_1, c_2, N - c_1 - c_2], eff


def psi_delta_check(seed=11):
    """'PSI delta vs CPU ref' on BASELINE configs[0] (200 x 500, +1 covariate) after the WHOLE BRIE2.fit default
    schedule (6 x 166 Adam steps, fresh optimiser per stage, model_TFProb.py:234-241): HIP vs the CPU restatement in
    fp64, next to what the reference's own fp32 precision (the same restatement in fp32) does on that trajectory.
    The parity rule these numbers are held to is tests/util.py::psi_parity_assert; all configs and both default
    schedules are in profiles/psi_delta_r02.json."""
    from brie_amd import _capi
    from oracle.c_oracle import COracle
    from tests import util
    Nc, Ng, Kc = 200, 500, 1
    P = util.problem(Nc, Ng, Kc, 2, theta=3.0)
    o64 = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float64)
    o32 = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, seed)
    for n, lr in util.staged_schedule(1000):
        for o in (o64, o32):
            o.reset_optimizer()
            o.minimize(n, lr, 1)
        sh.reset_optimizer()
        sh.step(n, lr, 1, trace=False)
    d = np.abs(sh.read(_capi.PSI) - o64.Psi)
    d32 = np.abs(o32.Psi - o64.Psi)
    covered = (P["counts"][0] + P["counts"][1]) > 0
    sh.close()
    out = {"workload": "200x500 Kc=1, 996 staged steps (BRIE2.fit defaults), same init + noise stream, vs fp64 CPU oracle"}
    out.update(util.psi_parity_assert(d, d32, "bench psi check"))
    out["covered_entries"] = {"max": float(d[covered].max()), "frac_gt_1e-4": float((d[covered] > 1e-4).mean())}
    out["share_of_exceedances_with_zero_coverage"] = float(((d > 1e-4) & ~covered).sum() / max(1, (d > 1e-4).sum()))
    out["rule"] = "tests/util.py::psi_parity_assert (bounded by the fp32 oracle's own distance from fp64); holds"
    return out


def hbm_traffic(args, world, storage):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from the PMC counters, collected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE
    (KiB), FETCH_SIZE x2 on gfx950 for wide coalesced reads.  Measured LIVE by two child runs of this script (the
    same workload, a few steps); falls back to the committed profile of the same workload when rocprofv3 is not there."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    committed = {}
    try:
        committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(
            "%s_%s" % (args.config, {"u8/u16 per gene quad": "mixed"}.get(storage, storage))) or {}
    except (OSError, ValueError):
        pass
    eligible = world == 1 and args.mc == 1 and args.kc is None and not args.emulate_shard_of
    fallback = {"traffic": committed.get("hbm_bytes_per_launch") if eligible else None,
                "traffic_source": ("committed: " + committed.get("source", "profiles/pmc_traffic.json")) if (eligible and committed)
                else None}
    if args.no_pmc or world != 1 or not shutil.which("rocprofv3"):
        return fallback
    n_steps, kib = 4, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            tmp = tempfile.mkdtemp(prefix="brie_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", str(n_steps - 1), "--warmup", "1",
                   "--config", args.config, "--mc", str(args.mc), "--count-storage", args.count_storage,
                   "--no-cpu-baseline", "--no-psi-check"]
            if args.kc is not None:
                cmd += ["--kc", str(args.kc)]
            if args.rows_per_chunk:
                cmd += ["--rows-per-chunk", str(args.rows_per_chunk)]
            subprocess.run(cmd, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=600)
            total = 0.0
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "elbo_adam_step" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        total += float(r["Counter_Value"])
            shutil.rmtree(tmp, ignore_errors=True)
            if total <= 0:
                raise RuntimeError("no %s samples of elbo_adam_step" % counter)
            kib[counter] = total / n_steps
        return {"traffic": (2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024.0,
                "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (%d launches each; "
                                  "KiB; FETCH_SIZE x2 gfx950 correction)" % n_steps,
                "FETCH_SIZE_KiB_per_launch": kib["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": kib["WRITE_SIZE"]}
    except Exception as exc:
        return dict(fallback, traffic_error=repr(exc))


def end_of_fit_allgather(torch, dist, sh, cfg, args, seed, lr, rank, world, local_rank, dev, Xc, size, elapsed_local):
    """What fitBRIE does when the loop is over (brie_amd/models/wrap.py): every rank contributes the per-gene
    vectors [Wc_loc (Kc rows), intercept, sigma, loss_gene] of its gene shard and receives all Ng columns.
    Runs over (a) torch.distributed (RCCL when the backend is nccl) -- the path fitBRIE takes -- and (b) the
    library's own communicator (brie_comm_allgather of include/brie_amd.h, librccl called from libbrie_amd.so).
    Rank 0 then re-fits the first gene quad of every OTHER rank's shard alone and demands the gathered columns
    bit for bit (genes are independent, the noise stream is keyed by the global gene index)."""
    from brie_amd import _capi
    from brie_amd.sharding import GeneComm, gene_shard
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    n_rep = 8
    ranges = [gene_shard(Ng, r, world) for r in range(world)]
    local = np.concatenate([sh.read(_capi.WC_LOC).reshape(Kc, -1), sh.read(_capi.INTERCEPT).reshape(1, -1),
                            sh.read(_capi.SIGMA).reshape(1, -1), sh.loss_gene(n_rep).reshape(1, -1)], axis=0)
    comm = GeneComm(device=dev)
    comm.allgather_genes(local, Ng, ranges)                       # warm-up: communicator set-up, first-call costs
    dist.barrier()
    t0 = time.perf_counter()
    full = comm.allgather_genes(local, Ng, ranges)
    ms_torch = (time.perf_counter() - t0) * 1e3
    info = {"what": "per-gene vectors [Wc_loc x%d, intercept, sigma, loss_gene] of every rank -> all %d genes on every "
                    "rank; after the timed region" % (Kc, Ng),
            "backend": dist.get_backend(), "rccl_ranks": world, "bytes_per_rank": int(local.size * 4),
            "allgather_ms": ms_torch}
    # (b) the C-ABI communicator.  Never exercised between two GPUs by the build (1-GPU boxes only), so it runs in a
    # watchdog thread: if RCCL set-up or the collective does not come back, the line reports it and the run goes on.
    import threading

    def native_leg():
        try:
            nat = comm.native_comm(local_rank)
            if nat is None:
                info["native"] = "no native communicator on backend %s (RCCL needs one GPU per rank)" % dist.get_backend()
                return
            per = max(b - a for a, b in ranges)
            buf = np.zeros((local.shape[0], per), np.float32)
            buf[:, :local.shape[1]] = local
            nat.allgather(buf)
            t1 = time.perf_counter()
            g = nat.allgather(buf).reshape(world, local.shape[0], per)
            info["allgather_native_ms"] = (time.perf_counter() - t1) * 1e3
            full_nat = np.concatenate([g[r][:, :b - a] for r, (a, b) in enumerate(ranges)], axis=1)
            info["native_equals_torch"] = bool(np.array_equal(full_nat, full))
        except Exception as exc:                                  # reported, not fatal: the torch path already ran
            info["native_error"] = repr(exc)
    th = threading.Thread(target=native_leg, daemon=True)
    th.start()
    th.join(180.0)
    if th.is_alive():
        info["native_error"] = "brie_comm leg did not return within 180 s"
    # every rank's step time (max-over-ranks is what `value` uses)
    t = torch.tensor([elapsed_local / args.steps * 1e3], dtype=torch.float64,
                     device=dev if dist.get_backend() == "nccl" else "cpu")
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    info["ms_per_step_per_rank"] = [float(o.item()) for o in outs]
    if rank == 0:
      checked, mismatched = [], []
      try:
        for r in range(1, world):
            q = ranges[r][0]                                      # first quad of rank r's shard, refitted here alone
            c0 = (q // GEN_CHUNK) * GEN_CHUNK
            cnt, eff = gen_chunk(torch, dev, cfg, Xc, size, c0, min(c0 + GEN_CHUNK, Ng), seed)
            one = _capi.Shard(Nc, 4, Kc, n_layers=L, has_efflen=L == 3, seed=seed, device=local_rank, gene_offset=q)
            if args.count_storage == "f32":
                one.set_count_storage(1)
            for l in range(L):
                one.upload(_capi.COUNT1 + l, cnt[l][:, q - c0:q - c0 + 4].contiguous())
            one.add_pseudo_count(0.01)
            if L == 3:
                one.upload(_capi.EFFLEN, eff[q - c0:q - c0 + 4].cpu().numpy())
            if Kc:
                one.upload(_capi.XC, Xc)
            one.init_state()
            if args.rows_per_chunk:
                one.set_tiling(args.rows_per_chunk)
            one.step(args.warmup, lr, args.mc, trace=False)
            one.step(args.steps, lr, args.mc, trace=False)
            one.step(1, lr, args.mc)
            ref = np.concatenate([one.read(_capi.WC_LOC).reshape(Kc, -1), one.read(_capi.INTERCEPT).reshape(1, -1),
                                  one.read(_capi.SIGMA).reshape(1, -1), one.loss_gene(n_rep).reshape(1, -1)], axis=0)
            one.close()
            (checked if np.array_equal(full[:, q:q + 4], ref) else mismatched).append(int(q))
        # a mismatch is reported in the line (and fails the run under BRIE_BENCH_STRICT=1); it must not cost the
        # scaling measurement its number
        info["recomputed_on_rank0"] = {"first_gene_of_quads": checked + mismatched, "bit_identical": not mismatched,
                                       "mismatched_quads": mismatched}
      except Exception as exc:
        info["recomputed_on_rank0"] = {"error": repr(exc)}
      if os.environ.get("BRIE_BENCH_STRICT") and not info["recomputed_on_rank0"].get("bit_identical", False):
        raise AssertionError("gathered per-gene vectors differ from rank 0's recomputation: %r" % (info["recomputed_on_rank0"],))
    dist.barrier()
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--mc", type=int, default=1, help="MC_size (API default 1, CLI default 3)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the config's genes are sharded over ranks (BASELINE configs[3]); "
                         "weak: every rank fits the whole config")
    ap.add_argument("--kc", type=int, default=None, help="override the config's number of cell covariates (experiments)")
    ap.add_argument("--rows-per-chunk", type=int, default=0)
    ap.add_argument("--emulate-shard-of", type=int, default=0,
                    help="experiments: with --gpus 1, run only rank 0's gene shard of an N-way split (per-rank step "
                         "time of the strong-scaling run without N GPUs); the line is labelled and is not a result")
    ap.add_argument("--count-storage", default="auto", choices=["auto", "f32"],
                    help="auto: integer counts <= 255 are kept as u8 in HBM (bit-identical results); f32: as uploaded")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic live (N=1 only)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the second timed leg with fp32 count storage")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the PCIe-inclusive leg: one whole BRIE2.fit + BRIE_RV from host arrays to host results")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psi-check", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    # RCCL / device-tensor sharing between the ranks needs dmabuf IPC on this driver (exported on the pool already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    from brie_amd import _capi
    from brie_amd.sharding import gene_shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("BRIE_BENCH_SINGLE_DEVICE"):     # testing aid: several ranks share GPU 0 (with gloo, see below)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:       # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ.get("BRIE_BENCH_SINGLE_DEVICE"):  # NCCL/RCCL refuses two ranks on one device: gloo for the dry run
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = dict(CONFIGS[args.config])
    if args.kc is not None:
        cfg["Kc"] = args.kc
        cfg["desc"] += " [Kc overridden to %d]" % args.kc
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    if args.emulate_shard_of:
        assert world == 1
        g0, g1 = gene_shard(Ng, 0, args.emulate_shard_of)
        cfg["desc"] += " [ONLY shard 0 of %d: per-rank dry run]" % args.emulate_shard_of
    elif args.scaling == "strong":
        g0, g1 = gene_shard(Ng, rank, world)
    else:
        g0, g1 = 0, Ng
    ng = g1 - g0
    seed = 20240617 + {"c1": 1, "c2": 2, "c3": 3, "c5": 5}[args.config]

    # ---- synthetic inputs, generated on the device, resident before the timed region
    t_gen = time.time()
    gx = torch.Generator(device=dev)
    gx.manual_seed(seed)
    Xc = torch.zeros(Nc, Kc, device=dev)
    if Kc:
        Xc[:, 0] = (torch.rand(Nc, generator=gx, device=dev) < 0.5).float()
        if Kc > 1:
            Xc[:, 1:] = torch.randn(Nc, Kc - 1, generator=gx, device=dev)
    size = torch.exp(0.5 * torch.randn(Nc, generator=gx, device=dev))
    layers = [torch.empty(Nc, ng, device=dev) for _ in range(L)]
    eff_all = torch.zeros(ng, 6, device=dev) if L == 3 else None
    c0 = (g0 // GEN_CHUNK) * GEN_CHUNK
    while c0 < g1:
        c1 = min(c0 + GEN_CHUNK, Ng)
        cnt, eff = gen_chunk(torch, dev, cfg, Xc, size, c0, c1, seed)
        a, b = max(c0, g0), min(c1, g1)
        for l in range(L):
            layers[l][:, a - g0:b - g0] = cnt[l][:, a - c0:b - c0]
        if eff is not None:
            eff_all[a - g0:b - g0] = eff[a - c0:b - c0]
        c0 = c1
    torch.cuda.synchronize()

    sh = _capi.Shard(Nc, ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed, device=local_rank, gene_offset=g0)
    if args.count_storage == "f32":
        sh.set_count_storage(1)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, layers[l])           # device -> device
    sh.add_pseudo_count(0.01)
    if L == 3:
        sh.upload(_capi.EFFLEN, eff_all.cpu().numpy())
    if Kc:
        sh.upload(_capi.XC, Xc)
    # host sample for the CPU baselines: three reference-sized gene batches (model_wrap.py:242), at least 64 genes
    n_keep = min(ng, max(64, 3 * min(int(math.ceil(500000 / float(Nc))), Ng)))
    sample_layers = [layers[l][:, :n_keep].cpu().numpy() for l in range(L)] if rank == 0 else None
    eff_host = eff_all[:n_keep].cpu().numpy() if (L == 3 and rank == 0) else None
    # one gene quad in the middle of the shard: the oracle re-runs it over ALL cells after the timed region
    q0 = (ng // 2) // 4 * 4
    quad_layers = [layers[l][:, q0:q0 + 4].cpu().numpy() for l in range(L)] if rank == 0 else None
    quad_eff = eff_all[q0:q0 + 4].cpu().numpy() if (L == 3 and rank == 0) else None
    Xc_host = Xc.cpu().numpy()
    e2e_inputs = None
    if rank == 0 and world == 1 and not (args.no_e2e or args.pmc_child or args.emulate_shard_of):
        e2e_inputs = [x.cpu().numpy() for x in layers]          # the API hands over HOST buffers
    del layers
    torch.cuda.empty_cache()
    sh.init_state()
    if args.rows_per_chunk:
        sh.set_tiling(args.rows_per_chunk)
    sh.synchronize()
    t_gen = time.time() - t_gen

    lr = 0.005
    sh.step(args.warmup, lr, args.mc, trace=False)
    sh.synchronize()
    sh.profile_enable(True)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    sh.step(args.steps, lr, args.mc, trace=False)
    sh.synchronize()
    fence()
    elapsed = time.perf_counter() - t0
    if args.pmc_child:                       # run under rocprofv3 --pmc by the parent bench: the launches are all it needs
        sh.close()
        return
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms, n_launch = sh.profile_read()
    stream_gbs = None
    if rank == 0:
        # measured ceiling of THIS box: a kernel with the same 16-B read/write stream mix and no arithmetic
        n_read = L + 6
        stream_gbs = max(_capi.calibrate_stream(n_read, 6, 1 << 30, iters=5, device=local_rank, nt=nt)
                         for nt in (False, True, False, True))
    sh.profile_enable(False)
    last = sh.step(1, lr, args.mc)                       # one traced step: loss must be finite
    assert np.isfinite(last).all(), last
    psi_dev = sh.read(_capi.PSI) if (rank == 0 and world == 1 and not args.no_psi_check) else None
    f32_leg = None
    storage_main, storage_bytes_main = sh.count_storage, sh.step_storage_bytes()
    if rank == 0 and world == 1 and not args.no_f32_leg and storage_main != "f32":
        # the same kernel on the fp32 layers as uploaded (compact storage is reported separately)
        sh.set_count_storage(1)
        sh.step(args.warmup, lr, args.mc, trace=False)
        sh.profile_enable(True)
        sh.step(args.steps, lr, args.mc, trace=False)
        ms32, n32 = sh.profile_read()
        sh.profile_enable(False)
        f32_leg = {"avg_kernel_ms": ms32 / max(n32, 1), "storage_bytes_per_launch": sh.step_storage_bytes()}
    total_elems = Nc * (Ng if args.scaling == "strong" else Ng * world)
    if args.emulate_shard_of:
        total_elems = Nc * ng
    value = args.steps * total_elems / elapsed

    # ---- N > 1: the end-of-fit exchange of a gene-sharded fit (BASELINE configs[3]: "RCCL weight all-gather"),
    # untimed by `value` (there is no collective inside the optimisation loop) but executed, checked and reported
    gather_info = None
    # (BRIE_BENCH_FORCE_GATHER=1: run this leg with a world of ONE rank too -- the only way to put the nccl branch and
    #  the C-ABI communicator through RCCL on a 1-GPU box)
    if dist is not None and args.scaling == "strong" and (world > 1 or os.environ.get("BRIE_BENCH_FORCE_GATHER")):
        gather_info = end_of_fit_allgather(torch, dist, sh, cfg, args, seed, lr, rank, world, local_rank, dev, Xc, size,
                                           elapsed_local)

    out = None
    if rank == 0:
        alg_bytes = sh.step_algorithmic_bytes()
        avg_ms = kern_ms / max(n_launch, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "definition": "achieved = ALGORITHMIC bytes (48 + 4 L per element, fp32 model of SURVEY 8d) / average "
                              "launch time of the dominant kernel; what the kernel physically moves is hbm_rate_GBs",
                "kernel": "elbo_adam_step<Kc=%d>" % Kc, "avg_kernel_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": n_launch,
                "count_storage": storage_main, "storage_bytes_per_launch": storage_bytes_main,
                # the physical HBM rate: bytes the current storage moves (integer counts are kept as u8 / u16,
                # bit-identical results) over the same time -- below `achieved` by construction
                "hbm_rate_GBs": storage_bytes_main / (avg_ms * 1e-3) / 1e9,
                "hbm_frac_of_peak": storage_bytes_main / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # ... and against what a pure streaming kernel with the same stream mix and access order reaches
                # on THIS box (brie_calibrate_stream)
                "measured_stream_ceiling_GBs": stream_gbs,
                "frac_of_measured_stream_ceiling": storage_bytes_main / (avg_ms * 1e-3) / 1e9 / stream_gbs}
        if f32_leg is not None:
            t32 = f32_leg["avg_kernel_ms"] * 1e-3
            roof["f32_count_storage"] = dict(f32_leg, achieved=alg_bytes / t32 / 1e9, frac=alg_bytes / t32 / 1e9 / HBM_PEAK_GBS,
                                             note="same kernel, counts kept as the uploaded fp32 layers: storage bytes "
                                                  "= algorithmic bytes")
        out = {
            "metric": "ELBO iterations/sec (cells x genes)",
            "value": value,
            "unit": "cell*gene*iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "iterations_per_s": args.steps / elapsed,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["desc"], "Nc": Nc, "Ng": Ng, "Kc": Kc, "count_layers": L,
                       "MC_size": args.mc, "genes_per_rank": ng, "parallelism": "gene-shard x%d" % world,
                       "inputs": "generated on device, resident in HBM (%.1f s, untimed)" % t_gen},
            "roofline": roof,
        }
        if gather_info is not None:
            out["allgather"] = gather_info
    sh.close()
    del sh
    torch.cuda.empty_cache()

    if rank == 0:
        out["roofline"].update(hbm_traffic(args, world, storage_main))
    if e2e_inputs is not None:
        # Secondary, PCIe-inclusive figure (never `value`): the reference's unit of work through the public API --
        # host count layers in, BRIE2.fit with the default schedule (996 staged steps + 500-draw loss_gene), host
        # Psi / Z_std / Psi95CI / Z_loc out (model_wrap.py:138-146) -- upload, compaction and read-back included.
        import brie_amd
        t0 = time.perf_counter()
        mdl = brie_amd.BRIE2(Nc, Ng, Kc=Kc, effLen=eff_all.cpu().numpy() if L == 3 else None, seed=seed, device=local_rank)
        mdl.fit(e2e_inputs, Xc=Xc_host, min_iter=1000, max_iter=1000, MC_size=args.mc, pseudo_count=0.01, verbose=False)
        rv = brie_amd.BRIE_RV(mdl)
        total = time.perf_counter() - t0
        assert np.isfinite(rv.Psi).all()
        out["pcie_inclusive"] = {"it_per_s_pcie_inclusive": 996 / total, "total_s": total, "steps": 996,
                                 "breakdown_s": {k: v for k, v in mdl.timing.items() if k.endswith("_s")},
                                 "what": "host numpy count layers -> BRIE2.fit (6 x 166 steps, 500-draw loss_gene) -> "
                                         "BRIE_RV with Psi, Z_std, Psi95CI, Z_loc on the host; result matrices stream "
                                         "out while loss_gene runs"}
        mdl.close()
        del e2e_inputs, rv

    if rank == 0 and world == 1 and not args.no_psi_check and q0 + 4 <= ng:
        # PSI delta ON THE HEADLINE WORKLOAD: genes are independent and the noise stream is keyed by the global gene
        # index, so the CPU oracle run on one gene quad over all Nc cells is an exact reference for those genes
        from oracle.brie_oracle import OracleBRIE2, add_pseudo_count
        n_total = args.warmup + args.steps + 1
        o = OracleBRIE2(Nc, 4, Kc, effLen=quad_eff, seed=seed, gene_offset=g0 + q0, dtype=np.float64)
        o.minimize(add_pseudo_count(quad_layers), Xc_host, n_total, lr, args.mc)
        d = np.abs(psi_dev[:, q0:q0 + 4] - o.Psi)
        out["psi_delta_headline_workload"] = {
            "what": "genes %d..%d of the timed %s run, all %d cells, %d Adam steps vs the fp64 CPU oracle"
                    % (g0 + q0, g0 + q0 + 3, args.config, Nc, n_total),
            "max": float(d.max()), "p99": float(np.percentile(d, 99))}
    if rank == 0 and world == 1:
        if not args.no_psi_check:
            out["psi_delta_vs_cpu_ref"] = psi_delta_check()
        if not args.no_cpu_baseline:
            from oracle.brie_oracle_torch import time_reference_shape
            import torch as _t
            cores = min(6, os.cpu_count() or 1)          # reference default --nproc 6 (bin/quant.py:183)
            n_gene = min(int(math.ceil(500000 / float(Nc))), sample_layers[0].shape[1])
            nb = max(1, min(3, sample_layers[0].shape[1] // n_gene))

            def counts_fn(a, b):
                return [np.ascontiguousarray(c[:, a:b]) for c in sample_layers]
            # calibrate so the sample costs ~cpu_seconds
            eps_s, el = time_reference_shape(Nc, n_gene * nb, counts_fn, Xc_host, 1, 3, args.mc, threads=cores,
                                             warmup_steps=1)
            n_steps = int(max(5, min(2000, args.cpu_seconds * eps_s / (Nc * n_gene * nb))))
            eps_s, el = time_reference_shape(Nc, n_gene * nb, counts_fn, Xc_host, nb, n_steps, args.mc, threads=cores)
            out["cpu_baseline"] = {
                "value": eps_s, "unit": "cell*gene*iterations/s", "cores": cores, "kind": "port",
                "sample": "%d reference-sized gene batches (%d genes x %d cells each, model_wrap.py:242) x %d Adam "
                          "steps, eager torch-CPU autograd restatement (TF absent), %.1f s"
                          % (nb, n_gene, Nc, n_steps, el),
                "gpu_over_cpu": value / eps_s,
            }
            # second, separately labelled baseline: the same algorithm as ONE fused
            # C/OpenMP pass (oracle/brie_oracle.c) on all host cores
            try:
                from oracle.c_oracle import COracle
                from oracle.brie_oracle import add_pseudo_count
                co = COracle(add_pseudo_count(sample_layers), Xc_host, effLen=None if L == 2 else eff_host, seed=seed)
                co.minimize(2, 0.005, args.mc)
                t0 = time.perf_counter()
                co.minimize(3, 0.005, args.mc)
                per_step = (time.perf_counter() - t0) / 3
                n_fused = int(max(3, min(500, 0.5 * args.cpu_seconds / per_step)))
                t0 = time.perf_counter()
                co.minimize(n_fused, 0.005, args.mc)
                el = time.perf_counter() - t0
                fused = n_fused * Nc * sample_layers[0].shape[1] / el
                out["cpu_baseline_fused"] = {
                    "value": fused, "unit": "cell*gene*iterations/s", "cores": co.threads(), "kind": "port-fused",
                    "sample": "%d genes x %d cells x %d Adam steps, fused single-pass C/OpenMP restatement (oracle/brie_oracle.c; scalar libm, "
                              "noise stream evaluated in double -- not vectorised), %.1f s"
                              % (sample_layers[0].shape[1], Nc, n_fused, el),
                    "gpu_over_cpu": value / fused}
            except Exception as exc:                      # gcc / OpenMP missing: the eager baseline above stands
                out["cpu_baseline_fused"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

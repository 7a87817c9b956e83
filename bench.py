#!/usr/bin/env python
"""bench.py -- ELBO iterations/s of the brie-quant hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N>1: one process per GPU under torch.distributed.run (RCCL).  Started WITHOUT a launcher (no WORLD_SIZE in the
environment) the script starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself -- as a child
process, before anything in this process has touched the GPU -- relays rank 0's JSON line and returns its exit code.

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

# the C oracle legs (OpenMP) run between device calls: idle OpenMP workers must sleep, not spin against the HIP runtime's
# threads on the few host cores a GPU box grants (set before libgomp is loaded; tests/conftest.py does the same)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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


def make_inputs(torch, dev, cfg, g0, g1, seed):
    """Xc, per-cell size factors, the L count layers of genes [g0, g1) and their effLen rows, all on the device.
    Seeded per GEN_CHUNK genes of the WHOLE problem, so a gene has the same data in every sharding."""
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    ng = g1 - g0
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
    return Xc, size, layers, eff_all


_T0 = time.time()


def note(msg):
    """Progress on stderr (stdout carries the one JSON line): which leg is running, seconds since start."""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write("[bench %6.1f s] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()


def config_seed(name):
    return 20240617 + {"c1": 1, "c2": 2, "c3": 3, "c5": 5}[name]


def psi_delta_check(seed=11):
    """'PSI delta vs CPU ref' on BASELINE configs[0] (200 x 500, +1 covariate) after the WHOLE BRIE2.fit default
    schedule (6 x 166 Adam steps, fresh optimiser per stage, model_TFProb.py:234-241): HIP vs the CPU restatement in
    fp32 (o32, the reference's precision), judged by what a SECOND fp32 CPU evaluation of the same algorithm (o32b:
    oracle/brie_oracle.c -DBRIE_ORACLE_B) does against that same o32 run -- tests/util.py::psi_null_rule.  The twenty
    cases on gene samples of configs[1] / configs[2] and both default schedules are in profiles/psi_null_r04.json."""
    from brie_amd import _capi
    from oracle.c_oracle import COracle
    from tests import util
    Nc, Ng, Kc = 200, 500, 1
    P = util.problem(Nc, Ng, Kc, 2, theta=3.0)
    o32 = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float32)
    o32b = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float32, variant_b=True)
    sh = util.device_shard(P, Nc, Ng, Kc, seed)
    for n, lr in util.staged_schedule(1000):
        for o in (o32, o32b):
            o.reset_optimizer()
            o.minimize(n, lr, 1)
        sh.reset_optimizer()
        sh.step(n, lr, 1, trace=False)
    psi_h = sh.read(_capi.PSI)
    d = np.abs(psi_h - o32.Psi)
    dn = np.abs(o32b.Psi.astype(np.float64) - o32.Psi)
    h = util.gene_summaries(psi_h, o32.Psi, util.run_params(sh), util.run_params(o32))
    nul = util.gene_summaries(o32b.Psi, o32.Psi, util.run_params(o32b), util.run_params(o32))
    sh.close()
    out = {"workload": "200x500 Kc=1, 996 staged steps (BRIE2.fit defaults), same init + noise stream, vs the fp32 CPU oracle",
           "max": float(d.max()), "p99": float(np.percentile(d, 99)), "frac_gt_1e-4": float((d > 1e-4).mean()),
           "second_fp32_cpu_evaluation_vs_the_same_oracle": {"max": float(dn.max()), "p99": float(np.percentile(dn, 99)),
                                                             "frac_gt_1e-4": float((dn > 1e-4).mean())}}
    rep = util.psi_null_rule(h, nul, "bench psi check")            # raises AssertionError when violated
    out["displaced_genes"], out["clustered_genes"] = rep["displaced_genes"], rep["clustered_genes"]
    out["quiet_genes"] = rep.get("quiet_genes")
    out["rule"] = ("tests/util.py::psi_null_rule (moved genes, and entries of the quiet genes, bounded by what a second fp32 CPU "
                   "evaluation does against the same fp32 oracle); holds")
    return out


def hbm_traffic(args, world, storage):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from the PMC counters, collected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE
    (KiB), FETCH_SIZE x2 on gfx950 for wide coalesced reads.  Measured LIVE by two child runs of this script (the
    same workload, a few steps; with N > 1 ranks: rank 0's gene shard, on rank 0's GPU, while the other ranks wait);
    falls back to the committed profile of the same workload when rocprofv3 is not there."""
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
    shard_of = args.emulate_shard_of or (world if (world > 1 and args.scaling == "strong") else 0)
    eligible = world == 1 and args.mc == 1 and args.kc is None and not shard_of
    fallback = {"traffic": committed.get("hbm_bytes_per_launch") if eligible else None,
                "traffic_source": ("committed: " + committed.get("source", "profiles/pmc_traffic.json")) if (eligible and committed)
                else None}
    if args.no_pmc or not shutil.which("rocprofv3"):
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
            if shard_of:
                cmd += ["--emulate-shard-of", str(shard_of)]
            subprocess.run(cmd, check=True, cwd="/tmp", env=dict(standalone_env(), TMPDIR="/tmp", BRIE_PLACEMENT_TRIES="1"), stdout=subprocess.DEVNULL,
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
                "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command%s (%d launches each; "
                                  "KiB; FETCH_SIZE x2 gfx950 correction)"
                                  % (" on rank 0's shard (1 of %d)" % shard_of if shard_of else "", n_steps),
                "FETCH_SIZE_KiB_per_launch": kib["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": kib["WRITE_SIZE"]}
    except Exception as exc:
        return dict(fallback, traffic_error=repr(exc))


def usable_cores():
    """Host cores this process can really use: its affinity mask, capped by the CPU quota of its cgroup (a GPU box may
    show 256 CPUs in the mask and grant a handful)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                   # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(math.ceil(float(q) / float(per)))))
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    return n


def per_gene_vectors(sh, Kc, n_rep):
    from brie_amd import _capi
    return np.concatenate([sh.read(_capi.WC_LOC).reshape(Kc, sh.Ng), sh.read(_capi.INTERCEPT).reshape(1, -1),
                           sh.read(_capi.SIGMA).reshape(1, -1), sh.loss_gene(n_rep).reshape(1, -1)], axis=0)


def end_of_fit_allgather(torch, dist, sh, cfg, args, seed, lr, rank, world, local_rank, dev, Xc, size):
    """What fitBRIE does when the loop is over (brie_amd/models/wrap.py): every rank contributes the per-gene
    vectors [Wc_loc (Kc rows), intercept, sigma, loss_gene] of its gene shard and receives all Ng columns.
    This leg goes over torch.distributed (RCCL when the backend is nccl); the library's own communicator
    (brie_comm_allgather, what fitBRIE uses when it exists) is exercised by native_allgather_leg at the very end.
    Rank 0 then re-fits the first gene quad of every OTHER rank's shard alone and demands the gathered columns
    bit for bit (genes are independent, the noise stream is keyed by the global gene index)."""
    from brie_amd import _capi
    from brie_amd.sharding import GeneComm, gene_shard
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    n_rep = 8
    ranges = [gene_shard(Ng, r, world) for r in range(world)]
    local = per_gene_vectors(sh, Kc, n_rep)
    comm = GeneComm(device=dev)
    comm.allgather_genes(local, Ng, ranges, native=False)         # warm-up: communicator set-up, first-call costs
    dist.barrier()
    t0 = time.perf_counter()
    full = comm.allgather_genes(local, Ng, ranges, native=False)
    ms_torch = (time.perf_counter() - t0) * 1e3
    info = {"what": "per-gene vectors [Wc_loc x%d, intercept, sigma, loss_gene] of every rank -> all %d genes on every "
                    "rank; after the timed region" % (Kc, Ng),
            "backend": dist.get_backend(), "rccl_ranks": world if dist.get_backend() == "nccl" else 0,
            "bytes_per_rank": int(local.size * 4), "allgather_ms": ms_torch}
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
            one.step(args.steps, lr, args.mc, trace=False)       # the timed pass ...
            one.step(args.steps, lr, args.mc, trace=False)       # ... the unprofiled pass ...
            one.step(1, lr, args.mc)                             # ... and the traced step of the shard it mirrors
            ref = per_gene_vectors(one, Kc, n_rep)
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
    return info, (comm, local, ranges, full)


def native_allgather_leg(info, state, local_rank, world, timeout_s=180.0):
    """The same gather through the C-ABI communicator (brie_comm_* of include/brie_amd.h: librccl called from
    libbrie_amd.so; the path fitBRIE takes when the process group runs on RCCL).  It has never run between two GPUs on
    the build's 1-GPU boxes, so it is the LAST thing a rank does -- every timed number is assembled before -- and it
    runs under a watchdog: returns False when RCCL set-up or the collective did not come back in `timeout_s`, in
    which case the caller prints its line and leaves without touching the process group again.
    BRIE_BENCH_NATIVE_COMM=0 skips the leg."""
    import threading
    comm, local, ranges, full = state
    if os.environ.get("BRIE_BENCH_NATIVE_COMM", "1") == "0":
        info["native"] = "skipped (BRIE_BENCH_NATIVE_COMM=0)"
        return True
    shared, info = info, {}            # the worker writes into a private dict, merged below only once it has come back

    def leg():
        try:
            nat = comm.native_comm(local_rank)
            if nat is None:
                info["native"] = "no native communicator on backend %s (RCCL needs one GPU per rank)" % comm.backend
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
    th = threading.Thread(target=leg, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        shared["native_error"] = "brie_comm leg did not return within %.0f s" % timeout_s
        return False
    shared.update(info)
    return True


LAUNCH_ENV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                   "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                   "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE",
                   "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "OMP_NUM_THREADS")


def launcher_command(argv, n_gpus, port):
    """The command `python bench.py --gpus N ...` turns into when no launcher started it: the driver's own launch line
    (one rank per GPU of ONE node, rendezvous on 127.0.0.1), with this script's arguments relayed unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, n_gpus):
    """Start the N ranks as a CHILD process tree and return its exit code.  Called before this process has imported
    torch.cuda or made any HIP call: a process that has initialised the GPU must never exec or be replaced."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(launcher_command(argv, n_gpus, port), env=env)


def standalone_env():
    """Environment of a single-process child of a rank (the rocprofv3 --pmc runs): the launcher's variables removed."""
    return {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV_KEYS}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 20; 200 / 500 for the sub-millisecond configs c2 / c1)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default: 3; 50 / 100 for c2 / c1)")
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
                    help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic live")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the second timed leg with fp32 count storage")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the PCIe-inclusive leg: one whole BRIE2.fit + BRIE_RV from host arrays to host results")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-precondition", action="store_true",
                    help="time the K steps directly after the W warm-up steps (no clock preconditioning by probe launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psi-check", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args(argv)
    # A step of configs[1] takes 0.45 ms and one of configs[0] 12 us: 3 + 20 of them are over before the GPU has left its idle
    # clocks (call r4au: 0.480 ms per launch in such a window, 0.435 ms for the same kernel on the same handle after 45 steps,
    # 0.440 inside the fit) -- the sub-millisecond configs get a longer warm-up and timed region by default.
    short = {"c1": (500, 100), "c2": (200, 50)}.get(args.config, (20, 3))
    if args.steps is None:
        args.steps = short[0]
    if args.warmup is None:
        args.warmup = short[1]
    # RCCL / device-tensor sharing between the ranks needs dmabuf IPC on this driver (exported on the pool already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus < 1:
        raise SystemExit("--gpus %d" % args.gpus)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves (child processes; nothing here has touched the GPU yet)
        raise SystemExit(self_launch(argv, args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("BRIE_BENCH_ECHO_RANK"):          # launcher test (CPU): what reached the ranks, no GPU touched
        echo = {"echo": True, "world": world, "argv": argv, "master_addr": os.environ.get("MASTER_ADDR"),
                "local_world": os.environ.get("LOCAL_WORLD_SIZE")}
        if os.environ["BRIE_BENCH_ECHO_RANK"] == "shards" and world > 1:
            # ... and the gene range every rank would fit, gathered over gloo (tests/test_world8_cpu.py: world 8 without GPUs)
            import torch.distributed as dist
            from brie_amd.sharding import gene_shard
            dist.init_process_group("gloo", rank=rank, world_size=world)
            Ng = CONFIGS[args.config]["Ng"]
            mine = {"rank": rank, "local_rank": local_rank,
                    "genes": list(gene_shard(Ng, rank, world)) if args.scaling == "strong" else [0, Ng]}
            box = [None] * world
            dist.all_gather_object(box, mine)
            dist.destroy_process_group()
            echo["shards"] = box
        if rank == 0:
            print(json.dumps(echo))
        return

    import torch
    from brie_amd import _capi
    from brie_amd.sharding import gene_shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    single_device = bool(os.environ.get("BRIE_BENCH_SINGLE_DEVICE"))   # testing aid: several ranks share GPU 0 (gloo)
    if single_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU: %d visible, --gpus %d (BRIE_BENCH_SINGLE_DEVICE=1 puts every rank on "
                         "GPU 0 over gloo for a dry run)" % (rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = side = None
    if world > 1 or "RANK" in os.environ:       # launched by torch.distributed.run: one rank per GPU over RCCL
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        patience = datetime.timedelta(minutes=30)        # rank 0 does minutes of host work while the others wait
        if single_device:                                # NCCL/RCCL refuses two ranks on one device: gloo for the dry run
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=patience)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=patience)
            # waits for rank 0's host-side legs (CPU baseline, counter runs) go through sockets, not through a
            # collective spinning on the GPU
            side = dist.new_group(backend="gloo", timeout=patience)

    def wait_for_rank0():
        if dist is not None:
            dist.barrier(group=side)

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
    if ng <= 0:
        raise SystemExit("rank %d holds no genes: %d genes cannot be sharded over %d ranks" % (rank, Ng, world))
    seed = config_seed(args.config)

    # ---- synthetic inputs, generated on the device, resident before the timed region
    note("generating inputs: %s" % cfg["desc"])
    t_gen = time.time()
    Xc, size, layers, eff_all = make_inputs(torch, dev, cfg, g0, g1, seed)

    sh = _capi.Shard(Nc, ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed, device=local_rank, gene_offset=g0)
    # the library's default since round 6 is ONE round of the placement search (4 sets: at most 3 further copies of the
    # streamed arrays held meanwhile); a caller that owns its GPU may ask for more -- this one does and says so in the line
    placement_sets = int(os.environ.get("BRIE_BENCH_PLACEMENT_SETS", "8"))
    if hasattr(sh, "placement_configure"):
        sh.placement_configure(max_sets=placement_sets)
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
    if rank == 0 and world == 1 and not (args.no_e2e or args.pmc_child):
        e2e_inputs = [x.cpu().numpy() for x in layers]          # the API hands over HOST buffers
    del layers
    torch.cuda.empty_cache()
    sh.init_state()
    if args.rows_per_chunk:
        sh.set_tiling(args.rows_per_chunk)
    sh.synchronize()
    t_gen = time.time() - t_gen

    lr = 0.005
    note("warm-up (the first step also compacts the counts and searches the placement)")
    t_w = time.perf_counter()
    sh.step(args.warmup, lr, args.mc, trace=False)
    sh.synchronize()
    t_w = time.perf_counter() - t_w
    note("placement: %r" % (sh.placement_info(),))
    # The GPU leaves its idle clocks over tens of milliseconds (call r4aw: the same kernel 8 % slower in a 10-ms window after
    # host work than 45 steps later).  A fit is 996 steps, so the steady state is what counts: when the W warm-up steps were
    # shorter than that, the library's effect-free placement probe (the step's traffic, no arithmetic, state bits written
    # back unchanged) keeps the device busy for ~0.2 s before the timed region.  The W + K steps themselves are as asked.
    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    precondition = {"probe_launches": 0, "seconds": 0.0}
    plain_steps = 0                              # Adam steps of the "plain" leg below (the oracle re-run of a gene quad counts them)
    if t_w < 0.25 and not args.pmc_child and not args.no_precondition:
        # ... but first the PLAIN figure, measured the way rounds 1 - 3 measured theirs (W warm-up steps, then K timed steps,
        # nothing in between), so that round-over-round comparisons stay like for like; --no-precondition makes
        # that the headline again
        plain = None
        if world == 1:                           # N = 1 only: with N > 1 rank 0 re-fits other ranks' genes step for step (allgather
            sh.profile_enable(True)              # check) and must know how many steps every rank took
            torch.cuda.synchronize()
            t_pl = time.perf_counter()
            sh.step(args.steps, lr, args.mc, trace=False)
            sh.synchronize()
            t_pl = time.perf_counter() - t_pl
            k_pl, n_pl = sh.profile_read()
            plain_steps = args.steps
            plain = {"ms_per_step": round(t_pl / args.steps * 1e3, 5), "avg_kernel_ms": round(k_pl / max(n_pl, 1), 5),
                     "frac": round(sh.step_algorithmic_bytes() / (k_pl / max(n_pl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "what": "the same K steps timed directly after the W warm-up steps, before any preconditioning: "
                             "the figure of rounds 1 - 3"}
        n_probe = int(min(4000, max(8, 0.2 * 6e12 / max(sh.step_storage_bytes(), 1))))
        t_p = time.perf_counter()
        sh.placement_probe(n_probe)
        sh.synchronize()
        precondition = {"probe_launches": n_probe, "seconds": round(time.perf_counter() - t_p, 4),
                        "what": "effect-free placement probe launches between the warm-up steps and the timed region (clocks)",
                        "plain_without_preconditioning": plain}
        note("clock preconditioning: %r" % (precondition,))
    sh.profile_enable(True)

    fence()
    t0 = time.perf_counter()
    sh.step(args.steps, lr, args.mc, trace=False)
    sh.synchronize()
    fence()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.3f ms per step" % (elapsed / args.steps * 1e3))
    if args.pmc_child:                       # run under rocprofv3 --pmc by the parent bench: the launches are all it needs
        sh.close()
        return
    elapsed_local = elapsed

    # ---- every timed number of every rank, assembled before anything else runs
    kern_ms, n_launch = sh.profile_read()
    sh.profile_enable(False)
    storage_main, storage_bytes_main, alg_bytes = sh.count_storage, sh.step_storage_bytes(), sh.step_algorithmic_bytes()
    mine = [elapsed_local / args.steps * 1e3, kern_ms / max(n_launch, 1), float(alg_bytes), float(storage_bytes_main),
            float(ng)]
    per_rank = [mine]
    if dist is not None:
        on_dev = dist.get_backend() == "nccl"
        t = torch.tensor(mine, dtype=torch.float64, device=dev if on_dev else "cpu")
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        per_rank = [[float(x) for x in o.cpu().tolist()] for o in outs]
        elapsed = max(r[0] for r in per_rank) * args.steps / 1e3            # MAX over ranks
    total_elems = Nc * (Ng if args.scaling == "strong" else Ng * world)
    if args.emulate_shard_of:
        total_elems = Nc * ng
    value = args.steps * total_elems / elapsed

    # the same K steps once more WITHOUT the two HIP events per launch the roofline needs: on a launch-bound problem
    # (configs[0]: two dependent ~6-us kernels per step) the event packets are a third of what the timed region measures
    t0 = time.perf_counter()
    sh.step(args.steps, lr, args.mc, trace=False)
    sh.synchronize()
    ms_unprofiled = (time.perf_counter() - t0) / args.steps * 1e3
    # small inputs (configs[0]): brie_step runs its K steps as ONE launch where that measured faster (never under
    # the per-launch events of the timed region above) -- both paths timed once more, same handle, same steps
    fusion = None
    if world == 1 and hasattr(sh, "step_fusion_info") and sh.step_fusion_info()["launches"] > 0:    # (N > 1: rank 0 replays the steps)
        fusion = {"launches_so_far": sh.step_fusion_info()}
        for mode, key in ((0, "two_launches_per_step_ms"), (-1, "one_launch_per_call_ms")):
            sh.set_step_fusion(mode)
            sh.step(args.steps, lr, args.mc, trace=False)
            sh.synchronize()
            t0 = time.perf_counter()
            sh.step(args.steps, lr, args.mc, trace=False)
            sh.synchronize()
            fusion[key] = (time.perf_counter() - t0) / args.steps * 1e3
        fusion["what"] = "brie_set_step_fusion 0 / automatic on the timed handle; state bit-identical either way (tests)"
    last = sh.step(1, lr, args.mc)                       # one traced step: loss must be finite
    assert np.isfinite(last).all(), last
    psi_quad = sh.read(_capi.PSI)[:, q0:q0 + 4].copy() if (rank == 0 and not args.no_psi_check and q0 + 4 <= ng) else None
    placement = sh.placement_info() if rank == 0 else None
    mc_other = None
    if rank == 0 and world == 1 and not args.no_f32_leg and not args.emulate_shard_of:
        # the other default Monte-Carlo sample size: API default 1 (model_TFProb.py:130), CLI default 3 (bin/quant.py:173)
        mc2 = 3 if args.mc == 1 else 1
        sh.step(args.warmup, lr, mc2, trace=False)
        sh.profile_enable(True)
        sh.step(args.steps, lr, mc2, trace=False)
        msm, nm = sh.profile_read()
        sh.profile_enable(False)
        tm = msm / max(nm, 1) * 1e-3
        mc_other = {"MC_size": mc2, "avg_kernel_ms": tm * 1e3, "achieved": alg_bytes / tm / 1e9,
                    "frac": alg_bytes / tm / 1e9 / HBM_PEAK_GBS, "launches_timed": int(nm),
                    "what": "the same handle and kernel family at MC_size %d (%s default); same algorithmic bytes"
                            % (mc2, "brie-quant CLI, bin/quant.py:173" if mc2 == 3 else "BRIE2.fit API, model_TFProb.py:130")}

    f32_leg = None
    if rank == 0 and world == 1 and not args.no_f32_leg and storage_main != "f32":
        # the same kernel on the fp32 layers as uploaded (compact storage is reported separately)
        sh.set_count_storage(1)
        sh.step(args.warmup, lr, args.mc, trace=False)
        sh.profile_enable(True)
        sh.step(args.steps, lr, args.mc, trace=False)
        ms32, n32 = sh.profile_read()
        sh.profile_enable(False)
        f32_leg = {"avg_kernel_ms": ms32 / max(n32, 1), "storage_bytes_per_launch": sh.step_storage_bytes()}
    # ---- N > 1: the end-of-fit exchange of a gene-sharded fit (BASELINE configs[3]: "RCCL weight all-gather"),
    # untimed by `value` (there is no collective inside the optimisation loop) but executed, checked and reported
    gather_info = gather_state = None
    # (BRIE_BENCH_FORCE_GATHER=1: run this leg with a world of ONE rank too -- the only way to put the nccl branch and
    #  the C-ABI communicator through RCCL on a 1-GPU box)
    if dist is not None and args.scaling == "strong" and (world > 1 or os.environ.get("BRIE_BENCH_FORCE_GATHER")):
        gather_info, gather_state = end_of_fit_allgather(torch, dist, sh, cfg, args, seed, lr, rank, world, local_rank,
                                                         dev, Xc, size)
        gather_info["ms_per_step_per_rank"] = [r[0] for r in per_rank]

    out = None
    if rank == 0:
        # the dominant kernel against the roofline, PER GPU: each rank's algorithmic bytes / its own average launch time
        gpus = [{"rank": r, "genes": int(v[4]), "ms_per_step": v[0], "avg_kernel_ms": v[1],
                 "achieved": v[2] / (v[1] * 1e-3) / 1e9, "frac": v[2] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "hbm_rate_GBs": v[3] / (v[1] * 1e-3) / 1e9} for r, v in enumerate(per_rank)]
        worst = min(gpus, key=lambda g: g["frac"])
        roof = {"bound": "hbm", "achieved": worst["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": worst["frac"], "traffic": None,
                "definition": "achieved = ALGORITHMIC bytes (48 + 4 L per element, fp32 model of SURVEY 8d) of ONE launch "
                              "on ONE GPU / average launch time of the dominant kernel there (HIP events on the handle's "
                              "stream); with N > 1 the slowest GPU's figure, every GPU's under per_gpu; what the kernel "
                              "physically moves is hbm_rate_GBs",
                "kernel": "elbo_adam_step<Kc=%d>" % Kc, "avg_kernel_ms": worst["avg_kernel_ms"],
                "algorithmic_bytes_per_launch": int(per_rank[worst["rank"]][2]), "launches_timed": n_launch,
                "count_storage": storage_main, "storage_bytes_per_launch": int(per_rank[worst["rank"]][3]),
                # the physical HBM rate: bytes the current storage moves (integer counts are kept as u8 / u16,
                # bit-identical results) over the same time -- below `achieved` by construction
                "hbm_rate_GBs": worst["hbm_rate_GBs"], "hbm_frac_of_peak": worst["hbm_rate_GBs"] / HBM_PEAK_GBS,
                "rccl_ranks": world if (dist is not None and dist.get_backend() == "nccl") else 0}
        if world > 1:
            roof["per_gpu"] = gpus
        # where the allocator put the streamed arrays decides 10 - 20 % of the step time: the library probes
        # and keeps the fastest of up to three placements before the first step; the rates are storage bytes / probe time
        roof["placement"] = dict(placement, max_sets_asked=placement_sets,
                                 note_on_sets="brie_placement_configure by bench.py; the library's default is 4 for arrays >= 1 GiB")
        if mc_other is not None:
            roof["mc%d" % mc_other["MC_size"]] = mc_other
        if f32_leg is not None:
            t32 = f32_leg["avg_kernel_ms"] * 1e-3
            roof["f32_count_storage"] = dict(f32_leg, achieved=alg_bytes / t32 / 1e9, frac=alg_bytes / t32 / 1e9 / HBM_PEAK_GBS,
                                             note="same kernel, counts kept as the uploaded fp32 layers: storage bytes "
                                                  "= algorithmic bytes")
        out = {
            "metric": "ELBO iterations/sec (cells x genes)",
            "value": value,
            "unit": "cell*gene*iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preconditioning": precondition,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_without_profiling_events": ms_unprofiled,      # rank 0, untimed second pass of the same K steps
            "step_fusion": fusion,
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

    # ---- rank 0's host-side legs; the other ranks wait on sockets (wait_for_rank0 below)
    # (the device arrays the closed shard left for a successor are released first: the whole-fit figure below allocates
    #  like a fresh process, and the counter runs further down are other processes on this GPU)
    _capi.trim_memory()
    if e2e_inputs is not None:
        note("whole BRIE2.fit + BRIE_RV from host arrays (pcie_inclusive)")
        # Secondary, PCIe-inclusive figure (never `value`): the reference's unit of work through the public API --
        # host count layers in, BRIE2.fit with the default schedule (996 staged steps + 500-draw loss_gene), host
        # Psi / Z_std / Psi95CI / Z_loc out (model_wrap.py:138-146) -- upload, compaction and read-back included.
        import brie_amd
        t0 = time.perf_counter()
        mdl = brie_amd.BRIE2(Nc, ng, Kc=Kc, effLen=eff_all.cpu().numpy() if L == 3 else None, seed=seed, device=local_rank,
                             gene_offset=g0)
        mdl.fit(e2e_inputs, Xc=Xc_host, min_iter=1000, max_iter=1000, MC_size=args.mc, pseudo_count=0.01, verbose=False)
        rv = brie_amd.BRIE_RV(mdl)
        total = time.perf_counter() - t0
        assert np.isfinite(rv.Psi).all()
        # stages 2..6 of the fit (the first one also pays the count compaction and the placement search)
        in_fit = float(np.sum(mdl.timing["stage_s"][1:])) / (5 * 166) * 1e3
        out["roofline"]["in_fit_ms_per_step"] = in_fit
        out["roofline"]["in_fit_over_timed"] = in_fit / out["ms_per_step"] if in_fit else None
        out["roofline"]["in_fit_placement"] = mdl.timing.get("placement")
        out["pcie_inclusive"] = {"it_per_s_pcie_inclusive": 996 / total, "total_s": total, "steps": 996,
                                 "breakdown_s": {k: v for k, v in mdl.timing.items() if k.endswith("_s")},
                                 "what": "host numpy count layers -> BRIE2.fit (6 x 166 steps, 500-draw loss_gene) -> "
                                         "BRIE_RV with Psi, Z_std, Psi95CI, Z_loc on the host; result matrices stream "
                                         "out while loss_gene runs"}
        mdl.close()
        del e2e_inputs, rv

    _capi.trim_memory()
    if rank == 0:
        note("HBM traffic: two rocprofv3 --pmc child runs")
        out["roofline"].update(hbm_traffic(args, world, storage_main))
        note("traffic: %r (%s)" % (out["roofline"].get("traffic"), out["roofline"].get("traffic_source") or out["roofline"].get("traffic_error")))
    if rank == 0 and psi_quad is not None:
        # PSI delta ON THE TIMED WORKLOAD: genes are independent and the noise stream is keyed by the global gene
        # index, so the CPU oracle run on one gene quad over all Nc cells is an exact reference for those genes
        from oracle.brie_oracle import OracleBRIE2, add_pseudo_count
        # warm-up, (the plain leg,) the timed pass, the unprofiled pass, the traced step
        n_total = args.warmup + plain_steps + 2 * args.steps + 1
        o = OracleBRIE2(Nc, 4, Kc, effLen=quad_eff, seed=seed, gene_offset=g0 + q0, dtype=np.float64)
        o.minimize(add_pseudo_count(quad_layers), Xc_host, n_total, lr, args.mc)
        d = np.abs(psi_quad - o.Psi)
        out["psi_delta_headline_workload"] = {
            "what": "genes %d..%d of the timed %s run, all %d cells, %d Adam steps vs the fp64 CPU oracle"
                    % (g0 + q0, g0 + q0 + 3, args.config, Nc, n_total),
            "max": float(d.max()), "p99": float(np.percentile(d, 99))}
    if rank == 0:
        if not args.no_psi_check:
            note("PSI delta check on configs[0] (HIP, fp32 oracle, second fp32 evaluation)")
            try:
                out["psi_delta_vs_cpu_ref"] = psi_delta_check()
            except AssertionError as exc:          # a violated parity rule is REPORTED in the line, it does not cost the line
                out["psi_delta_vs_cpu_ref"] = {"rule": "tests/util.py::psi_null_rule VIOLATED", "violated": repr(exc)}
                if os.environ.get("BRIE_BENCH_STRICT"):
                    raise
        if not args.no_cpu_baseline:
            from oracle.brie_oracle_torch import time_reference_shape
            usable = usable_cores()
            cores = min(6, usable)                        # reference default --nproc 6 (bin/quant.py:183)
            note("CPU baselines (%d usable cores)" % usable)
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
                          "steps, eager torch-CPU autograd restatement (TF absent), %.1f s; rank 0's host cores%s"
                          % (nb, n_gene, Nc, n_steps, el, " while the other ranks idle" if world > 1 else ""),
                "gpu_over_cpu": value / eps_s,
            }
            # the same eager baseline on ALL usable host cores (SURVEY 8d asks for both; bin/quant.py:183 defaults to 6)
            if usable > cores:
                # calibrated on its own: eager per-op dispatch over many threads can be SLOWER than over six (call r4b: a
                # box with 256 cores in the mask never finished the six-thread step count)
                note("eager CPU baseline on all %d usable cores" % usable)
                eps_c, _ = time_reference_shape(Nc, n_gene * nb, counts_fn, Xc_host, 1, 3, args.mc, threads=usable,
                                                warmup_steps=1)
                n_all = int(max(3, min(n_steps, args.cpu_seconds * eps_c / (Nc * n_gene * nb))))
                eps_all, el_all = time_reference_shape(Nc, n_gene * nb, counts_fn, Xc_host, nb, n_all, args.mc, threads=usable)
                out["cpu_baseline_all_cores"] = {
                    "value": eps_all, "unit": "cell*gene*iterations/s", "cores": usable, "kind": "port",
                    "sample": "the cpu_baseline batches x %d Adam steps on all %d usable cores, %.1f s" % (n_all, usable, el_all),
                    "gpu_over_cpu": value / eps_all}
            else:
                out["cpu_baseline_all_cores"] = dict(out["cpu_baseline"], sample="this process may run on %d cores: all "
                                                     "usable cores = the cpu_baseline measurement above" % usable)
            # second, separately labelled baseline: the same algorithm as ONE fused
            # C/OpenMP pass (oracle/brie_oracle.c) on all host cores
            try:
                from oracle.c_oracle import COracle
                from oracle.brie_oracle import add_pseudo_count
                co = COracle(add_pseudo_count(sample_layers), Xc_host, effLen=None if L == 2 else eff_host, seed=seed)
                # all cores THIS PROCESS may run on (a box reports 256 CPUs and grants 6 of them); set explicitly because
                # torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks
                note("fused C/OpenMP CPU baseline")
                co.set_threads(usable)
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

    # ---- LAST: the C-ABI communicator's gather (never run between two GPUs by the build).  Nothing measured comes
    # after it; if it does not come back, rank 0 prints its line and every rank leaves without another collective.
    note("host legs done")
    wait_for_rank0()
    came_back = True
    if gather_state is not None:
        came_back = native_allgather_leg(gather_info, gather_state, local_rank, world)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if not came_back:
        # the stuck leg is reported in the line (allgather.native_error); every rank leaves with code 0 so that the
        # launcher does not tear rank 0 down before its line is out -- the other ranks give it a moment first
        if rank != 0:
            time.sleep(5.0)
        sys.stdout.flush()
        os._exit(3 if os.environ.get("BRIE_BENCH_STRICT") else 0)
    if dist is not None:
        dist.barrier(group=side)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

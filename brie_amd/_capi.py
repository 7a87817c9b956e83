"""ctypes binding of include/brie_amd.h (the C ABI of libbrie_amd.so).

The north star names cffi; cffi is not installed in this image, ctypes is, and
both bind the same C header in ABI mode.  There is NO CPU fallback: if the HIP
library is missing or no GPU is visible the calls raise.
"""
import ctypes
import os

import numpy as np

from .build import LIB_PATH

# brie_array ids (include/brie_amd.h)
COUNT1, COUNT2, COUNT3, XC, EFFLEN, XG = 0, 1, 2, 3, 4, 5
Z_LOC, Z_STD_LOG, WC_LOC, INTERCEPT, SIGMA_LOG, WG_LOC = 8, 9, 10, 11, 12, 13
PSI, Z_STD, PSI95CI, SIGMA = 16, 17, 18, 19
ABI_VERSION = 3
MAX_KC = 1024        # 0..8 in registers, 9..64 on the matrix cores inside the streaming pass, beyond in 64-feature panels
MAX_KG = 1024        # 0..4 in registers, 5..64 as a tile in LDS / on the matrix cores, beyond in 64-feature panels

PLACEMENT_MAX_SETS = 8          # BRIE_PLACEMENT_MAX_SETS
PLACEMENT_STATES = ("not_run", "good", "best_of_all", "stopped_memory", "stopped_time", "stopped_error", "off")   # brie_placement_state
EXPORTS = [
    "brie_create", "brie_destroy", "brie_upload", "brie_upload_typed", "brie_upload_sparse", "brie_add_pseudo_count", "brie_init_state",
    "brie_reset_optimizer", "brie_step", "brie_step_begin", "brie_rowstat_buffer", "brie_set_rowstat_buffer",
    "brie_step_end", "brie_set_gene_mask", "brie_read_loss_window", "brie_set_target", "brie_loss_gene", "brie_read", "brie_get_draw",
    "brie_set_draw", "brie_synchronize", "brie_profile_enable", "brie_profile_read",
    "brie_set_tiling", "brie_step_algorithmic_bytes", "brie_step_storage_bytes", "brie_set_count_storage",
    "brie_get_count_storage", "brie_calibrate_stream", "brie_simulate_psi", "brie_simulate_counts", "brie_device_memory", "brie_trim_memory",
    "brie_last_error", "brie_abi_version",
    "brie_comm_available", "brie_comm_unique_id", "brie_comm_init", "brie_comm_destroy", "brie_comm_rank", "brie_comm_world",
    "brie_comm_allgather", "brie_comm_allreduce", "brie_attach_comm",
    "brie_read_results_async", "brie_read_wait", "brie_host_register", "brie_host_unregister", "brie_reconfigure",
    "brie_loglik_mc", "brie_get_loss", "brie_debug_address", "brie_host_convert_u16", "brie_host_convert_slab",
    "brie_placement_probe", "brie_placement_tune", "brie_placement_info", "brie_placement_status", "brie_probe_layouts",
    "brie_placement_configure", "brie_debug_inject_placement_failure", "brie_probe_vmm", "brie_set_step_fusion",
    "brie_step_fusion_info", "brie_debug_step_fusion",
]
COMM_ID_BYTES = 128
#: numpy dtype -> brie_dtype of brie_upload_typed (count layers held as integers / float64 go up without a host cast)
TYPED_DTYPES = {"float64": 1, "int32": 2, "int64": 3, "uint8": 4, "uint16": 5, "int16": 6, "uint32": 7}


class BrieProblem(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("device", ctypes.c_int32),
        ("Nc", ctypes.c_int64), ("Ng", ctypes.c_int64), ("gene_offset", ctypes.c_int64),
        ("Kc", ctypes.c_int32), ("Kg", ctypes.c_int32), ("n_layers", ctypes.c_int32),
        ("has_efflen", ctypes.c_int32), ("intercept_mode", ctypes.c_int32),
        ("train_intercept", ctypes.c_int32), ("train_sigma", ctypes.c_int32),
        ("sharded", ctypes.c_int32), ("seed", ctypes.c_uint64),
    ]


class BrieError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libbrie_amd.so and declare every prototype of include/brie_amd.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("BRIE_AMD_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(
            "libbrie_amd.so not found at %s -- build it with `python -m brie_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % path)
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).
    # If torch is imported first, the dynamic loader binds libbrie_amd.so to that copy (SONAME match) and all
    # is well; loaded the other way round the process ends up with TWO runtimes and the one that initialises
    # second reports "no ROCm-capable device".  So: when torch is installed, load it before the library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.brie_create.argtypes = [ctypes.POINTER(BrieProblem), ctypes.POINTER(vp)]
    lib.brie_destroy.argtypes = [vp]
    lib.brie_upload.argtypes = [vp, ctypes.c_int, vp, i64, i64, i64]
    lib.brie_upload_typed.argtypes = [vp, ctypes.c_int, vp, i32, i64, i64, i64]
    lib.brie_upload_sparse.argtypes = [vp, ctypes.c_int, i32, vp, vp, vp, i64, i64, i64]
    lib.brie_add_pseudo_count.argtypes = [vp, f32]
    lib.brie_init_state.argtypes = [vp, f32, f32]
    lib.brie_reset_optimizer.argtypes = [vp]
    lib.brie_step.argtypes = [vp, i32, f32, i32, vp]
    lib.brie_step_begin.argtypes = [vp, f32, i32]
    lib.brie_rowstat_buffer.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.brie_set_rowstat_buffer.argtypes = [vp, vp]
    lib.brie_step_end.argtypes = [vp, ctypes.POINTER(f32)]
    lib.brie_set_gene_mask.argtypes = [vp, vp]
    lib.brie_read_loss_window.argtypes = [vp, i32, vp]
    lib.brie_set_target.argtypes = [vp, i32]
    lib.brie_loss_gene.argtypes = [vp, i32, vp]
    lib.brie_read.argtypes = [vp, ctypes.c_int, vp, i64, i64, i64]
    lib.brie_get_draw.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.brie_set_draw.argtypes = [vp, ctypes.c_uint32]
    lib.brie_synchronize.argtypes = [vp]
    lib.brie_profile_enable.argtypes = [vp, i32]
    lib.brie_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
    lib.brie_set_tiling.argtypes = [vp, i32]
    lib.brie_calibrate_stream.argtypes = [i32, i32, i32, i64, i32, i32, ctypes.POINTER(ctypes.c_double)]
    lib.brie_device_memory.argtypes = [i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.brie_simulate_psi.argtypes = [i32, i64, i64, i64, ctypes.c_uint64, vp, vp, vp]
    lib.brie_simulate_counts.argtypes = [i32, i64, i64, i64, ctypes.c_uint64, vp, vp, vp, vp, vp, vp]
    lib.brie_step_algorithmic_bytes.argtypes = [vp]
    lib.brie_step_algorithmic_bytes.restype = i64
    lib.brie_step_storage_bytes.argtypes = [vp]
    lib.brie_step_storage_bytes.restype = i64
    lib.brie_set_count_storage.argtypes = [vp, i32]
    lib.brie_get_count_storage.argtypes = [vp]
    lib.brie_comm_available.argtypes = [i32]
    lib.brie_comm_unique_id.argtypes = [vp]
    lib.brie_comm_init.argtypes = [i32, i32, i32, vp, ctypes.POINTER(vp)]
    lib.brie_comm_destroy.argtypes = [vp]
    lib.brie_comm_rank.argtypes = [vp]
    lib.brie_comm_world.argtypes = [vp]
    lib.brie_comm_allgather.argtypes = [vp, vp, i64, vp]
    lib.brie_comm_allreduce.argtypes = [vp, vp, i64, i32, i32]
    lib.brie_attach_comm.argtypes = [vp, vp]
    lib.brie_read_results_async.argtypes = [vp, vp, vp, vp, vp, i64]
    lib.brie_read_wait.argtypes = [vp]
    lib.brie_debug_address.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    lib.brie_loglik_mc.argtypes = [vp, i32, vp, i64]
    lib.brie_get_loss.argtypes = [vp, i32, i32, vp]
    lib.brie_host_convert_u16.argtypes = [vp, i64, i64, i64, vp, ctypes.POINTER(i32)]
    lib.brie_host_convert_slab.argtypes = [vp, i32, i64, i64, i64, vp, ctypes.POINTER(i32)]
    lib.brie_reconfigure.argtypes = [vp, i32, ctypes.c_uint64, i32, i32]
    lib.brie_host_register.argtypes = [vp, i64]
    lib.brie_host_unregister.argtypes = [vp]
    lib.brie_probe_layouts.argtypes = [i32, i64, i64, i64, i32, vp, i32, vp]
    lib.brie_probe_vmm.argtypes = [i32, i64, i64, i32, i32, vp, vp, i32, vp, vp]
    lib.brie_placement_probe.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double)]
    lib.brie_placement_tune.argtypes = [vp, i32, ctypes.c_double]
    lib.brie_placement_status.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, i32]
    lib.brie_placement_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_double), i32,
                                        ctypes.POINTER(ctypes.c_double)]
    lib.brie_placement_configure.argtypes = [vp, i32, ctypes.c_double, ctypes.c_double]
    lib.brie_set_step_fusion.argtypes = [vp, i32]
    lib.brie_debug_step_fusion.argtypes = [vp, i32]
    lib.brie_step_fusion_info.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.brie_debug_inject_placement_failure.argtypes = [vp, i32]
    lib.brie_last_error.restype = ctypes.c_char_p
    lib.brie_abi_version.restype = ctypes.c_int
    for name in EXPORTS:
        if name not in ("brie_step_algorithmic_bytes", "brie_step_storage_bytes", "brie_last_error",
                        "brie_abi_version", "brie_comm_rank", "brie_comm_world"):
            getattr(lib, name).restype = ctypes.c_int
    if lib.brie_abi_version() != ABI_VERSION:
        raise ImportError("libbrie_amd.so ABI %d != binding %d" % (lib.brie_abi_version(), ABI_VERSION))
    if path == os.environ.get("BRIE_AMD_LIB", LIB_PATH):
        _lib = lib
    return lib


class Comm(object):
    """`brie_comm`: an RCCL communicator created through the C ABI (one per process / GPU).

    `unique_id()` on rank 0 -> hand the 128 bytes to every rank (any transport) -> `Comm(device, rank, world, id)`."""

    @staticmethod
    def available(device=0):
        """None when librccl binds and `device` exists (brie_comm_available), else the reason as a string."""
        lib = load_library()
        if lib.brie_comm_available(int(device)) == 0:
            return None
        return (lib.brie_last_error() or b"brie_comm_available failed").decode()

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
        _check(lib, lib.brie_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def __init__(self, device, rank, world, unique_id):
        self.lib = load_library()
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % COMM_ID_BYTES)
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._c = ctypes.c_void_p()
        _check(self.lib, self.lib.brie_comm_init(int(device), int(rank), int(world), ctypes.cast(buf, ctypes.c_void_p),
                                                 ctypes.byref(self._c)))
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    def allgather(self, send):
        """(n,) float32 per rank -> (world, n) on every rank."""
        send = np.ascontiguousarray(send, np.float32).ravel()
        out = np.empty((self.world, send.size), np.float32)
        _check(self.lib, self.lib.brie_comm_allgather(self._c, send.ctypes.data, send.size, out.ctypes.data))
        return out

    def allreduce(self, a, op="sum"):
        """In-place-style reduction of a host float32 / float64 array over ranks; returns the reduced copy."""
        a = np.array(a, dtype=np.float64 if np.asarray(a).dtype == np.float64 else np.float32, copy=True)
        a = np.ascontiguousarray(a)
        _check(self.lib, self.lib.brie_comm_allreduce(self._c, a.ctypes.data, a.size, 1 if a.dtype == np.float64 else 0,
                                                      {"sum": 0, "max": 1, "min": 2}[op]))
        return a

    def allreduce_device(self, ptr, count, dtype="f32", op="sum"):
        """In place on a device buffer (raw pointer)."""
        _check(self.lib, self.lib.brie_comm_allreduce(self._c, ctypes.c_void_p(int(ptr)), int(count),
                                                      {"f32": 0, "f64": 1}[dtype], {"sum": 0, "max": 1, "min": 2}[op]))

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self.lib.brie_comm_destroy(self._c)
            self._c = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check(lib, rc):
    if rc != 0:
        msg = lib.brie_last_error().decode("utf-8", "replace")
        if rc in (-1,):
            raise ValueError("brie_amd: " + msg)
        if rc == -4:
            raise NotImplementedError("brie_amd: " + msg)
        if rc == -5:
            raise BrieError("brie_amd (RCCL): " + msg)
        raise BrieError("brie_amd (status %d): %s" % (rc, msg))


def _matrix_pointer(x):
    """(pointer, rows, cols, ld, keepalive) of a float32 row-major matrix.

    Accepts numpy arrays (host) and anything exposing `data_ptr()`/`stride()`
    (e.g. a torch tensor already resident in HBM) -- the library copies with
    hipMemcpyDefault, so both kinds of pointer are fine.
    """
    if hasattr(x, "data_ptr") and hasattr(x, "stride"):
        import torch
        if x.dtype != torch.float32:
            x = x.float()
        if x.dim() == 1:
            x = x.reshape(1, -1)
        if x.stride(1) != 1:
            x = x.contiguous()
        return ctypes.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], max(x.stride(0), x.shape[1]), x
    a = np.asarray(x, dtype=np.float32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    return a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1], a.shape[1], a


def calibrate_stream(n_read, n_write, bytes_per_stream=1 << 30, iters=5, device=0, lds_bytes=0, nt=False):
    """HBM GB/s of a pure streaming kernel with the given read/write stream mix."""
    if nt:
        lds_bytes = -(int(lds_bytes) + 1)
    lib = load_library()
    out = ctypes.c_double()
    _check(lib, lib.brie_calibrate_stream(int(device), int(n_read), int(n_write), int(bytes_per_stream),
                                          int(iters), int(lds_bytes), ctypes.byref(out)))
    return out.value


def host_convert_u16(a):
    """The host half of the staged count ingest (brie_host_convert_u16): (uint16 copy, not_integral flag)."""
    lib = load_library()
    a = np.asarray(a, np.float32)
    if a.ndim != 2 or a.strides[1] != 4 or a.strides[0] % 4:
        a = np.ascontiguousarray(a)
    out = np.empty(a.shape, np.uint16)
    flag = ctypes.c_int32()
    if a.size == 0:
        return out, False
    _check(lib, lib.brie_host_convert_u16(a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1], a.strides[0] // 4,
                                          out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(flag)))
    return out, bool(flag.value)


def probe_layouts(Nc, Ng, slab_bytes, offsets, iters=3, device=0):
    """GB/s of the placement probe for every row of `offsets` (n_layouts, 8) inside ONE slab (brie_probe_layouts)."""
    lib = load_library()
    off = np.ascontiguousarray(offsets, np.int64).reshape(-1, 8)
    out = np.zeros(off.shape[0], np.float64)
    _check(lib, lib.brie_probe_layouts(int(device), int(Nc), int(Ng), int(slab_bytes), off.shape[0], off.ctypes.data, int(iters),
                                       out.ctypes.data))
    return out


def probe_vmm(Nc, Ng, n_layers, chunk_bytes, order, iters=3, device=0):
    """(GB/s, seconds to build) of the placement probe on sets of streamed arrays built from the HIP virtual-memory API
    (brie_probe_vmm): per layout a physical chunk size and the order the chunks are created in (0 array after array,
    1 round robin over the arrays, 2 plain hipMalloc)."""
    lib = load_library()
    cb = np.ascontiguousarray(chunk_bytes, np.int64)
    od = np.ascontiguousarray(order, np.int32)
    assert cb.shape == od.shape and cb.ndim == 1
    gbs, secs = np.zeros(cb.shape[0], np.float64), np.zeros(cb.shape[0], np.float64)
    _check(lib, lib.brie_probe_vmm(int(device), int(Nc), int(Ng), int(n_layers), cb.shape[0], cb.ctypes.data, od.ctypes.data,
                                   int(iters), gbs.ctypes.data, secs.ctypes.data))
    return gbs, secs


def host_convert_slab(a):
    """One slab of the typed ingest (brie_host_convert_slab): the array as u16 when it holds nothing but integers in
    [0, 65535], else as the float32 cast; returns (converted, went_as_float32)."""
    lib = load_library()
    a = np.ascontiguousarray(a)
    dt = 0 if a.dtype == np.float32 else TYPED_DTYPES[a.dtype.name]
    buf = np.empty(a.shape, np.float32)
    flag = ctypes.c_int32()
    if a.size:
        _check(lib, lib.brie_host_convert_slab(a.ctypes.data_as(ctypes.c_void_p), dt, a.shape[0], a.shape[1], a.shape[1],
                                               buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(flag)))
    if flag.value:
        return buf, True
    return buf.view(np.uint16).ravel()[:a.size].reshape(a.shape).copy(), False


def host_register(a):
    """Page-lock a numpy array in place (hipHostRegister); ctypes drops the GIL for the duration of the call."""
    lib = load_library()
    _check(lib, lib.brie_host_register(a.ctypes.data_as(ctypes.c_void_p), a.nbytes))


def host_unregister(a):
    lib = load_library()
    _check(lib, lib.brie_host_unregister(a.ctypes.data_as(ctypes.c_void_p)))


def device_memory(device=0):
    """(free, total) HBM bytes of `device`."""
    lib = load_library()
    free, total = ctypes.c_int64(), ctypes.c_int64()
    _check(lib, lib.brie_device_memory(int(device), ctypes.byref(free), ctypes.byref(total)))
    return free.value, total.value


def trim_memory():
    """Release the device arrays the last closed shard left for its successor (brie_trim_memory)."""
    lib = load_library()
    _check(lib, lib.brie_trim_memory())


def shard_bytes(Nc, Ng, n_layers, Kc=0):
    """Upper estimate of the HBM a gene shard needs while it is being set up: state + 4 Adam moments (24 B per
    element), fp32 count layers before they are compacted, read-back staging, the residual buffer of wide designs,
    per-chunk partial sums."""
    ld = -(-int(Ng) // 256) * 256
    per_elem = 24 + 4 * n_layers + 4 + (4 if Kc > 8 else 0)
    return int(Nc) * ld * per_elem + (256 << 20)


def _f32_matrix(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("brie_amd: expected an array of shape %s, got %s" % (tuple(shape), a.shape))
    return a


def simulate_psi(mean_logit, sigma, seed=0, gene_offset=0, device=0):
    """brie_simulate_psi: Psi = sigmoid(clip(mean_logit + sigma_j * N(0,1), -9, 9)) for host (Nc, Ng) arrays."""
    lib = load_library()
    mean_logit = _f32_matrix(mean_logit)
    Nc, Ng = mean_logit.shape
    sigma = _f32_matrix(np.broadcast_to(np.asarray(sigma, np.float32).reshape(-1), (Ng,)))
    out = np.empty((Nc, Ng), np.float32)
    _check(lib, lib.brie_simulate_psi(int(device), Nc, Ng, int(gene_offset), int(seed) & (2 ** 64 - 1),
                                      mean_logit.ctypes.data, sigma.ctypes.data, out.ctypes.data))
    return out


def simulate_counts(psi, total, effLen=None, seed=0, gene_offset=0, device=0):
    """brie_simulate_counts: Multinomial(total, phi) reads per (cell, gene); returns (c1, c2, c3 or None)."""
    lib = load_library()
    psi = _f32_matrix(psi)
    Nc, Ng = psi.shape
    total = _f32_matrix(total, (Nc, Ng))
    eff = None if effLen is None else _f32_matrix(effLen, (Ng, 6))
    out = [np.empty((Nc, Ng), np.float32) for _ in range(3 if eff is not None else 2)]
    _check(lib, lib.brie_simulate_counts(int(device), Nc, Ng, int(gene_offset), int(seed) & (2 ** 64 - 1),
                                         psi.ctypes.data, total.ctypes.data,
                                         None if eff is None else eff.ctypes.data,
                                         out[0].ctypes.data, out[1].ctypes.data,
                                         None if eff is None else out[2].ctypes.data))
    return out[0], out[1], (out[2] if eff is not None else None)


class Shard(object):
    """Thin object wrapper over one `brie_handle` (one gene shard on one GPU)."""
    first_touch_results = True   # BRIE2.fit first-touches its (pageable) result arrays on a background thread

    def __init__(self, Nc, Ng, Kc=0, n_layers=2, has_efflen=False, train_intercept=True,
                 train_sigma=True, seed=0, device=0, gene_offset=0, Kg=0, intercept_mode=0, sharded=False):
        self.lib = load_library()
        self.Nc, self.Ng, self.Kc, self.Kg = int(Nc), int(Ng), int(Kc), int(Kg)
        self.cell_mode = int(intercept_mode) == 1
        p = BrieProblem(ABI_VERSION, int(device), int(Nc), int(Ng), int(gene_offset), int(Kc), int(Kg),
                        int(n_layers), int(bool(has_efflen)), int(intercept_mode),
                        int(bool(train_intercept)), int(bool(train_sigma)), int(bool(sharded)),
                        int(seed) & (2 ** 64 - 1))
        self._h = ctypes.c_void_p()
        _check(self.lib, self.lib.brie_create(ctypes.byref(p), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.brie_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, which, x):
        if hasattr(x, "tocsc") and hasattr(x, "indptr") is False:
            x = x.tocsc()
        if hasattr(x, "indptr") and hasattr(x, "indices"):           # scipy CSC / CSR: densified on the device
            fmt = 1 if x.format == "csr" else 0
            if x.format not in ("csc", "csr"):
                x = x.tocsc()
            indptr = np.ascontiguousarray(x.indptr, np.int64)
            indices = np.ascontiguousarray(x.indices, np.int32)
            data = np.ascontiguousarray(x.data, np.float32)
            _check(self.lib, self.lib.brie_upload_sparse(
                self._h, which, fmt, indptr.ctypes.data_as(ctypes.c_void_p), indices.ctypes.data_as(ctypes.c_void_p),
                data.ctypes.data_as(ctypes.c_void_p), int(x.nnz), x.shape[0], x.shape[1]))
            return
        if hasattr(x, "tocsc"):                                       # other scipy sparse formats
            return self.upload(which, x.tocsc())
        if (which in (COUNT1, COUNT2, COUNT3) and isinstance(x, np.ndarray) and x.ndim == 2 and x.dtype.name in TYPED_DTYPES
                and x.dtype.isnative and x.flags.aligned      # (dtype.name ignores byte order: '>i4' is 'int32' too)
                and x.strides[1] == x.itemsize and x.strides[0] % x.itemsize == 0 and x.strides[0] >= x.shape[1] * x.itemsize):
            # integer / float64 layers: converted by the library's ingest threads instead of a numpy astype pass
            _check(self.lib, self.lib.brie_upload_typed(self._h, which, x.ctypes.data_as(ctypes.c_void_p),
                                                        TYPED_DTYPES[x.dtype.name], x.shape[0], x.shape[1],
                                                        x.strides[0] // x.itemsize))
            return
        ptr, rows, cols, ld, keep = _matrix_pointer(x)
        _check(self.lib, self.lib.brie_upload(self._h, which, ptr, rows, cols, ld))
        del keep

    def reconfigure(self, Kc, seed, train_intercept=True, train_sigma=True):
        """Next model on the same count layers (brie_reconfigure): new design width and seed, counts stay in HBM."""
        _check(self.lib, self.lib.brie_reconfigure(self._h, int(Kc), int(seed) & (2 ** 64 - 1),
                                                   int(bool(train_intercept)), int(bool(train_sigma))))
        self.Kc = int(Kc)

    def add_pseudo_count(self, pc):
        _check(self.lib, self.lib.brie_add_pseudo_count(self._h, float(pc)))

    def init_state(self, intercept=None, sigma=None):
        nan = float("nan")
        _check(self.lib, self.lib.brie_init_state(
            self._h, nan if intercept is None else float(intercept), nan if sigma is None else float(sigma)))

    def reset_optimizer(self):
        _check(self.lib, self.lib.brie_reset_optimizer(self._h))

    def step(self, n_steps, lr, mc_size=1, trace=True):
        if trace:
            out = np.empty(int(n_steps), np.float32)
            _check(self.lib, self.lib.brie_step(self._h, int(n_steps), float(lr), int(mc_size),
                                                out.ctypes.data_as(ctypes.c_void_p)))
            return out
        _check(self.lib, self.lib.brie_step(self._h, int(n_steps), float(lr), int(mc_size), None))
        return None

    def rowstat_size(self):
        """Number of floats of the per-cell statistics buffer of a coupled fit (brie_rowstat_buffer)."""
        dev, n = ctypes.c_void_p(), ctypes.c_int64()
        _check(self.lib, self.lib.brie_rowstat_buffer(self._h, ctypes.byref(dev), ctypes.byref(n)))
        return int(n.value)

    def step_sharded(self, n_steps, lr, mc_size, allreduce_inplace, stat_tensor):
        """Coupled gene-sharded steps: begin -> all-reduce of the per-cell statistics -> end.

        `stat_tensor`: a float32 device tensor of rowstat_size() elements registered as the statistics
        buffer; `allreduce_inplace(t)` sums it over ranks (RCCL).  Returns the LOCAL loss trace."""
        _check(self.lib, self.lib.brie_set_rowstat_buffer(self._h, ctypes.c_void_p(stat_tensor.data_ptr())))
        out = np.empty(int(n_steps), np.float32)
        loss = ctypes.c_float()
        for i in range(int(n_steps)):
            _check(self.lib, self.lib.brie_step_begin(self._h, float(lr), int(mc_size)))
            self.synchronize()                      # statistics are complete on the handle's stream
            allreduce_inplace(stat_tensor)          # blocks until the reduced values are visible
            _check(self.lib, self.lib.brie_step_end(self._h, ctypes.byref(loss)))
            out[i] = loss.value
        return out

    def attach_comm(self, comm):
        """Sharded coupled fit: all-reduce the per-cell statistics inside brie_step (None detaches)."""
        _check(self.lib, self.lib.brie_attach_comm(self._h, comm._c if comm is not None else None))
        self._comm_keepalive = comm

    def set_gene_mask(self, active=None):
        """Per-gene train mask (bool (Ng,)); None = all genes active."""
        if active is None:
            _check(self.lib, self.lib.brie_set_gene_mask(self._h, None))
            return
        a = np.ascontiguousarray(np.asarray(active).astype(np.uint8))
        if a.shape != (self.Ng,):
            raise ValueError("mask must have shape (%d,)" % self.Ng)
        _check(self.lib, self.lib.brie_set_gene_mask(self._h, a.ctypes.data_as(ctypes.c_void_p)))

    def read_loss_window(self, n_last):
        """(n_last, Ng) per-gene losses of the last steps, oldest first."""
        out = np.empty((int(n_last), self.Ng), np.float32)
        _check(self.lib, self.lib.brie_read_loss_window(self._h, int(n_last), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_target(self, target):
        """'ELBO' (default) or 'marginLik' (model_TFProb.py:194-211)."""
        _check(self.lib, self.lib.brie_set_target(self._h, {"ELBO": 0, "marginLik": 1}[target]))

    def loss_gene(self, n_repeats=500):
        out = np.empty(self.Ng, np.float32)
        _check(self.lib, self.lib.brie_loss_gene(self._h, int(n_repeats), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def read(self, which):
        par = (self.Nc, 1) if self.cell_mode else (1, self.Ng)
        shape = {XC: (self.Nc, self.Kc), WC_LOC: (self.Kc, self.Ng), WG_LOC: (self.Nc, self.Kg), INTERCEPT: par,
                 SIGMA_LOG: par, SIGMA: par}.get(which, (self.Nc, self.Ng))
        out = np.empty(shape, np.float32)
        if out.size:
            _check(self.lib, self.lib.brie_read(self._h, which, out.ctypes.data_as(ctypes.c_void_p),
                                                shape[0], shape[1], shape[1]))
        return out

    def debug_address(self, which):
        a = ctypes.c_uint64()
        _check(self.lib, self.lib.brie_debug_address(self._h, int(which), ctypes.byref(a)))
        return a.value

    def placement_probe(self, iters=3):
        """GB/s (storage bytes of a step) of the effect-free probe kernel on the handle's arrays as they are placed."""
        g = ctypes.c_double()
        _check(self.lib, self.lib.brie_placement_probe(self._h, int(iters), ctypes.byref(g)))
        return g.value

    def placement_tune(self, max_tries=3, good_gbs=1e30):
        """Probe, and while the rate is below `good_gbs` try up to `max_tries` placements in all; keep the fastest."""
        _check(self.lib, self.lib.brie_placement_tune(self._h, int(max_tries), float(good_gbs)))
        return self.placement_info()

    def set_step_fusion(self, mode):
        """Many Adam steps per launch for small inputs (brie_set_step_fusion): -1 automatic, 0 never, 1 whenever allowed."""
        _check(self.lib, self.lib.brie_set_step_fusion(self._h, int(mode)))

    def debug_step_fusion(self, flags):
        """Tests / experiments (brie_debug_step_fusion): phase switches and the poll bound of the fused launches."""
        _check(self.lib, self.lib.brie_debug_step_fusion(self._h, int(flags)))

    def step_fusion_info(self):
        n, k = ctypes.c_int64(), ctypes.c_int64()
        _check(self.lib, self.lib.brie_step_fusion_info(self._h, ctypes.byref(n), ctypes.byref(k)))
        return {"launches": n.value, "steps": k.value}

    def placement_configure(self, max_sets=0, hbm_fraction=0.0, max_seconds=0.0):
        """Per-handle limits of the placement search (brie_placement_configure; 0 keeps a default): sets in all, the share of
        the free HBM a round may take, seconds after which nothing more is allocated."""
        _check(self.lib, self.lib.brie_placement_configure(self._h, int(max_sets), float(hbm_fraction), float(max_seconds)))

    def inject_placement_failure(self, point):
        """Tests: the next search of THIS handle fails at `point` (brie_debug_inject_placement_failure)."""
        _check(self.lib, self.lib.brie_debug_inject_placement_failure(self._h, int(point)))

    def placement_info(self):
        """Sets probed, the one in use (0 = the original), their rates, the seconds the search took, how it ended
        (PLACEMENT_STATES), its peak transient holding of candidate sets and, when it ended short of a fast set, why."""
        t, k, s = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        g = (ctypes.c_double * PLACEMENT_MAX_SETS)()
        _check(self.lib, self.lib.brie_placement_info(self._h, ctypes.byref(t), ctypes.byref(k), g, PLACEMENT_MAX_SETS, ctypes.byref(s)))
        st, peak, note = ctypes.c_int32(), ctypes.c_int64(), ctypes.create_string_buffer(192)
        _check(self.lib, self.lib.brie_placement_status(self._h, ctypes.byref(st), ctypes.byref(peak), note, 192))
        out = {"tries": t.value, "kept": k.value, "GBs": [round(g[i], 1) for i in range(t.value)],
               "seconds": round(s.value, 4), "status": PLACEMENT_STATES[st.value], "peak_extra_bytes": peak.value}
        if note.value:
            out["note"] = note.value.decode()
        return out

    def loglik_mc(self, size=10):
        """(Nc, Ng) Monte-Carlo log-likelihood of every entry under the current target (brie_loglik_mc)."""
        out = np.empty((self.Nc, self.Ng), np.float32)
        _check(self.lib, self.lib.brie_loglik_mc(self._h, int(size), out.ctypes.data_as(ctypes.c_void_p), self.Ng))
        return out

    def get_loss(self, mc_size=1, axis=0):
        """One stochastic loss evaluation under the current target, per gene (axis 0, (Ng,)) or per cell (axis 1, (Nc,));
        brie_get_loss."""
        out = np.empty(self.Ng if int(axis) == 0 else self.Nc, np.float32)
        _check(self.lib, self.lib.brie_get_loss(self._h, int(mc_size), int(axis), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def read_results_async(self, psi=None, z_std=None, psi95ci=None, z_loc=None):
        """Start the one-pass export of the result matrices into the given C-contiguous (Nc, Ng) float32 arrays
        (None = skip); returns immediately, `read_wait()` completes them."""
        ptrs = []
        for a in (psi, z_std, psi95ci, z_loc):
            if a is None:
                ptrs.append(None)
                continue
            if a.dtype != np.float32 or a.shape != (self.Nc, self.Ng) or not a.flags.c_contiguous:
                raise ValueError("destination must be a C-contiguous float32 (%d, %d) array" % (self.Nc, self.Ng))
            ptrs.append(a.ctypes.data_as(ctypes.c_void_p))
        self._async_keep = (psi, z_std, psi95ci, z_loc)
        _check(self.lib, self.lib.brie_read_results_async(self._h, ptrs[0], ptrs[1], ptrs[2], ptrs[3], self.Ng))

    def read_wait(self):
        _check(self.lib, self.lib.brie_read_wait(self._h))
        self._async_keep = None

    @property
    def draw(self):
        d = ctypes.c_uint32()
        _check(self.lib, self.lib.brie_get_draw(self._h, ctypes.byref(d)))
        return d.value

    @draw.setter
    def draw(self, value):
        _check(self.lib, self.lib.brie_set_draw(self._h, int(value)))

    def synchronize(self):
        _check(self.lib, self.lib.brie_synchronize(self._h))

    def profile_enable(self, enable=True):
        _check(self.lib, self.lib.brie_profile_enable(self._h, int(bool(enable))))

    def profile_read(self):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _check(self.lib, self.lib.brie_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def set_tiling(self, rows_per_chunk=0):
        _check(self.lib, self.lib.brie_set_tiling(self._h, int(rows_per_chunk)))

    def step_algorithmic_bytes(self):
        return int(self.lib.brie_step_algorithmic_bytes(self._h))

    def step_storage_bytes(self):
        return int(self.lib.brie_step_storage_bytes(self._h))

    def set_count_storage(self, mode):
        """0 = auto (u8 when every count is an integer <= 255), 1 = always fp32."""
        _check(self.lib, self.lib.brie_set_count_storage(self._h, int(mode)))

    @property
    def count_storage(self):
        return {0: "f32", 1: "u8", 2: "u16", 3: "u8/u16 per gene quad"}.get(int(self.lib.brie_get_count_storage(self._h)), "?")

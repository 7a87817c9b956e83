"""Counter-based normal noise shared by the CPU oracle and the HIP kernels.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws its Monte-Carlo noise with TensorFlow's unseeded Philox
stream (`tfd.Normal(...).sample(MC_size)`, /root/reference/brie/models/
model_TFProb.py:157-159) and its initial state with `tf.random.normal`
(model_TFProb.py:18,27-31).  TF is not available, and its stream could not be
reproduced on a GPU kernel anyway, so parity is defined under a *shared,
specified* noise stream instead:

    eps(seed, draw, k, cell, gene)  ~  N(0, 1)

    (x0,x1,x2,x3) = Philox4x32-10(counter=(gene//4, cell, draw, k),
                                  key=(seed & 0xffffffff, seed >> 32))
    U(x)   = ((x >> 9) + 0.5) * 2**-23                     in (0, 1), exact fp32
    pair 0 = sqrt(-2 ln U(x0)) * (cos, sin)(2 pi U(x1))   -> genes 4q+0, 4q+1
    pair 1 = sqrt(-2 ln U(x2)) * (cos, sin)(2 pi U(x3))   -> genes 4q+2, 4q+3

`gene` is the GLOBAL gene index (shard offset included) so the stream does not
depend on how genes are sharded over GPUs.  The oracle evaluates the
Box-Muller expression in float64 and rounds once to float32; the device
evaluates it in float32 and must agree to a few ulp.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

#: draw id reserved for the initial state (Model_init analogue)
INIT_DRAW = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds (Salmon et al., SC'11), vectorised.

    All inputs broadcastable unsigned 32-bit values. Returns 4 uint32 arrays.
    """
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0          # 64-bit products
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def _u01(x):
    """uint32 -> float64 uniform in (0,1) on a 2^-23 grid (exact in fp32)."""
    return ((x >> np.uint32(9)).astype(np.float64) + 0.5) * (2.0 ** -23)


def normal_quads(seed, draw, k, cells, quads, float_box_muller=False):
    """eps for all `cells` x gene quads `quads` -> float32 (len(cells), len(quads), 4).
    float_box_muller: log / sqrt / cos / sin evaluated in float32 on the (exact) float32 uniforms instead of in float64
    and rounded once -- what another fp32 implementation of the stream does (within 2e-6 of the defined values)."""
    cells = np.asarray(cells, dtype=np.uint64).reshape(-1, 1)
    quads = np.asarray(quads, dtype=np.uint64).reshape(1, -1)
    seed = int(seed)
    x0, x1, x2, x3 = philox4x32_10(quads, cells, int(draw), int(k),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty(x0.shape + (4,), dtype=np.float32)
    for p, (xa, xb) in enumerate(((x0, x1), (x2, x3))):
        if float_box_muller:
            ua, ub = _u01(xa).astype(np.float32), _u01(xb).astype(np.float32)
            r = np.sqrt(np.float32(-2.0) * np.log(ua))
            th = np.float32(6.283185307179586) * ub
            out[..., 2 * p] = r * np.cos(th)
            out[..., 2 * p + 1] = r * np.sin(th)
            continue
        r = np.sqrt(-2.0 * np.log(_u01(xa)))
        th = 2.0 * np.pi * _u01(xb)
        out[..., 2 * p] = (r * np.cos(th)).astype(np.float32)
        out[..., 2 * p + 1] = (r * np.sin(th)).astype(np.float32)
    return out


def normal(seed, draw, k, n_cells, n_genes, gene_offset=0, cell_offset=0, float_box_muller=False):
    """eps[(cell_offset..+n_cells), (gene_offset..+n_genes)] as float32 (n_cells, n_genes)."""
    g0 = int(gene_offset)
    g1 = g0 + int(n_genes)
    q0, q1 = g0 // 4, (g1 + 3) // 4
    cells = np.arange(cell_offset, cell_offset + n_cells)
    e = normal_quads(seed, draw, k, cells, np.arange(q0, q1), float_box_muller=float_box_muller)
    e = e.reshape(len(cells), (q1 - q0) * 4)
    return np.ascontiguousarray(e[:, g0 - 4 * q0: g0 - 4 * q0 + n_genes])

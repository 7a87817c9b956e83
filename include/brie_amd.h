/*
 * brie_amd.h -- C ABI of the MI355X-native brie-quant inference core.
 *
 * The reference (huangyh09/brie v2.3.0) has NO native/FFI boundary on this
 * path: the per-gene variational ELBO loop lives behind plain Python
 * callables (brie/models/__init__.py:1-2) whose arithmetic is executed by
 * TensorFlow / TensorFlow-Probability.  This header is the boundary a
 * maintainer would bind instead (ctypes / cffi ABI mode).
 * Each entry point cites the reference code it replaces
 * (paths relative to /root/reference/).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - every function returns BRIE_OK (0) or a negative brie_status; the message
 *    of the last failure on the calling thread is brie_last_error().
 *  - all matrices are float32, row-major, `ld` = elements between rows.
 *    Cell x gene matrices are (Nc, Ng) C-order exactly as the reference's
 *    count layers (brie/models/model_wrap.py:108-111); no host repack needed.
 *  - `src`/`dst` may be host OR device pointers (hipMemcpyDefault); the
 *    library copies in at upload and out at read and never retains them.
 *    brie_upload of a DEVICE `src` first waits for all prior work on the device
 *    (it may still be being written on another stream; host sources need no
 *    such wait); brie_read returns after the copy has completed.
 *  - a handle owns one gene shard on one device; a handle is not thread-safe,
 *    distinct handles are independent.
 */
#ifndef BRIE_AMD_H
#define BRIE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRIE_AMD_ABI_VERSION 3      /* 3: brie_placement_info takes the capacity of the caller's rate buffer */

typedef enum brie_status {
    BRIE_OK = 0,
    BRIE_ERR_INVALID = -1,      /* bad argument / shape                     */
    BRIE_ERR_STATE = -2,        /* call order: something not uploaded yet   */
    BRIE_ERR_HIP = -3,          /* HIP runtime failure (no GPU, OOM, ...)   */
    BRIE_ERR_UNSUPPORTED = -4,  /* mode outside the hot path built so far   */
    BRIE_ERR_COMM = -5          /* RCCL failure / librccl not loadable      */
} brie_status;

/* Arrays addressable through brie_upload / brie_read. */
typedef enum brie_array {
    /* inputs (upload) */
    BRIE_COUNT1 = 0,      /* (Nc, Ng) isoform1 / spliced   -- count_layers[0], model_TFProb.py:165 */
    BRIE_COUNT2 = 1,      /* (Nc, Ng) isoform2 / unspliced -- count_layers[1], model_TFProb.py:166 */
    BRIE_COUNT3 = 2,      /* (Nc, Ng) ambiguous            -- count_layers[2], model_TFProb.py:184-185 */
    BRIE_XC = 3,          /* (Nc, Kc) cell features        -- model_TFProb.py:220 */
    BRIE_EFFLEN = 4,      /* (Ng, 6) effective lengths     -- model_TFProb.py:49,175-176 */
    BRIE_XG = 5,          /* (Ng, Kg) gene features        -- model_TFProb.py:221,124-125 */
    /* state (upload = warm start / init_obj hook model_TFProb.py:45,62-65; read) */
    BRIE_Z_LOC = 8,       /* (Nc, Ng)  model_TFProb.py:80 */
    BRIE_Z_STD_LOG = 9,   /* (Nc, Ng)  model_TFProb.py:82 */
    BRIE_WC_LOC = 10,     /* (Kc, Ng)  model_TFProb.py:84 */
    BRIE_INTERCEPT = 11,  /* (1, Ng), or (Nc, 1) in cell mode   model_TFProb.py:53-60,67-71 */
    BRIE_SIGMA_LOG = 12,  /* (1, Ng), or (Nc, 1) in cell mode   model_TFProb.py:73-78 */
    BRIE_WG_LOC = 13,     /* (Nc, Kg)  model_TFProb.py:85 */
    /* derived (read only) */
    BRIE_PSI = 16,        /* (Nc, Ng) sigmoid(Z_loc)              model_TFProb.py:92-95 */
    BRIE_Z_STD = 17,      /* (Nc, Ng) exp(Z_std_log)              model_TFProb.py:88-90 */
    BRIE_PSI95CI = 18,    /* (Nc, Ng) q(.975)-q(.025) logit-normal model_TFProb.py:102-106 */
    BRIE_SIGMA = 19       /* (1, Ng) / (Nc, 1)  exp(sigma_log)    model_TFProb.py:108-111 */
} brie_array;

/* Problem description == the constructor arguments of BRIE2
 * (brie/models/model_TFProb.py:42-45) for one gene shard. */
typedef struct brie_problem {
    int32_t abi_version;      /* BRIE_AMD_ABI_VERSION */
    int32_t device;           /* HIP device ordinal */
    int64_t Nc;               /* cells */
    int64_t Ng;               /* genes in this shard */
    int64_t gene_offset;      /* global index of the shard's first gene (multiple of 4);
                                 keys the noise stream so results do not depend on sharding */
    int32_t Kc;               /* cell features (0..BRIE_MAX_KC_PANELS) */
    int32_t Kg;               /* gene features (0..BRIE_MAX_KG_PANELS); > 0 couples the genes of a shard */
    int32_t n_layers;         /* 2 or 3 count layers */
    int32_t has_efflen;       /* 0: 2-category likelihood (model_TFProb.py:162-167);
                                 1: effLen likelihood (model_TFProb.py:168-185) */
    int32_t intercept_mode;   /* 0 = 'gene' (1,Ng) intercept and sigma; 1 = 'cell' (Nc,1) (model_TFProb.py:53-60) */
    int32_t train_intercept;  /* 1: intercept is a variable clipped to [-9,9] (model_TFProb.py:67-69) */
    int32_t train_sigma;      /* 1: sigma_log is a variable (model_TFProb.py:73-75) */
    int32_t sharded;          /* 1 = this handle is ONE gene shard of a coupled fit (Kg > 0 or cell
                                 mode); its per-cell statistics must be all-reduced every step (see below) */
    uint64_t seed;            /* key of the Philox4x32-10 noise stream */
} brie_problem;

#define BRIE_MAX_KC 8         /* cell features fused into the streaming kernel (Xc row in SGPRs) */
#define BRIE_MAX_KC_WIDE 64   /* wider designs: W tile in LDS for Xc.W, hand-written fp32 MFMA kernel for Xc^T.r */
#define BRIE_MAX_KC_PANELS 1024   /* beyond 64: Xc.W and Xc^T.r in panels of 64 features around the streaming kernel, which
                                     exchanges them through one extra array (the reference has no limit,
                                     model_TFProb.py:84,122-123; no BASELINE config has more than 5) */
#define BRIE_MAX_KG 4          /* gene features kept in registers; above: Xg tile in LDS */
#define BRIE_MAX_KG_WIDE 64
#define BRIE_MAX_KG_PANELS 1024   /* beyond 64: Wg_loc.Xg^T and r.Xg in panels of 64 features around the streaming kernel, like
                                     a cell design beyond 64 (the reference has no limit, model_TFProb.py:85,124-125) */

typedef struct brie_handle brie_handle;

/* BRIE2.__init__ (model_TFProb.py:42-85): allocate the shard's state in HBM. */
int brie_create(const brie_problem *problem, brie_handle **out);
int brie_destroy(brie_handle *h);

/* The next model of a likelihood-ratio test (model_wrap.py:155-187: one fresh BRIE2 per tested feature, all of them
 * fitted to the SAME count layers): the handle keeps its count layers exactly as they sit in HBM -- uploaded,
 * pseudo-counted and compacted once -- and everything that depends on the design width and the seed is replaced
 * (Xc, Wc_loc and its moments, the wide-design buffers; iteration, draw counter, loss ring, gene mask are reset).
 * Afterwards: brie_upload(BRIE_XC), brie_init_state, fit.  Nc, Ng, the layers, effLen, Kg / Xg and the intercept mode
 * stay as created. */
int brie_reconfigure(brie_handle *h, int32_t Kc, uint64_t seed, int32_t train_intercept, int32_t train_sigma);

/* Copy a host or device matrix into the shard (densified fp32 layers of
 * model_wrap.py:108-111; Xc of model_TFProb.py:220; init_obj of :62-65). */
int brie_upload(brie_handle *h, int which, const float *src,
                int64_t rows, int64_t cols, int64_t ld);

/* A count layer in the element type the caller holds it in (the reference casts whatever it gets on the host,
 * .astype(np.float32): io_utils.py:18, model_wrap.py:111 -- a single-threaded pass over 4-8 GB at configs[2]).  Host
 * memory only; the staged pipeline's threads convert row slabs to u16 when they hold nothing but integers in
 * [0, 65535], else to the float32 the reference's cast would give; the layer ends up bit-identical to
 * brie_upload(astype(float32)).  `ld` in ELEMENTS of the source type. */
typedef enum brie_dtype {
    BRIE_DT_F32 = 0, BRIE_DT_F64 = 1, BRIE_DT_I32 = 2, BRIE_DT_I64 = 3, BRIE_DT_U8 = 4, BRIE_DT_U16 = 5, BRIE_DT_I16 = 6,
    BRIE_DT_U32 = 7
} brie_dtype;
int brie_upload_typed(brie_handle *h, int which, const void *src, int32_t dtype, int64_t rows, int64_t cols, int64_t ld);

/* A count layer given as scipy-style compressed sparse (format 0 = CSC: indptr per gene, indices =
 * cells; 1 = CSR: indptr per cell, indices = genes; int64 indptr, int32 indices, fp32 data; host or
 * device pointers).  Densified ON THE DEVICE (duplicates summed) -- the reference densifies on the
 * host with .toarray() (model_wrap.py:108-111, model_TFProb.py:135-137). */
int brie_upload_sparse(brie_handle *h, int which, int32_t format, const int64_t *indptr,
                       const int32_t *indices, const float *data, int64_t nnz, int64_t rows, int64_t cols);

/* model_wrap.py:113-117: where count1+count2 > 0 add `pseudo_count` to BOTH
 * unique layers -- applied to the device copy, caller arrays are untouched. */
int brie_add_pseudo_count(brie_handle *h, float pseudo_count);

/* Model_init (model_TFProb.py:12-31) drawn from the shared Philox stream
 * (draw id 0xFFFFFFFF; k=0 Z_loc, 1 log Z_std, 2 Wc_loc, 3 intercept).
 * `intercept` / `sigma`: NaN = reference default (N(0,1) / ones), otherwise the
 * constant of model_TFProb.py:20,25. */
int brie_init_state(brie_handle *h, float intercept, float sigma);

/* A fresh tf.optimizers.Adam (model_TFProb.py:237): zero moments, iteration 0. */
int brie_reset_optimizer(brie_handle *h);

/* tfp.math.minimize(loss_fn, num_steps, optimizer) (model_TFProb.py:239-241,
 * 255-257) with Keras Adam(lr) + the clip[-9,9] constraints (:69,81).
 * `loss_trace` (host, n_steps floats, may be NULL) receives the ELBO loss
 * BEFORE each update.  With loss_trace == NULL the call only enqueues work.
 * The FIRST step of a handle that streams >= 256 MiB per step also compacts integer count layers and runs the placement
 * search ("Placement of the streamed arrays" below): it blocks for 0.1 - 3 s and transiently holds up to seven further
 * copies of the handle's streamed arrays (bounded by the free HBM read at that moment; BRIE_PLACEMENT_TRIES=1 switches
 * it off).  The search is best effort -- it never fails the step -- and changes addresses only, never values. */
int brie_step(brie_handle *h, int32_t n_steps, float lr, int32_t mc_size,
              float *loss_trace);

/* MANY steps per launch for small inputs.  One step is two dependent launches (the streaming pass, then the per-gene reduction
 * and Adam): 11 - 13 us whatever they compute, which is what a step of a few hundred cells costs (BASELINE configs[0]; the
 * reference's own 10-gene batches, model_wrap.py:242).  brie_step can run its n_steps as ONE launch when the model allows it --
 * uncoupled, Kc <= 8, ELBO target, no frozen gene, MC_size 1 or 3, n_steps >= 2, the whole grid resident: the workgroups of a
 * gene block meet at a barrier per step (rows published write-through, read past L1) and each applies the per-gene update
 * itself.  State, loss trace and loss ring are bit-identical to the two-launch path.  It is AUTOMATIC only where it measured
 * faster: at most 16 cell chunks (Nc <= 256) and either 5 x workgroups <= CUs or at most 8 gene blocks (2 048 genes: each gets
 * an XCD of its own) with Kc <= 3 -- 8.0 against 11.3 us per step at configs[0].
 * brie_set_step_fusion: -1 automatic (default; BRIE_FUSE_STEPS=0 / 1 overrides), 0 never, 1 whenever the model allows it.
 * brie_step_fusion_info: launches of that kind so far and the steps they carried. */
int brie_set_step_fusion(brie_handle *h, int32_t mode);
int brie_step_fusion_info(const brie_handle *h, int64_t *launches, int64_t *steps);
/* Tests / experiments only (an explicit call on one handle, nothing in the environment; results are then WRONG): flags of the
 * fused launches -- 1 no barrier wait, 2 no per-gene update, 4 no rows (phase timing, profiles/fuse_debug.py), 8 chunk 0 skips
 * its second arrival; bits 8.. = log2 of the barrier's poll bound (default 24, ~10 s: a workgroup that never arrives ends the
 * kernel with a flag the next call turns into BRIE_ERR_HIP instead of a kernel spinning for ever). */
int brie_debug_step_fusion(brie_handle *h, int32_t flags);

/* Per-batch convergence.  The reference fits ~batch_size/Nc genes at a time and lets every batch stop on
 * its own windowed loss (model_wrap.py:241-260 + model_TFProb.py:247-258).  All genes are fitted
 * concurrently here, so the host reads the per-gene losses of the last steps, decides per batch and
 * freezes the genes of finished batches: a frozen gene keeps state, moments and its last loss; a
 * 256-gene block with no active gene is skipped by the kernels.
 * brie_set_gene_mask: active[Ng] bytes (1 = train, 0 = frozen), NULL = all active.
 * brie_read_loss_window: out[n_last][Ng] = per-gene loss (KL - ll) of the last n_last <= 128 steps. */
int brie_set_gene_mask(brie_handle *h, const uint8_t *active);
int brie_read_loss_window(brie_handle *h, int32_t n_last, float *out);

/* Objective of brie_step / brie_loss_gene: 0 = "ELBO" (default; model_TFProb.py:206-211),
 * 1 = "marginLik" (model_TFProb.py:156-157,188-189,202-205: z sampled from the prior, log-mean-exp
 * over the MC samples, no KL; only Wc_loc / intercept / sigma_log are updated). */
int brie_set_target(brie_handle *h, int32_t target);

/* Gene-sharded COUPLED fits (Kg > 0 or intercept_mode 'cell'; SURVEY 8e "when it does not shard
 * freely"): the per-cell parameters are replicated on every rank and need the sum over ALL genes of
 * the per-cell statistics before their Adam update.  One step is then
 *     brie_step_begin(h, lr, mc)            main pass + per-gene Adam + local per-cell sums
 *     all-reduce(sum) the statistics buffer (RCCL; brie_rowstat_buffer reports it and its size,
 *                                            (max(4, Kg rounded up to 4) + 2) * Nc floats laid out as
 *                                            [Nc][kgp] Wg_loc gradient | [Nc] | [Nc]; or a caller-owned
 *                                            device buffer registered with brie_set_rowstat_buffer)
 *     brie_step_end(h, &loss)               Adam for Wg_loc / per-cell intercept / sigma; local loss
 * brie_step() = begin + end without the exchange (single shard). */
int brie_step_begin(brie_handle *h, float lr, int32_t mc_size);
int brie_rowstat_buffer(brie_handle *h, float **dev, int64_t *n_floats);
int brie_set_rowstat_buffer(brie_handle *h, float *dev);
int brie_step_end(brie_handle *h, float *loss);

/* ---- Communication (SURVEY 8b sketch, 8e): RCCL over xGMI, one communicator per process ------------------
 * The reference has no communication layer; its only scale-out device is sequential gene batching
 * (model_wrap.py:241-260).  Here genes are sharded over the GPUs of a node, one process (rank) per GPU:
 *   rank 0: brie_comm_unique_id(id)  ->  the caller hands `id` (BRIE_COMM_ID_BYTES) to every rank by its own
 *   means (file, socket, MPI, torch.distributed store)  ->  every rank: brie_comm_init(device, rank, world, id, &c).
 * librccl.so.1 is bound at run time; without it these calls return BRIE_ERR_COMM and the rest of the library works.
 *  brie_comm_allgather: every rank contributes `count` floats, recv gets world*count in rank order -- the
 *                       end-of-fit gather of per-gene vectors [Wc_loc, intercept, sigma, loss_gene] ("RCCL weight
 *                       all-gather" of BASELINE configs[3]); 60-120 KB per rank, latency-bound.
 *  brie_comm_allreduce: in place over `count` elements (dtype BRIE_F32 / BRIE_F64, op BRIE_SUM / MAX / MIN) -- the
 *                       loss-trace windows of the global convergence rule (model_TFProb.py:250), timings.
 *  Buffers may be host or device pointers; both calls return when the result is visible to the caller.
 *  brie_attach_comm:    a handle created with sharded=1 (one gene shard of a COUPLED fit, Kg > 0 or
 *                       intercept_mode 'cell') all-reduces its per-cell statistics INSIDE brie_step /
 *                       brie_step_end, on the handle's own stream, between the local reduction and the Adam
 *                       update of Wg_loc / per-cell intercept / sigma -- no host round trip per step.
 *                       comm == NULL detaches.  The loss trace stays LOCAL (sum it with brie_comm_allreduce). */
#define BRIE_COMM_ID_BYTES 128
enum { BRIE_F32 = 0, BRIE_F64 = 1 };
enum { BRIE_SUM = 0, BRIE_MAX = 1, BRIE_MIN = 2 };
typedef struct brie_comm brie_comm;
/* brie_comm_available: BRIE_OK when librccl binds with every symbol this library calls and `device` exists -- the check
 * every rank makes (and the ranks agree on) BEFORE rank 0 publishes a unique id: a rank that cannot even load RCCL must
 * not leave the others waiting inside ncclCommInitRank.  A failure of ncclCommInitRank itself remains collective. */
int brie_comm_available(int32_t device);
int brie_comm_unique_id(uint8_t *id_out /* BRIE_COMM_ID_BYTES */);
int brie_comm_init(int32_t device, int32_t rank, int32_t world, const uint8_t *unique_id, brie_comm **out);
int brie_comm_destroy(brie_comm *c);
int brie_comm_rank(const brie_comm *c);
int brie_comm_world(const brie_comm *c);
int brie_comm_allgather(brie_comm *c, const float *send, int64_t count, float *recv);
int brie_comm_allreduce(brie_comm *c, void *buf, int64_t count, int32_t dtype, int32_t op);
int brie_attach_comm(brie_handle *h, brie_comm *c);

/* model_TFProb.py:261-264: mean over `n_repeats` stochastic evaluations of
 * get_loss(axis=0) with MC_size = 1 -> out[Ng] (host). */
int brie_loss_gene(brie_handle *h, int32_t n_repeats, float *out);

/* BRIE2.logLik_MC (model_TFProb.py:130-191) per element: out[Nc][Ng] (host or device, `ld` elements between
 * rows) = mean over `size` samples z ~ q of the log-likelihood (target ELBO, :159,191), or their log-mean-exp
 * with z ~ prior (target marginLik, :157,188-189).  Consumes one noise draw id (samples k = 0..size-1). */
int brie_loglik_mc(brie_handle *h, int32_t size, float *out, int64_t ld);

/* BRIE2.get_loss(count_layers, target, axis, MC_size) (model_TFProb.py:194-211) as ONE stochastic evaluation on the
 * current state, no update: reduce_sum(KL, axis) - reduce_sum(mean_k ll(z_k), axis) for target ELBO, -reduce_sum(
 * log-mean-exp_k ll(z_k), axis) for marginLik (:202-205), each sum in fp64.  axis 0 -> out[Ng] (per gene), axis 1 ->
 * out[Nc] (per cell); the scalar of axis=None is the sum of either.  Noise: ELBO takes its mc_size samples at mc_size
 * consecutive draw ids (the loss is linear in them: identical to brie_loss_gene(mc_size) per gene); marginLik takes
 * them at ONE draw id (k = 0..mc_size-1), like a step with that MC_size.  An accessor (per-entry temporaries). */
int brie_get_loss(brie_handle *h, int32_t mc_size, int32_t axis, float *out);

/* BRIE_RV.__init__ (model_wrap.py:18-40): copy state / derived arrays out. */
int brie_read(brie_handle *h, int which, float *dst,
              int64_t rows, int64_t cols, int64_t ld);

/* BRIE_RV.__init__ reads Psi, Z_std, Psi95CI and Z_loc one after the other (model_wrap.py:28-35): 16 GB at
 * configs[2].  brie_read_results_async exports all four in ONE pass over the state (any destination may be NULL),
 * slab by slab on a second stream, and returns at once: the copies run while the caller goes on with
 * brie_loss_gene (which only reads the state).  brie_read_wait blocks until the destinations are complete; any
 * call that changes the state waits for a pending read first.  Destinations: (Nc, Ng) row-major, `ld` elements
 * between rows, host or device.  For host destinations the copy engine only reaches PCIe speed into page-locked
 * memory: brie_host_register / brie_host_unregister pin / unpin a caller-owned buffer (hipHostRegister) -- callable
 * from another host thread while the fit runs, which also takes the first-touch page faults off the critical path. */
int brie_read_results_async(brie_handle *h, float *psi, float *z_std, float *psi95ci, float *z_loc, int64_t ld);
int brie_read_wait(brie_handle *h);
int brie_host_register(void *ptr, int64_t bytes);
int brie_host_unregister(void *ptr);

/* Host ingest of the count layers (the reference densifies and casts on the host, model_wrap.py:108-111, and hands
 * TF pageable fp32 arrays).  brie_upload of a count layer that lives in PAGEABLE host memory and holds >= 16 M entries
 * goes through a staged pipeline: a few host threads convert row slabs to u16 in page-locked buffers (a slab holding
 * anything but integers in [0, 65535] travels as fp32), each slab's copy runs asynchronously while the next is being
 * converted, a kernel writes it into the tiled fp32 layer.  Bit-identical to the plain copy; BRIE_INGEST=direct|staged
 * forces either, BRIE_INGEST_THREADS sets the thread count (default: the cores of the process, at most 8).
 * brie_host_convert_u16 is the host half on its own (tests, host-bandwidth measurements; no GPU involved):
 * dst[r][c] = (uint16) src[r*ld + c], *not_integral = 1 when some value is not a non-negative integer <= 65535. */
int brie_host_convert_u16(const float *src, int64_t rows, int64_t cols, int64_t ld, uint16_t *dst, int32_t *not_integral);
/* ... and one slab of a typed source (brie_dtype): dst receives rows x cols u16 (*is_f32 = 0: every value an integer in
 * [0, 65535]) or the float32 cast of the values (*is_f32 = 1); dst must hold rows x cols x 4 bytes. */
int brie_host_convert_slab(const void *src, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, void *dst, int32_t *is_f32);

/* Noise-draw counter (one draw id per loss evaluation). */
int brie_get_draw(brie_handle *h, uint32_t *draw);
int brie_set_draw(brie_handle *h, uint32_t draw);

/* Block until all enqueued work of this handle is done. */
int brie_synchronize(brie_handle *h);

/* Per-launch HIP-event timing of the dominant kernel (elbo_adam_step) on the
 * handle's own stream, for bench.py's roofline.  enable=1 starts/clears. */
int brie_profile_enable(brie_handle *h, int32_t enable);
int brie_profile_read(brie_handle *h, double *kernel_ms_total, int64_t *n_launches);

/* Measurement aid: device address of a cell x gene array of the handle (which = 0..2 count layers as stored,
 * BRIE_Z_LOC, BRIE_Z_STD_LOG, 20..23 = Adam moments m, v of Z_loc and m, v of Z_std_log). */
int brie_debug_address(brie_handle *h, int which, uint64_t *addr);

/* Tuning knob of the tiling: cells one workgroup streams for its 256-gene block (0 = library default: a function of Nc only,
 * so that a gene's fp32 partial sums -- and its whole trajectory -- are the same in any gene shard; 128 or 256 chunks for
 * Nc >= 6144, which keeps every round of workgroups full whatever the shard's gene count). */
int brie_set_tiling(brie_handle *h, int32_t rows_per_chunk);

/* Algorithmic HBM bytes of one elbo_adam_step launch: Nc*Ng*(48 + 4*n_layers) (fp32 model of
 * SURVEY 8d), and the bytes the CURRENT storage moves: Nc*Ng*(48 + {1,2}*n_layers) once integer
 * counts have been compacted to one / two bytes per element (results are bit-identical either way). */
int64_t brie_step_algorithmic_bytes(const brie_handle *h);
int64_t brie_step_storage_bytes(const brie_handle *h);

/* Count-layer storage (the reference densifies to fp32, model_wrap.py:108-111).
 * mode 0 = auto (default): integer counts in [0,255] are kept as u8, in [0,65535] as u16; the
 * pseudo-count of model_wrap.py:113-117 is then applied in registers; mode 1 = always fp32.
 * brie_get_count_storage: 0 = fp32, 1 = u8, 2 = u16, 3 = u8 or u16 per gene quad (4 consecutive genes; decided at
 * brie_add_pseudo_count / first step; a quad holding a count > 255 keeps two bytes per count, the others one). */
int brie_set_count_storage(brie_handle *h, int32_t mode);
int brie_get_count_storage(const brie_handle *h);

/* Measurement utility (no reference counterpart): HBM rate in GB/s of a kernel that only reads
 * `n_read` and writes `n_write` 16-B-vector streams -- (1,1) = copy, (8,6)/(9,6) = the access mix of
 * elbo_adam_step with 2/3 count layers; lds_bytes_per_block > 0 caps the occupancy (waves per CU),
 * a negative value -(lds+1) selects non-temporal loads/stores.
 * bench.py reports the fused kernel against this ceiling. */
int brie_calibrate_stream(int32_t device, int32_t n_read, int32_t n_write, int64_t bytes_per_stream,
                          int32_t iters, int32_t lds_bytes_per_block, double *gbps);

/* Placement of the streamed arrays (no reference counterpart; the reference leaves memory to TensorFlow).  How fast a
 * handle's step kernel streams depends on where the allocator put its arrays: byte-identical code on identical data runs
 * at 8.1 or at 9.5 ms per step at configs[2] -- arrays that lie next to each other in physical
 * memory stream slower than arrays that lie far apart.  The library therefore measures: a probe kernel with the step
 * kernel's traffic and no effect on the data is timed on the handle's arrays; while its rate (GB/s of
 * brie_step_storage_bytes) is below `good_gbs`, candidate sets are allocated in rounds of three, INTERLEAVED (array 0 of
 * every candidate, then array 1 of every candidate, ...: the arrays of one set end up a few arrays apart; one candidate at a
 * time for arrays below 1 GiB), filled by device-to-device copies and probed; every set is held until the search is over
 * (a freed set is what the allocator hands out again), then the fastest is kept and the others are freed (results are
 * bit-identical: only addresses change).  brie_step does this by itself before the first step of a handle that streams
 * >= 256 MiB per step (see brie_step): BRIE_PLACEMENT_TRIES sets in all (default: ONE round = 4 sets for arrays of a gigabyte
 * and more, BRIE_PLACEMENT_MAX_SETS = 8 for smaller arrays whose candidates cost a few GB and milliseconds; 1 = off;
 * brie_placement_configure sets it, the memory fraction and the time limit per handle),
 * good_gbs = BRIE_PLACEMENT_GOOD_GBS (default 6050 -- 5850 for handles whose arrays are below 1 GiB, which never read
 * faster than 6.0 TB/s --, or 0.97 x the best rate a handle of that size class has reached in this process on the device
 * when that is higher; a first set of a large handle between 6050 and 6150 still buys ONE round of candidates).
 *  BOUNDS.  Memory: before every round hipMemGetInfo is read afresh and the round takes at most BRIE_PLACEMENT_HBM_FRACTION
 *  (default 0.8) of the HBM that is free beyond a 2-GiB reserve; with the sets of earlier rounds still held the transient
 *  peak is (sets held) x brie_step_storage_bytes-worth of arrays -- at configs[2] 26 GB per set, 78 GB after one round,
 *  182 GB if all seven candidates are needed -- and it is reported by brie_placement_status.  A process that shares the GPU
 *  and cannot spare that sets BRIE_PLACEMENT_TRIES or BRIE_PLACEMENT_HBM_FRACTION lower.  Time: no new candidate once
 *  BRIE_PLACEMENT_SECONDS (default 3) have passed.  Typical cost at configs[2]: 0.17 s; 0.8 - 3 s when the allocator has a
 *  slow moment.
 *  BEST EFFORT.  Nothing that fails inside the search (probe launch, event, allocation, copy) fails the caller: the fastest
 *  complete set is re-adopted, the candidates are freed, the HIP error is cleared and the reason is kept for
 *  brie_placement_status.  The measurements are dropped (and the next step searches again) when the streamed arrays are
 *  replaced: counts re-uploaded and compacted again, count storage changed, count tiers unified when gene quads are packed.
 *  brie_placement_probe : rate of the probe on the arrays as they are (iters timed launches after one warm-up).
 *  brie_placement_tune  : the procedure above on demand, at most max_tries <= BRIE_PLACEMENT_MAX_SETS sets.
 *  brie_placement_info  : sets probed so far, which one is in use (0 = the original), their rates
 *                         (the first n_gbs of them into gbs[n_gbs]; BRIE_PLACEMENT_MAX_SETS is always enough) and the seconds spent
 *                         probing, allocating and copying.
 *  brie_placement_status: how the last search ended (BRIE_PLACEMENT_*), its peak transient holding in bytes and a
 *                         human-readable note (e.g. "no set reached the stop rate: best of 8 sets 5210 GB/s ...").
 *  Any out-pointer may be NULL. */
#define BRIE_PLACEMENT_MAX_SETS 8
typedef enum brie_placement_state {
    BRIE_PLACEMENT_NOT_RUN = 0,          /* no search yet (or its measurements were dropped) */
    BRIE_PLACEMENT_GOOD = 1,             /* a set reached the stop rate */
    BRIE_PLACEMENT_BEST_OF_ALL = 2,      /* every allowed set was probed, none reached the stop rate: the fastest is in use */
    BRIE_PLACEMENT_STOPPED_MEMORY = 3,   /* free HBM did not allow another round */
    BRIE_PLACEMENT_STOPPED_TIME = 4,     /* BRIE_PLACEMENT_SECONDS passed */
    BRIE_PLACEMENT_STOPPED_ERROR = 5,    /* something failed inside the search (note says what); the step goes on */
    BRIE_PLACEMENT_OFF = 6               /* handle below 256 MiB per step, or BRIE_PLACEMENT_TRIES <= 1 */
} brie_placement_state;
int brie_placement_probe(brie_handle *h, int32_t iters, double *gbs);
int brie_placement_tune(brie_handle *h, int32_t max_tries, double good_gbs);
int brie_placement_info(const brie_handle *h, int32_t *tries, int32_t *kept, double *gbs, int32_t n_gbs, double *seconds);
/* Per-handle limits of the search, before the process environment: at most max_sets sets in all (<= BRIE_PLACEMENT_MAX_SETS;
 * 1 = no search), a round takes at most hbm_fraction of the free HBM, no new allocation after max_seconds; a value <= 0 keeps
 * the default.  Several ranks that share ONE GPU size their rounds on the same hipMemGetInfo reading without knowing of each
 * other -- their caller does: it hands every rank 0.8 / (ranks on the device). */
int brie_placement_configure(brie_handle *h, int32_t max_sets, double hbm_fraction, double max_seconds);
/* Tests only (no environment variable can switch it on): the NEXT search of this handle fails at point 1 = its first probe,
 * 2 = a candidate allocation, 3 = a candidate's probe, 4 = a candidate copy; 0 = none.  The search is best effort: the step
 * goes on, bit-identical (tests/test_gpu_placement.py). */
int brie_debug_inject_placement_failure(brie_handle *h, int32_t point);
int brie_placement_status(const brie_handle *h, int32_t *status, int64_t *peak_bytes, char *note, int32_t note_len);

/* Experiment aid: one slab of slab_bytes, the eight streamed arrays of a 2-layer u8-count problem of (Nc, Ng) placed at the
 * caller's byte offsets (offsets[n_layouts][8]: six state arrays, two count layers; multiples of 16), the placement probe
 * timed for every layout on the SAME physical memory -> gbs[n_layouts].  Separates "where the memory is" from "how the
 * arrays are spaced" (profiles/layout_probe.py). */
int brie_probe_layouts(int32_t device, int64_t Nc, int64_t Ng, int64_t slab_bytes, int32_t n_layouts, const int64_t *offsets,
                       int32_t iters, double *gbs);

/* Experiment aid (round 6): the streamed arrays of a u8-count problem built from the HIP virtual-memory API (hipMemAddressReserve /
 * hipMemCreate / hipMemMap) with physical chunks of chunk_bytes[l] created in order[l] -- 0: array after array, 1: round robin over the
 * arrays (interleaved), 2: plain hipMalloc per array -- and the placement probe timed on each layout -> gbs[n_layouts] (storage bytes
 * of a step / probe time) and seconds[n_layouts] (building the set).  No reference counterpart (profiles/vmm_probe.py). */
int brie_probe_vmm(int32_t device, int64_t Nc, int64_t Ng, int32_t n_layers, int32_t n_layouts, const int64_t *chunk_bytes,
                   const int32_t *order, int32_t iters, double *gbs, double *seconds);

/* Free / total HBM of a device in bytes (hipMemGetInfo).  fitBRIE uses it to split a gene range that does not
 * fit into sequential super-batches -- the role of the reference's batch_size (model_wrap.py:241-260), sized for
 * 288 GB instead of for 500k elements. */
int brie_device_memory(int32_t device, int64_t *free_bytes, int64_t *total_bytes);

/* brie_destroy keeps the cell x gene arrays of the handle (>= 256 MB each) for the next handle of the SAME size on the
 * same device -- sequential fits of one size (the super-batches above, one fit after another in a service) otherwise
 * pay hipFree + hipMalloc of tens of GB again, which is milliseconds on a good day and seconds on a bad one -- and a
 * placement that brie_placement_tune found fast stays in use.  One generation only: a handle of another size,
 * brie_device_memory, a failed brie_create and this call release them, and every large allocation of the library that
 * fails is tried once more after releasing them.  (The fp32 count layers a LIVE handle gives up when it compacts its
 * counts are freed, not cached.)  The cache belongs to the process: another process on the same GPU cannot reclaim it.
 * BRIE_DEVICE_CACHE=0 switches the cache off. */
int brie_trim_memory(void);

/* Count simulator -- brie/models/simulator.py:7-75.  Stateless; every array is C-order (Nc, Ng) fp32 in host or
 * device memory (copied through HBM in row slabs), genes are addressed globally (gene_offset) so a gene shard
 * simulates exactly its columns of the whole matrix.  All draws come from the library's Philox stream keyed by
 * `seed` (the reference uses TensorFlow's unseeded samplers), see DESIGN.md section 7.3.
 *  brie_simulate_psi   : psi = sigmoid(clip(mean_logit + sigma_j * N(0,1), -9, 9))           (simulator.py:31-41)
 *  brie_simulate_counts: (c1,c2,c3) ~ Multinomial(floor(total), phi), phi ~ [psi, 1-psi, 1] * effLen[:, [0,4,5]]
 *                        (simulator.py:45-69); effLen == NULL: two categories, c1 ~ Binomial(total, psi), c2 the rest,
 *                        out3 unused.  Exact sampling (inversion / BTRS) in fp64. */
int brie_simulate_psi(int32_t device, int64_t Nc, int64_t Ng, int64_t gene_offset, uint64_t seed,
                      const float *mean_logit, const float *sigma /* (Ng) */, float *psi_out);
int brie_simulate_counts(int32_t device, int64_t Nc, int64_t Ng, int64_t gene_offset, uint64_t seed,
                         const float *psi, const float *total, const float *effLen /* (Ng, 6) or NULL */,
                         float *out1, float *out2, float *out3);

const char *brie_last_error(void);
int brie_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BRIE_AMD_H */

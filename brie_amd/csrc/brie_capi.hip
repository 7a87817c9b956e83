// brie_capi.hip -- host side of the C ABI declared in include/brie_amd.h.
// One handle = one gene shard resident in the HBM of one MI355X, driven on its
// own HIP stream.  No torch, no C++ types across the boundary.
#include "brie_amd.h"
#define BRIE_HOST_TU 1
#include "brie_launch.h"
#include "brie_tile.hip.h"
#include "brie_comm_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <mutex>
#include <sched.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    // a HIP failure is reported through this return value: do not leave it behind as the runtime's sticky
    // "last error", where the next hipGetLastError() check of an unrelated call would find it
    if (code == BRIE_ERR_HIP) (void)hipGetLastError();
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(BRIE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                               \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// A source in HBM may still be being produced on another stream (e.g. a torch tensor on torch's stream) while the
// handle's stream is non-blocking: such a copy is ordered after ALL prior device work.  Host sources need no such wait.
int order_after_device_source(const void *src) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) != hipSuccess) { (void)hipGetLastError(); return BRIE_OK; }   // plain host memory
    if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) HIP_TRY(hipDeviceSynchronize());
    return BRIE_OK;
}

}  // namespace

// brie_comm.hip reports its failures through the same thread-local message
extern "C" int brie_internal_fail(int code, const char *msg) { return fail(code, "%s", msg); }

struct brie_handle {
    brie_problem p{};
    brie_comm *comm = nullptr;      // attached communicator: sharded coupled fits all-reduce rowstat in-library
    int64_t ld = 0;                 // gene_blocks * 256: pitch of per-gene vectors and of row-major matrices
    int64_t row_stride = 0, gb_stride = 0;   // matrix addressing (see StepScalars)
    int S = 0;                      // statistics per gene per chunk = Kc + 4
    int mode = 0;                   // likelihood mode (brie::kLik2 ...)
    hipStream_t stream = nullptr;
    // (Nc, ld)
    float *c[3] = {nullptr, nullptr, nullptr};
    void *cu[3] = {nullptr, nullptr, nullptr};      // compact (u8 / u16) count layers, same tiled indexing
    int cs = brie::kCountF32;       // current count storage (kCountMixed: u8 or u16 per gene quad)
    // tiers per gene quad (brie::TierTables): a row of a gene block's count tile is the concatenation of its quads' 4- or
    // 8-byte pieces; ONE launch, every lane picks up its quad's width and offset at run time
    std::vector<uint8_t> q_esz_host;        // per quad: 1 or 2 bytes per count
    brie::TierTables tt{nullptr, nullptr, nullptr, nullptr};     // device tables
    size_t tier_layer_bytes = 0;            // size of one tiered count layer
    bool allow_compact = true;      // BRIE_COUNT_STORAGE=f32 / brie_set_count_storage(h, 1) disable it
    bool compact_tried = false;
    float pc = 0.0f;                // pseudo-count applied in registers when cs == kCountU8
    float *mu = nullptr, *rho = nullptr, *m_mu = nullptr, *v_mu = nullptr, *m_rho = nullptr,
          *v_rho = nullptr;
    float *Xc = nullptr;            // (Nc, Kc)
    float *W = nullptr, *m_W = nullptr, *v_W = nullptr;              // (Kc, ld)
    float *b = nullptr, *m_b = nullptr, *v_b = nullptr;              // (ld)
    float *lam = nullptr, *m_lam = nullptr, *v_lam = nullptr;        // (ld)
    float *effL = nullptr;          // (6, ld)
    // coupled modes: gene features (Kg > 0) and/or per-cell intercept / sigma (intercept_mode 'cell')
    bool coupled = false, cell_mode = false;
    int kgp = brie::kKgMax;         // pitch of a Wg_loc row / rows of Xg: 4 for Kg <= 4, else Kg rounded up to 4
    bool gwide = false;             // Kg > 4: Xg tile in LDS (GW kernel variant)
    float *Xg = nullptr;            // (kgp, ld) transposed gene features
    float *Wg = nullptr, *m_Wg = nullptr, *v_Wg = nullptr;           // (Nc, kgp)
    float *cb = nullptr, *m_cb = nullptr, *v_cb = nullptr;           // (Nc)
    float *clam = nullptr, *m_clam = nullptr, *v_clam = nullptr;     // (Nc)
    float *row_partials = nullptr;  // (gene_blocks, (kgp + 2) * Nc), see brie::CoupledArgs
    float *rowstat = nullptr;       // ((kgp + 2) * Nc)
    bool have_xg = false;
    float *rowstat_ext = nullptr;   // caller-owned (6, Nc) buffer used instead of rowstat (multi-GPU all-reduce)
    int target = 0;                 // 0 = "ELBO", 1 = "marginLik" (model_TFProb.py:194-211)
    // wide cell designs (Kc > BRIE_MAX_KC): W tile in LDS for Xc.W, MFMA kernel for Xc^T.r
    bool wide = false;
    bool vwide = false;             // Kc > BRIE_MAX_KC_WIDE or Kg > BRIE_MAX_KG_WIDE: 64-feature panels (see setup_paths)
    bool vgwide = false;            // Kg > BRIE_MAX_KG_WIDE: Wg_loc . Xg^T and r . Xg in panels as well
    float *XgT = nullptr;           // (ld, kgp) gene-major copy of Xg (Kg > 64: the A operand of gene_design_grad)
    float *Xg_zero = nullptr;       // (4, ld) / (Nc, 4) zeros: what the register slots of the coupled step variant read
    float *Wg_zero = nullptr;       //   when the gene design goes through the panels
    int part_kgp = brie::kKgMax;    // pitch of the Wg part of a row chunk (= kgp unless vgwide)
    int kernel_kc = 0;              // KC of the kernel instantiation (0 for wide designs)
    // Wide designs on the matrix cores (brie_tile.hip.h): Kc > 8 and / or Kg > 4 with forward and backward products
    // fused into the streaming pass.  wide_like = the per-gene statistics carry no Xc rows (S = 4), Wc_loc is updated
    // from Gpart by wide_w_adam and loss_gene takes Xc.Wc_loc from Mbuf -- true for `wide` and for every tile handle.
    bool tile = false, wide_like = false;
    int tile_lds = 0, tile_nacc = 0, tile_njt = 0, tile_nw = 1;     // tile_nw: 4-wave halves per workgroup (1 or 2)
    int tile_kcr = 0;               // 4: gene-feature model with <= 4 cell features, kept in registers (brie_tile.hip.h)
    size_t gpart_elems = 0, rbuf_elems = 0;
    float *win_scratch = nullptr;   // brie_read_loss_window staging
    size_t win_elems = 0;
    float *Mbuf = nullptr;          // (Nc, ld) tiled: Xc.Wc_loc for loss_gene_eval (allocated on first use)
    float *Rbuf = nullptr;          // (Nc, ld) tiled: residual r written by the step kernel
    float *Gpart = nullptr;         // (n_gchunks, Kc, ld): per-chunk partial sums of Xc^T . r
    int gchunk_rows = 1024, n_gchunks = 0;
    bool step_open = false;         // between brie_step_begin and brie_step_end
    brie::CellFinalizeArgs pending_cf{};
    float *gene_tmp = nullptr;      // (ld) scratch per-gene output
    float *gene_active = nullptr;   // (ld) per-gene train mask (per-batch convergence), default all ones
    int32_t *block_active = nullptr;   // (gene_blocks)
    float *ring_kl = nullptr, *ring_ll = nullptr;   // (kLossRing, ld) per-gene loss terms of the last steps
    int64_t ring_pos = 0;           // optimisation steps taken so far
    // packing of active gene quads (per-batch convergence): position -> original quad
    int32_t *quad_ids = nullptr;    // device (ld / 4)
    std::vector<int32_t> perm;      // host mirror; empty or identity when not packed
    std::vector<float> mask_host;   // gene mask in ORIGINAL gene order (ld entries)
    bool packed = false;
    bool allow_pack = true;         // BRIE_PACK_ACTIVE=0 keeps frozen genes in place (A/B, tests)
    void *pack_scratch = nullptr;   // one matrix-sized scratch buffer for the gathers (swaps with live arrays)
    size_t pack_scratch_bytes = 0;
    float *row_scratch = nullptr;   // scratch for per-gene row sets
    float *io_scratch = nullptr;    // (Nc, ld) staging buffer of brie_read, allocated on first use and kept
    // asynchronous result read-back (brie_read_results_async): second stream + slab staging
    hipStream_t io_stream = nullptr, io_stream2 = nullptr;     // slab k of an export runs on stream k & 1
    hipEvent_t io_event = nullptr;
    float *io_slab = nullptr;
    size_t io_slab_elems = 0;
    bool io_pending = false;
    // the slab loop runs on a worker thread: a copy into PAGEABLE host memory blocks its caller (the runtime stages
    // it), and the caller of brie_read_results_async wants to go on with brie_loss_gene
    std::thread io_thread;
    int io_rc = 0;
    std::string io_err;
    float *partials = nullptr;
    size_t partials_elems = 0;
    double *loss_parts = nullptr;
    size_t loss_parts_elems = 0;
    int rows_per_chunk = 0, n_chunks = 0, gene_blocks = 0, fin_blocks = 0;
    int user_rows_per_chunk = 0;
    uint32_t draw = 0;
    int64_t t = 0;                  // Adam iteration of the current optimiser
    bool have_c[3] = {false, false, false}, have_xc = false, have_eff = false, have_state = false;
    // profiling of the dominant kernel
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
    // placement of the streamed arrays (brie_placement_tune): rates of the sets that were probed, which one was kept
    bool placement_done = false;
    int placement_tries = 0, placement_kept = 0, placement_status = BRIE_PLACEMENT_NOT_RUN;
    double placement_gbs[BRIE_PLACEMENT_MAX_SETS] = {};
    double placement_seconds = 0.0;
    int64_t placement_peak_bytes = 0;       // largest transient holding of candidate sets during a search
    char placement_note[192] = {0};         // why the search ended the way it did (brie_placement_status)
    // many steps per launch (PERSIST variant of the step kernel; run_steps_persist)
    int persist_mode = -1;                  // brie_set_step_fusion: -1 automatic, 0 never, 1 whenever the model allows it
    int persist_debug = 0;                  // brie_debug_step_fusion
    bool any_frozen = false;                // a gene mask with frozen genes is set
    brie::PersistArgs *persist_args = nullptr;
    float *persist_alphas = nullptr, *partials2 = nullptr;
    size_t persist_alphas_elems = 0, partials2_elems = 0;
    uint32_t *persist_barrier = nullptr;
    uint32_t *persist_flag_host = nullptr;  // pinned: the time-out word of the last fused launch, copied back behind it
    std::vector<float> persist_alphas_host; // host sources of the per-call uploads (alive until persist_copy_event has passed)
    brie::PersistArgs persist_args_host{};
    hipEvent_t persist_copy_event = nullptr;
    int64_t persist_launches = 0, persist_steps = 0;
    int persist_serial = -1;                // which fused-launching handle of this process this is (column offset of its gene blocks)
    int placement_cfg_sets = 0;             // brie_placement_configure: 0 = the library's default
    double placement_cfg_frac = 0.0, placement_cfg_seconds = 0.0;
    int placement_inject = 0;               // brie_debug_inject_placement_failure (tests): fails ONE search of this handle
};

namespace {

int set_device(const brie_handle *h) {
    HIP_TRY(hipSetDevice(h->p.device));
    return BRIE_OK;
}

// an asynchronous read-back still exports the state: let it finish before anything changes the state
int io_wait(brie_handle *h) {
    if (h->io_pending) {
        if (h->io_thread.joinable()) h->io_thread.join();
        h->io_pending = false;
        HIP_TRY(hipStreamSynchronize(h->io_stream));
        if (h->io_stream2) HIP_TRY(hipStreamSynchronize(h->io_stream2));
        if (h->io_rc != BRIE_OK) {
            const int rc = h->io_rc;
            h->io_rc = BRIE_OK;
            return fail(rc, "asynchronous read-back: %s", h->io_err.c_str());
        }
    }
    return BRIE_OK;
}

// ---- the cell x gene arrays of the last destroyed handle, kept for the next one of the same size ----------------------
// hipMalloc / hipFree of the 4 - 12 GB arrays of a shard are usually milliseconds and sometimes seconds (1.5 - 5.4 s in 3
// of ~40 creations of round 3, 3.0 s for the 96 GB of configs[4]: profiles/history/r3a_ingest_ab_c3.json, r3k_bench_c5_whole_n1.json).
// Sequential fits of one size -- fitBRIE's super-batches, a fit after a bench, the models of an LRT that cannot share a
// handle -- give the arrays back and ask for the same sizes a moment later.  ONE generation is kept: the blocks of the
// last destroyed handle, all of one size on one device; a handle of another size, brie_device_memory and
// brie_trim_memory release them, and so does an allocation failure before it is reported.  BRIE_DEVICE_CACHE=0: off.
struct BigBlockCache {
    std::mutex mu;
    int device = -1;
    size_t bytes = 0;
    std::vector<void *> blocks;
    static constexpr size_t kMinBytes = size_t(256) << 20;
    static constexpr size_t kMaxBlocks = 12;
    static bool enabled() {
        static const bool on = [] { const char *e = getenv("BRIE_DEVICE_CACHE"); return !(e && e[0] == '0'); }();
        return on;
    }
    void trim_locked() {
        if (blocks.empty()) return;
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        for (void *q : blocks) (void)hipFree(q);
        (void)hipSetDevice(cur);
        blocks.clear();
        bytes = 0;
        device = -1;
    }
    void trim() { std::lock_guard<std::mutex> l(mu); trim_locked(); }
    void *take(int dev, size_t n) {
        std::lock_guard<std::mutex> l(mu);
        if (blocks.empty() || dev != device || n != bytes) return nullptr;
        void *q = blocks.back();
        blocks.pop_back();
        return q;
    }
    void give(int dev, size_t n, void *q) {
        if (!q) return;
        std::lock_guard<std::mutex> l(mu);
        if (!enabled() || n < kMinBytes) { (void)hipFree(q); return; }
        if (!blocks.empty() && (dev != device || n != bytes)) trim_locked();     // another size: the old generation goes
        if (blocks.size() >= kMaxBlocks) { (void)hipFree(q); return; }
        device = dev; bytes = n;
        blocks.push_back(q);
    }
    // before a handle of (dev, n) allocates: blocks of any other size are of no use to it
    void prepare(int dev, size_t n) {
        std::lock_guard<std::mutex> l(mu);
        if (!blocks.empty() && (dev != device || n != bytes)) trim_locked();
    }
};
BigBlockCache g_blocks;

// every large device allocation of the library: what the block cache holds counts as free memory, so an allocation
// that fails is tried once more after the cache has been released
hipError_t dev_alloc(void **q, size_t bytes) {
    hipError_t e = hipMalloc(q, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        g_blocks.trim();
        e = hipMalloc(q, bytes);
    }
    return e;
}
template <typename T>
hipError_t dev_alloc(T **q, size_t bytes) { return dev_alloc(reinterpret_cast<void **>(q), bytes); }

// a cell x gene array: from the cache when the last handle left one of this size, else hipMalloc (after an allocation
// failure the cache is released and the allocation tried once more)
int alloc_mat(float **p, size_t elems, int device, hipStream_t s) {
    if (elems == 0) { *p = nullptr; return BRIE_OK; }
    const size_t bytes = elems * sizeof(float);
    void *q = g_blocks.take(device, bytes);
    if (!q) HIP_TRY(dev_alloc(&q, bytes));
    *p = static_cast<float *>(q);
    HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
    return BRIE_OK;
}

int alloc_f32(float **p, size_t elems, hipStream_t s) {
    if (elems == 0) { *p = nullptr; return BRIE_OK; }
    HIP_TRY(dev_alloc(p, elems * sizeof(float)));
    HIP_TRY(hipMemsetAsync(*p, 0, elems * sizeof(float), s));
    return BRIE_OK;
}

int prepare_wide_panels(int device);       // (defined beside launch_panels)

// decide the kernel family of this handle from (Kc, Kg, layout); called at create and at brie_reconfigure
void setup_paths(brie_handle *h) {
    const int Kc = h->p.Kc;
    h->vgwide = h->coupled && h->p.Kg > BRIE_MAX_KG_WIDE;
    h->wide = Kc > BRIE_MAX_KC || h->vgwide;
    const char *wp = getenv("BRIE_WIDE_PATH");         // "lds": the round-1 LDS-broadcast variants (A/B runs)
    const bool want_tile = !(wp && strcmp(wp, "lds") == 0);
    const int kgp = h->coupled ? h->kgp : 0;
    // gene features next to <= 4 cell features: those stay in registers, nothing of the cell design goes through LDS / MFMA
    const char *kre = getenv("BRIE_TILE_KC_REGS");      // "0": the MFMA path for them too (A/B runs)
    const bool kcr = kgp > 0 && Kc > 0 && Kc <= 4 && !(kre && kre[0] == '0');
    const int kcm = kcr ? 0 : Kc;
    // LDS of one workgroup: the read-only Wc_loc / Xg tiles of the gene block + one T tile per 4-wave half.  Two plain
    // workgroups per CU while that is under 80 KB; else ONE workgroup of two independent halves sharing the tiles.
    const int tiles = static_cast<int>(sizeof(float)) * (kcm * brie::kGenesPerBlock + kgp * brie::kXgStride);
    // per half: the T tile and the design rows [Xc | Wg_loc] of the tile's 32 cells
    const int ttile = static_cast<int>(sizeof(float)) * brie::kTileRows * (brie::kTileStride + brie::tile_a_stride(kcm + kgp));
    const char *nhe = getenv("BRIE_TILE_HALVES");      // 1 / 2: force (A/B runs)
    h->tile_nw = nhe ? (atoi(nhe) == 2 ? 2 : 1) : (tiles + ttile <= 80 * 1024 ? 1 : 2);
    h->tile_lds = tiles + h->tile_nw * ttile;
    if (h->tile_lds > 160 * 1024 - 64 && h->tile_nw == 2) { h->tile_nw = 1; h->tile_lds = tiles + ttile; }   // one half still fits
    // (gene features 5..8 stay on the LDS-broadcast variant: measured 1.06 x vs 1.09 x the narrow model's step time)
    const char *mk = getenv("BRIE_TILE_MIN_KG");       // A/B runs: smallest Kg that takes the tile kernel (default 8)
    const int min_kg = mk ? atoi(mk) : 8;
    // Kc > 64 ("very wide"): no single kernel holds the W tile; Xc . Wc_loc and Xc^T . r are formed in 64-feature panels
    // around the LDS-free WIDE variant of the streaming kernel (run_steps, wide_forward_mean, wide_backward)
    // Kg > 64: the same for the gene design (Wg_loc . Xg^T into the same buffer, r . Xg by gene_design_grad); whatever
    // cell design stands next to it then goes through the panels too, one family of kernels
    h->vwide = Kc > BRIE_MAX_KC_WIDE || h->vgwide;
    h->tile = want_tile && !h->vwide && (h->wide || (h->gwide && h->p.Kg >= min_kg)) && h->tile_lds <= 160 * 1024 - 64;
    h->wide_like = h->wide || h->tile;
    h->tile_kcr = kcr ? 4 : 0;
    h->tile_nacc = kcm == 0 ? 0 : (Kc <= 32 ? 1 : 2);
    h->tile_njt = !h->coupled ? 0 : (kgp <= 32 ? 1 : 2);
    h->kernel_kc = h->wide_like ? 0 : Kc;
    h->S = h->kernel_kc + 4;
    h->n_gchunks = static_cast<int>((h->p.Nc + h->gchunk_rows - 1) / h->gchunk_rows);
}

int ensure_f32(float **p, size_t *have, size_t need, hipStream_t s) {
    if (need <= *have) return BRIE_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
    int rc = alloc_f32(p, need, s);
    if (rc == BRIE_OK) *have = need;
    return rc;
}

void configure_tiling(brie_handle *h) {
    const int64_t Nc = h->p.Nc;
    h->gene_blocks = static_cast<int>((h->p.Ng + brie::kGenesPerBlock - 1) / brie::kGenesPerBlock);
    int rpc = h->user_rows_per_chunk;
    if (rpc <= 0) {
        // A function of Nc ONLY (never of the shard's gene count): the per-gene fp32 partial sums are
        // then formed in the same order however the genes are sharded, so a gene's trajectory is
        // bit-identical in a 1-GPU fit and in any gene shard.  Aim for >= 128 cell chunks (measured best:
        // 256 rows at Nc = 50k, 64 rows at Nc = 10k; profiles/history/r01_rows_per_chunk.log).
        rpc = 256;
        while (rpc > 16 && (Nc + rpc - 1) / rpc < 128) rpc /= 2;
        // The streaming kernel runs one workgroup per CU, so a launch is gene_blocks x n_chunks / 256 rounds of workgroups
        // and the last, partly filled round is lost time: with n_chunks = 256 (or 128) every round is full (or half full)
        // WHATEVER the shard's gene count.  Re-swept with the round-4 kernel (profiles/r4at_rpc_*.json): configs[1] 0.4407 ms
        // at 64 rows (12.3 rounds) -> 0.4304 at 79 (127 chunks, 9.9 rounds); configs[2] 8.170 -> 8.167 at 196 (256 chunks);
        // an 8-way shard of it 1.0834 -> 1.0744.  The MFMA tile kernel works in 32-row tiles and keeps the powers of two.
        if (!h->wide_like && Nc >= 6144) {
            int64_t n = Nc >= 32768 ? 256 : 128;
            while ((Nc + n - 1) / n > 400) n *= 2;
            rpc = static_cast<int>((Nc + n - 1) / n);
        }
    }
    h->rows_per_chunk = rpc;
    h->n_chunks = static_cast<int>((Nc + rpc - 1) / rpc);
    h->fin_blocks = static_cast<int>((h->p.Ng + brie::kBlock - 1) / brie::kBlock);
}

int ensure_partials(brie_handle *h) {
    const size_t need = static_cast<size_t>(h->n_chunks) * h->S * h->ld;
    if (need > h->partials_elems) {
        if (h->partials) HIP_TRY(hipFree(h->partials));
        h->partials = nullptr;
        HIP_TRY(dev_alloc(&h->partials, need * sizeof(float)));
        h->partials_elems = need;
    }
    return BRIE_OK;
}

int check_ready(const brie_handle *h) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    for (int l = 0; l < h->p.n_layers; ++l)
        if (!h->have_c[l]) return fail(BRIE_ERR_STATE, "count layer %d not uploaded", l + 1);
    if (h->p.Kc > 0 && !h->have_xc) return fail(BRIE_ERR_STATE, "Xc not uploaded (Kc=%d)", h->p.Kc);
    if (h->p.has_efflen && !h->have_eff) return fail(BRIE_ERR_STATE, "effLen not uploaded");
    if (h->p.Kg > 0 && !h->have_xg) return fail(BRIE_ERR_STATE, "Xg not uploaded (Kg=%d)", h->p.Kg);
    if (!h->have_state)
        return fail(BRIE_ERR_STATE, "state not initialised: call brie_init_state or upload Z_loc..sigma_log");
    if (h->persist_flag_host && *h->persist_flag_host != 0)
        return fail(BRIE_ERR_HIP, "a many-steps-per-launch step kernel gave up waiting for its workgroups (the device was not "
                                  "free to hold the grid): the handle's state is undefined; BRIE_FUSE_STEPS=0 keeps two launches per step");
    return BRIE_OK;
}

int launch_step_persist(const brie_handle *h, const brie::LaunchCfg &c, const brie::StepPointers &q, const brie::StepScalars &a,
                        const brie::PersistArgs *dev_args, int n_cus) {
    switch (h->kernel_kc) {
#define BRIE_CASE(N) case N: return brie::launch_step_persist_kc##N(c, q, a, dev_args, n_cus);
        BRIE_CASE(0) BRIE_CASE(1) BRIE_CASE(2) BRIE_CASE(3) BRIE_CASE(4) BRIE_CASE(5) BRIE_CASE(6) BRIE_CASE(7)
#undef BRIE_CASE
        default: return brie::launch_step_persist_kc8(c, q, a, dev_args, n_cus);
    }
}

void launch_step(const brie_handle *h, const brie::LaunchCfg &c, const brie::StepPointers &q,
                 const brie::StepScalars &a, const brie::CoupledArgs &cp) {
    if (h->wide_like) { brie::launch_step_wide(c, q, a, cp); return; }
    switch (h->p.Kc) {
#define BRIE_CASE(N) case N: brie::launch_step_kc##N(c, q, a, cp); break;
        BRIE_CASE(0) BRIE_CASE(1) BRIE_CASE(2) BRIE_CASE(3) BRIE_CASE(4) BRIE_CASE(5) BRIE_CASE(6) BRIE_CASE(7)
        default: brie::launch_step_kc8(c, q, a, cp); break;
#undef BRIE_CASE
    }
}
void launch_tile(const brie_handle *h, const brie::LaunchCfg &c, const brie::StepPointers &q, const brie::StepScalars &a,
                 const brie::TileArgs &t) {
    switch (h->mode) {
        case brie::kLik2: brie::launch_tile_mode0(c, q, a, t, h->tile_nacc, h->tile_njt, h->tile_nw, h->tile_kcr, h->tile_lds); break;
        case brie::kLikEff2: brie::launch_tile_mode1(c, q, a, t, h->tile_nacc, h->tile_njt, h->tile_nw, h->tile_kcr, h->tile_lds); break;
        default: brie::launch_tile_mode2(c, q, a, t, h->tile_nacc, h->tile_njt, h->tile_nw, h->tile_kcr, h->tile_lds); break;
    }
}
void launch_margin(const brie_handle *h, const brie::LaunchCfg &c, const brie::StepPointers &q,
                   const brie::StepScalars &a) {
    switch (h->kernel_kc) {
#define BRIE_CASE(N) case N: brie::launch_margin_kc##N(c, q, a); break;
        BRIE_CASE(0) BRIE_CASE(1) BRIE_CASE(2) BRIE_CASE(3) BRIE_CASE(4) BRIE_CASE(5) BRIE_CASE(6) BRIE_CASE(7)
        default: brie::launch_margin_kc8(c, q, a); break;
#undef BRIE_CASE
    }
}
void launch_loss_gene(const brie_handle *h, const brie::LaunchCfg &c, const brie::LossGeneArgs &a) {
    switch (h->kernel_kc) {
#define BRIE_CASE(N) case N: brie::launch_loss_gene_kc##N(c, a); break;
        BRIE_CASE(0) BRIE_CASE(1) BRIE_CASE(2) BRIE_CASE(3) BRIE_CASE(4) BRIE_CASE(5) BRIE_CASE(6) BRIE_CASE(7)
        default: brie::launch_loss_gene_kc8(c, a); break;
#undef BRIE_CASE
    }
}

int grid_1d(int64_t n);

// Try to switch the count layers to a compact storage: u8 if every value is an integer in [0,255],
// u16 if in [0,65535], else stay fp32.  Called once, when the counts are final (at the pseudo-count,
// or at the first step if none is added).
int free_tier_tables(brie_handle *h) {
    if (h->tt.q_esz) HIP_TRY(hipFree(const_cast<uint8_t *>(h->tt.q_esz)));
    if (h->tt.q_off) HIP_TRY(hipFree(const_cast<int32_t *>(h->tt.q_off)));
    if (h->tt.row_bytes) HIP_TRY(hipFree(const_cast<int32_t *>(h->tt.row_bytes)));
    if (h->tt.blk_base) HIP_TRY(hipFree(const_cast<int64_t *>(h->tt.blk_base)));
    h->tt = brie::TierTables{nullptr, nullptr, nullptr, nullptr};
    h->q_esz_host.clear();
    h->tier_layer_bytes = 0;
    return BRIE_OK;
}

size_t compact_layer_bytes(const brie_handle *h, int cs) {
    const size_t tile = static_cast<size_t>(h->p.Nc) * brie::kGenesPerBlock;
    if (cs == brie::kCountU8) return tile * h->gene_blocks;
    if (cs == brie::kCountU16) return tile * h->gene_blocks * 2;
    return h->tier_layer_bytes;
}

// device tables of the per-quad tiers from q_esz_host
int build_tier_tables(brie_handle *h) {
    const int nq = h->gene_blocks * brie::kWave;
    std::vector<int32_t> q_off(static_cast<size_t>(nq)), row_bytes(static_cast<size_t>(h->gene_blocks));
    std::vector<int64_t> blk_base(static_cast<size_t>(h->gene_blocks));
    int64_t base = 0;
    for (int g = 0; g < h->gene_blocks; ++g) {
        int32_t off = 0;
        for (int q = g * brie::kWave; q < (g + 1) * brie::kWave; ++q) { q_off[q] = off; off += 4 * h->q_esz_host[q]; }
        row_bytes[g] = off;
        blk_base[g] = base;
        base += static_cast<int64_t>(off) * h->p.Nc;
    }
    h->tier_layer_bytes = static_cast<size_t>(base);
    // (each table is entered into the handle as soon as it exists, so a failure further down leaves nothing behind that
    // free_tier_tables / brie_destroy would not release)
    uint8_t *d_esz = nullptr; int32_t *d_off = nullptr, *d_row = nullptr; int64_t *d_base = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_esz), nq));
    h->tt.q_esz = d_esz;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_off), nq * sizeof(int32_t)));
    h->tt.q_off = d_off;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_row), h->gene_blocks * sizeof(int32_t)));
    h->tt.row_bytes = d_row;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_base), h->gene_blocks * sizeof(int64_t)));
    h->tt.blk_base = d_base;
    HIP_TRY(hipMemcpyAsync(d_esz, h->q_esz_host.data(), nq, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_off, q_off.data(), nq * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_row, row_bytes.data(), h->gene_blocks * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_base, blk_base.data(), h->gene_blocks * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // the host vectors go out of scope
    return BRIE_OK;
}

// the streamed arrays were replaced (counts expanded / re-tiered): what the search measured no longer describes them
void placement_invalidate(brie_handle *h) {
    h->placement_done = false;
    h->placement_tries = 0; h->placement_kept = 0; h->placement_status = BRIE_PLACEMENT_NOT_RUN;
    for (double &g : h->placement_gbs) g = 0.0;
    h->placement_note[0] = '\0';
}

int try_compact_counts(brie_handle *h) {
    if (h->compact_tried) return BRIE_OK;
    h->compact_tried = true;
    const int64_t n = h->p.Nc * h->ld, n4 = n / 4;
    // one flag word per gene quad
    const int n_flags = h->gene_blocks * brie::kWave;
    int *flag = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&flag), n_flags * sizeof(int)));
    HIP_TRY(hipMemsetAsync(flag, 0, n_flags * sizeof(int), h->stream));
    for (int l = 0; l < h->p.n_layers; ++l)
        hipLaunchKernelGGL(brie::count_range_check, dim3(grid_1d(n4)), dim3(256), 0, h->stream, h->c[l], n4,
                           static_cast<int>(h->p.Nc), flag);
    std::vector<int> bits_q(static_cast<size_t>(n_flags), 1);
    hipError_t e = hipMemcpyAsync(bits_q.data(), flag, n_flags * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(flag);
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "count range check: %s", hipGetErrorString(e));
    int bits = 0;
    for (int b : bits_q) bits |= b;
    if (bits & 4) {                               // the reference would silently produce NaN posteriors
        h->compact_tried = false;                 // checked again (and refused again) on the next attempt
        return fail(BRIE_ERR_INVALID, "count layers contain negative or non-finite values");
    }
    if ((bits & 1) || !h->allow_compact) return BRIE_OK;     // fractional / huge: stay fp32
    int n16 = 0;
    for (int b : bits_q) n16 += (b & 2) ? 1 : 0;
    const char *tm = getenv("BRIE_COUNT_TIERS");  // "uniform": one tier for the whole shard (A/B runs)
    const int n_real = static_cast<int>((h->p.Ng + brie::kVec - 1) / brie::kVec);      // the padding quads hold zeros
    const bool uniform = n16 == 0 || n16 == n_real || (tm && strcmp(tm, "uniform") == 0);
    const int cs = uniform ? ((bits & 2) ? brie::kCountU16 : brie::kCountU8) : brie::kCountMixed;
    if (cs == brie::kCountMixed) {
        h->q_esz_host.resize(static_cast<size_t>(n_flags));
        for (int q = 0; q < n_flags; ++q) h->q_esz_host[q] = (bits_q[q] & 2) ? 2 : 1;
        int rc = build_tier_tables(h);
        if (rc != BRIE_OK) return rc;
    }
    const size_t bytes = compact_layer_bytes(h, cs);
    for (int l = 0; l < h->p.n_layers; ++l) {
        HIP_TRY(dev_alloc(&h->cu[l], bytes));
        if (cs == brie::kCountMixed)
            hipLaunchKernelGGL(brie::count_compact_mixed, dim3(grid_1d(n4)), dim3(256), 0, h->stream, h->c[l], h->cu[l], n4,
                               static_cast<int>(h->p.Nc), h->tt);
        else
            hipLaunchKernelGGL(brie::count_compact, dim3(grid_1d(n4)), dim3(256), 0, h->stream, h->c[l], h->cu[l], n4, cs);
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    // the fp32 layers have done their job.  They are FREED, not handed to the block cache: the handle lives on, and
    // what compaction saves (2 - 3 arrays of Nc x Ng floats) is memory the caller may be counting on
    for (int l = 0; l < h->p.n_layers; ++l) { HIP_TRY(hipFree(h->c[l])); h->c[l] = nullptr; }
    h->cs = cs;
    placement_invalidate(h);                      // fresh count arrays (a re-upload after a search): the next step probes again
    return BRIE_OK;
}

// compact layer l -> fp32 tile layout in `dst`, with (apply_pc) or without the in-register pseudo-count rule
void launch_expand(brie_handle *h, int l, float *dst, float pc, int apply_pc) {
    const int64_t n4 = h->p.Nc * h->ld / 4;
    if (h->cs == brie::kCountMixed)
        hipLaunchKernelGGL(brie::count_expand_mixed, dim3(grid_1d(n4)), dim3(256), 0, h->stream, h->cu[0], h->cu[1], h->cu[l],
                           dst, n4, static_cast<int>(h->p.Nc), h->tt, pc, apply_pc);
    else
        hipLaunchKernelGGL(brie::count_expand, dim3(grid_1d(n4)), dim3(256), 0, h->stream, h->cu[0], h->cu[1], h->cu[l], dst,
                           n4, pc, apply_pc, h->cs);
}

// Mixed tiers -> one u16 tier for the whole shard (before the gene quads are permuted: the tier tables describe the
// unpermuted order).  Values are unchanged.
int retier_uniform_u16(brie_handle *h) {
    if (h->cs != brie::kCountMixed) return BRIE_OK;
    const int64_t n = h->p.Nc * h->ld, n4 = n / 4;
    float *tmp = nullptr;
    HIP_TRY(dev_alloc(&tmp, static_cast<size_t>(n) * sizeof(float)));
    void *fresh[3] = {nullptr, nullptr, nullptr};
    for (int l = 0; l < h->p.n_layers; ++l) {
        HIP_TRY(dev_alloc(&fresh[l], static_cast<size_t>(n) * 2));
        launch_expand(h, l, tmp, 0.0f, 0);
        hipLaunchKernelGGL(brie::count_compact, dim3(grid_1d(n4)), dim3(256), 0, h->stream, tmp, fresh[l], n4,
                           static_cast<int>(brie::kCountU16));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(tmp));
    for (int l = 0; l < h->p.n_layers; ++l) { HIP_TRY(hipFree(h->cu[l])); h->cu[l] = fresh[l]; }
    h->cs = brie::kCountU16;
    placement_invalidate(h);                      // fresh count arrays: the next step probes again
    return free_tier_tables(h);
}

// Back to fp32 layers (pseudo-count materialised) -- used when counts are modified again.
int expand_counts(brie_handle *h) {
    if (h->cs == brie::kCountF32) return BRIE_OK;
    const int64_t n = h->p.Nc * h->ld;
    for (int l = 0; l < h->p.n_layers; ++l) {
        {
            int rc_a = alloc_mat(&h->c[l], static_cast<size_t>(n), h->p.device, h->stream);
            if (rc_a != BRIE_OK) return rc_a;
        }
        launch_expand(h, l, h->c[l], h->pc, l < 2 ? 1 : 0);
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int l = 0; l < h->p.n_layers; ++l) { HIP_TRY(hipFree(h->cu[l])); h->cu[l] = nullptr; }
    int rc_t = free_tier_tables(h);
    if (rc_t != BRIE_OK) return rc_t;
    h->cs = brie::kCountF32;
    h->pc = 0.0f;
    h->allow_compact = false;                     // values are no longer integers
    placement_invalidate(h);                      // fresh count arrays: the next step probes again
    return BRIE_OK;
}

// ---- packing of active quads ------------------------------------------------------------------
// apply_quad_gather: every gene-indexed array of the shard is re-ordered so that position p holds what
// was at position from[p].  Matrices go through one scratch buffer (pointer swap), vectors likewise.
int apply_quad_gather(brie_handle *h, const std::vector<int32_t> &from) {
    {
        int rc = retier_uniform_u16(h);
        if (rc != BRIE_OK) return rc;
    }
    const int nq = static_cast<int>(h->ld / 4);
    const int Nc = static_cast<int>(h->p.Nc);
    int32_t *d_from = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_from), nq * sizeof(int32_t)));
    hipError_t e = hipMemcpyAsync(d_from, from.data(), nq * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
    // two scratch buffers: one cell x gene matrix (it swaps roles with the live arrays, so it must have
    // exactly their size) and one for the largest per-gene row set (the loss rings / Wc_loc)
    const size_t mat_bytes = static_cast<size_t>(Nc) * h->ld * sizeof(float);
    const size_t row_bytes = static_cast<size_t>(brie::kLossRing > h->p.Kc ? brie::kLossRing : h->p.Kc) * h->ld * sizeof(float);
    if (e == hipSuccess && !h->pack_scratch) {
        e = dev_alloc(&h->pack_scratch, mat_bytes);
        if (e == hipSuccess) h->pack_scratch_bytes = mat_bytes;
    }
    if (e == hipSuccess && !h->row_scratch) e = hipMalloc(reinterpret_cast<void **>(&h->row_scratch), row_bytes);
    if (e != hipSuccess) { hipFree(d_from); return fail(BRIE_ERR_HIP, "pack: %s", hipGetErrorString(e)); }
    const dim3 grid(grid_1d(static_cast<int64_t>(nq) * Nc)), block(256);
    // cell x gene matrices: the live pointer and the scratch pointer swap roles after each gather
    auto move_f32 = [&](float *&arr) {
        if (!arr) return;
        hipLaunchKernelGGL(brie::gather_quads_tiled<float4>, grid, block, 0, h->stream,
                           reinterpret_cast<const float4 *>(arr), reinterpret_cast<float4 *>(h->pack_scratch), d_from, nq, Nc);
        float *old = arr;
        arr = static_cast<float *>(h->pack_scratch);
        h->pack_scratch = old;
    };
    for (int l = 0; l < h->p.n_layers; ++l) {
        if (h->cs == brie::kCountF32) { move_f32(h->c[l]); continue; }
        // compact layers are smaller than the scratch: gather into it, then copy back in place
        const size_t bytes = static_cast<size_t>(Nc) * h->ld * (h->cs == brie::kCountU16 ? 2 : 1);
        if (h->cs == brie::kCountU16)
            hipLaunchKernelGGL(brie::gather_quads_tiled<uint2>, grid, block, 0, h->stream,
                               static_cast<const uint2 *>(h->cu[l]), static_cast<uint2 *>(h->pack_scratch), d_from, nq, Nc);
        else
            hipLaunchKernelGGL(brie::gather_quads_tiled<uint32_t>, grid, block, 0, h->stream,
                               static_cast<const uint32_t *>(h->cu[l]), static_cast<uint32_t *>(h->pack_scratch), d_from,
                               nq, Nc);
        e = hipMemcpyAsync(h->cu[l], h->pack_scratch, bytes, hipMemcpyDeviceToDevice, h->stream);
        if (e != hipSuccess) break;
    }
    if (e == hipSuccess) {
        move_f32(h->mu); move_f32(h->rho); move_f32(h->m_mu); move_f32(h->v_mu); move_f32(h->m_rho); move_f32(h->v_rho);
        // per-gene vectors (rows, ld): gather into the scratch, copy back
        auto move_rows = [&](float *arr, int rows) {
            if (!arr || rows == 0 || e != hipSuccess) return;
            hipLaunchKernelGGL(brie::gather_quads_rows, dim3(grid_1d(static_cast<int64_t>(rows) * nq)), block, 0, h->stream,
                               arr, h->row_scratch, d_from, nq, rows, h->ld);
            e = hipMemcpyAsync(arr, h->row_scratch, static_cast<size_t>(rows) * h->ld * sizeof(float),
                               hipMemcpyDeviceToDevice, h->stream);
        };
        move_rows(h->W, h->p.Kc); move_rows(h->m_W, h->p.Kc); move_rows(h->v_W, h->p.Kc);
        move_rows(h->b, 1); move_rows(h->m_b, 1); move_rows(h->v_b, 1);
        move_rows(h->lam, 1); move_rows(h->m_lam, 1); move_rows(h->v_lam, 1);
        move_rows(h->effL, 6); move_rows(h->gene_active, 1);
        move_rows(h->ring_kl, brie::kLossRing); move_rows(h->ring_ll, brie::kLossRing);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_from);
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "pack: %s", hipGetErrorString(e));
    return BRIE_OK;
}

int upload_quad_ids_and_blocks(brie_handle *h, const std::vector<float> &mask_by_position) {
    std::vector<int32_t> blocks(static_cast<size_t>(h->gene_blocks), 0);
    for (int64_t j = 0; j < h->ld; ++j)
        if (mask_by_position[j] != 0.0f) blocks[j / brie::kGenesPerBlock] = 1;
    HIP_TRY(hipMemcpyAsync(h->quad_ids, h->perm.data(), h->perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->block_active, blocks.data(), blocks.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BRIE_OK;
}

// Back to the identity order (before anything that addresses genes by their original index).
int ensure_identity(brie_handle *h) {
    {
        int rc = io_wait(h);
        if (rc != BRIE_OK) return rc;
    }
    if (!h->packed) return BRIE_OK;
    const int nq = static_cast<int>(h->ld / 4);
    std::vector<int32_t> inv(nq);
    for (int p = 0; p < nq; ++p) inv[h->perm[p]] = p;          // original quad q currently lives at inv[q]
    int rc = apply_quad_gather(h, inv);
    if (rc != BRIE_OK) return rc;
    for (int p = 0; p < nq; ++p) h->perm[p] = p;
    h->packed = false;
    return upload_quad_ids_and_blocks(h, h->mask_host);     // identity quad ids, block flags of the unpacked mask
}

// matrix-sized staging buffer for read-backs (kept: BRIE_RV reads four matrices in a row)
int io_buffer(brie_handle *h, float **out) {
    if (!h->io_scratch)
        HIP_TRY(dev_alloc(&h->io_scratch, static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float)));
    *out = h->io_scratch;
    return BRIE_OK;
}

int matrix_target(brie_handle *h, int which, float **dev, int64_t *rows, int64_t *cols, int64_t *ldd) {
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    switch (which) {
        case BRIE_COUNT1: case BRIE_COUNT2: case BRIE_COUNT3:
            if (which - BRIE_COUNT1 >= h->p.n_layers)
                return fail(BRIE_ERR_INVALID, "count layer %d outside n_layers=%d", which + 1, h->p.n_layers);
            *dev = h->c[which - BRIE_COUNT1]; *rows = Nc; *cols = Ng; *ldd = h->ld; return BRIE_OK;
        case BRIE_Z_LOC: *dev = h->mu; *rows = Nc; *cols = Ng; *ldd = h->ld; return BRIE_OK;
        case BRIE_Z_STD_LOG: *dev = h->rho; *rows = Nc; *cols = Ng; *ldd = h->ld; return BRIE_OK;
        case BRIE_XC: *dev = h->Xc; *rows = Nc; *cols = h->p.Kc; *ldd = h->p.Kc; return BRIE_OK;
        case BRIE_WC_LOC: *dev = h->W; *rows = h->p.Kc; *cols = Ng; *ldd = h->ld; return BRIE_OK;
        case BRIE_INTERCEPT:
            if (h->cell_mode) { *dev = h->cb; *rows = Nc; *cols = 1; *ldd = 1; return BRIE_OK; }
            *dev = h->b; *rows = 1; *cols = Ng; *ldd = h->ld; return BRIE_OK;
        case BRIE_SIGMA_LOG:
            if (h->cell_mode) { *dev = h->clam; *rows = Nc; *cols = 1; *ldd = 1; return BRIE_OK; }
            *dev = h->lam; *rows = 1; *cols = Ng; *ldd = h->ld; return BRIE_OK;
        case BRIE_WG_LOC:
            if (h->p.Kg == 0) { *dev = nullptr; *rows = Nc; *cols = 0; *ldd = 1; return BRIE_OK; }
            *dev = h->Wg; *rows = Nc; *cols = h->p.Kg; *ldd = h->kgp; return BRIE_OK;
        default: return fail(BRIE_ERR_INVALID, "array id %d is not a stored matrix", which);
    }
}

// Copy a (Nc, Ng) matrix between the caller's row-major buffer and the device layout.
// One strided 2-D copy per 256-gene block (1-KiB device rows).
int copy_cellgene(brie_handle *h, float *dev, const float *ext, float *ext_out, int64_t ld_ext) {
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    const int64_t G = brie::kGenesPerBlock;
    for (int64_t g = 0; g * G < Ng; ++g) {
        const int64_t cols = (Ng - g * G) < G ? (Ng - g * G) : G;
        float *d = dev + g * h->gb_stride;
        if (ext_out)
            HIP_TRY(hipMemcpy2DAsync(ext_out + g * G, ld_ext * sizeof(float), d, G * sizeof(float), cols * sizeof(float),
                                     Nc, hipMemcpyDefault, h->stream));
        else
            HIP_TRY(hipMemcpy2DAsync(d, G * sizeof(float), ext + g * G, ld_ext * sizeof(float), cols * sizeof(float), Nc,
                                     hipMemcpyDefault, h->stream));
    }
    return BRIE_OK;
}

// ---- staged host ingest of a count layer (replaces the densify + cast of model_wrap.py:108-111) ----------
// The API hands over (Nc, Ng) fp32 layers in PAGEABLE host memory: 8 GB at configs[2].  Left to the runtime, such a
// copy is pinned or staged page by page by ONE runtime thread and its speed is the box's (0.17 s on one, 0.69 s on the
// next).  Here T host threads each own a lane -- two page-locked slabs, two device slabs, a stream -- and walk their
// share of the row slabs: convert the slab to u16 while checking that it holds nothing but integers in [0, 65535]
// (else the slab travels as the fp32 values themselves), start its asynchronous copy, enqueue the kernel that writes
// it into the tiled fp32 layer, and go on converting the next slab while the copy engine works.  Half the bytes cross
// PCIe, the host side runs at the memory bandwidth of T cores, and what reaches the layer is bit-identical.
struct IngestLane {
    void *pin[2] = {nullptr, nullptr};
    void *dev[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
};
struct IngestPool {
    std::mutex mu;                      // one staged upload at a time per process
    int device = -1;
    size_t slab_bytes = 0;
    std::vector<IngestLane> lanes;
    void release() {
        for (IngestLane &ln : lanes) {
            for (int b = 0; b < 2; ++b) {
                if (ln.pin[b]) hipHostFree(ln.pin[b]);
                if (ln.dev[b]) hipFree(ln.dev[b]);
                if (ln.ev[b]) hipEventDestroy(ln.ev[b]);
            }
            if (ln.stream) hipStreamDestroy(ln.stream);
        }
        lanes.clear();
        device = -1;
        slab_bytes = 0;
    }
};
IngestPool g_ingest;

int ingest_threads() {
    const char *e = getenv("BRIE_INGEST_THREADS");
    if (e && atoi(e) > 0) return std::min(64, atoi(e));
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = static_cast<int>(std::thread::hardware_concurrency());
    return std::max(1, std::min(8, n));
}

// (host code only: the device pass of this translation unit knows no x86 function multiversioning)
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__)
#define BRIE_HOST_SIMD_CLONES __attribute__((noinline))
#else
#define BRIE_HOST_SIMD_CLONES __attribute__((target_clones("arch=x86-64-v4", "arch=x86-64-v3", "default")))
#endif
// rows x cols floats (row pitch ld) -> u16; returns nonzero when some value is not a non-negative integer <= 65535
// (negative zero counts as "not": its sign bit would be lost)
BRIE_HOST_SIMD_CLONES uint32_t convert_rows_u16(const float *src, int64_t ld, int64_t rows, int64_t cols, uint16_t *dst) {
    uint32_t bad = 0;
    for (int64_t r = 0; r < rows; ++r) {
        const float *s = src + r * ld;
        uint16_t *d = dst + r * cols;
        uint32_t b = 0;
#pragma clang loop vectorize(enable) interleave(enable)
        for (int64_t j = 0; j < cols; ++j) {
            const float v = s[j];
            uint32_t bits;
            memcpy(&bits, &v, sizeof(bits));
            const int32_t iv = static_cast<int32_t>(v < 65536.0f ? (v > -1.0f ? v : -1.0f) : 65536.0f);   // NaN -> 65536
            b |= static_cast<uint32_t>(static_cast<float>(iv) != v) | (bits >> 31) | static_cast<uint32_t>(iv > 65535);
            d[j] = static_cast<uint16_t>(iv);
        }
        bad |= b;
    }
    return bad;
}
// The other element types a caller may hold its count layers in (the reference casts whatever it is given with
// .astype(np.float32), io_utils.py:18 / model_wrap.py:111): the same two conversions per type -- exact u16 when every
// value of the slab is an integer in [0, 65535], else the float32 cast the reference would have made.
BRIE_HOST_SIMD_CLONES uint32_t convert_rows_u16_f64(const double *src, int64_t ld, int64_t rows, int64_t cols, uint16_t *dst) {
    uint32_t bad = 0;
    for (int64_t r = 0; r < rows; ++r) {
        const double *s = src + r * ld;
        uint16_t *d = dst + r * cols;
        uint32_t b = 0;
#pragma clang loop vectorize(enable) interleave(enable)
        for (int64_t j = 0; j < cols; ++j) {
            const double v = s[j];
            uint64_t bits;
            memcpy(&bits, &v, sizeof(bits));
            const int32_t iv = static_cast<int32_t>(v < 65536.0 ? (v > -1.0 ? v : -1.0) : 65536.0);
            b |= static_cast<uint32_t>(static_cast<double>(iv) != v) | static_cast<uint32_t>(bits >> 63) | static_cast<uint32_t>(iv > 65535);
            d[j] = static_cast<uint16_t>(iv);
        }
        bad |= b;
    }
    return bad;
}
#define BRIE_CONVERT_INT(NAME, T)                                                                                          \
    BRIE_HOST_SIMD_CLONES uint32_t NAME(const T *src, int64_t ld, int64_t rows, int64_t cols, uint16_t *dst) {            \
        uint32_t bad = 0;                                                                                                  \
        for (int64_t r = 0; r < rows; ++r) {                                                                               \
            const T *s = src + r * ld;                                                                                     \
            uint16_t *d = dst + r * cols;                                                                                  \
            uint32_t b = 0;                                                                                                \
            _Pragma("clang loop vectorize(enable) interleave(enable)") for (int64_t j = 0; j < cols; ++j) {                \
                const T v = s[j];                                                                                          \
                b |= static_cast<uint32_t>(v < static_cast<T>(0)) | static_cast<uint32_t>(static_cast<uint64_t>(v) > 65535u); \
                d[j] = static_cast<uint16_t>(v);                                                                           \
            }                                                                                                              \
            bad |= b;                                                                                                      \
        }                                                                                                                  \
        return bad;                                                                                                        \
    }
BRIE_CONVERT_INT(convert_rows_u16_i32, int32_t)
BRIE_CONVERT_INT(convert_rows_u16_i64, int64_t)
BRIE_CONVERT_INT(convert_rows_u16_u32, uint32_t)
BRIE_CONVERT_INT(convert_rows_u16_i16, int16_t)
BRIE_CONVERT_INT(convert_rows_u16_u16, uint16_t)
BRIE_CONVERT_INT(convert_rows_u16_u8, uint8_t)
#undef BRIE_CONVERT_INT
template <typename T>
void cast_rows_f32(const T *src, int64_t ld, int64_t rows, int64_t cols, float *dst) {
    for (int64_t r = 0; r < rows; ++r) {
        const T *s = src + r * ld;
        float *d = dst + r * cols;
        for (int64_t j = 0; j < cols; ++j) d[j] = static_cast<float>(s[j]);       // round to nearest, as astype(float32)
    }
}
size_t dtype_size(int32_t dtype) {
    switch (dtype) {
        case BRIE_DT_F32: case BRIE_DT_I32: case BRIE_DT_U32: return 4;
        case BRIE_DT_F64: case BRIE_DT_I64: return 8;
        case BRIE_DT_I16: case BRIE_DT_U16: return 2;
        case BRIE_DT_U8: return 1;
        default: return 0;
    }
}
// rows of a typed source -> `pin`: u16 (returns 0) or, when the slab holds anything else, float32 (returns 1)
int convert_slab(const void *src, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, void *pin) {
    uint16_t *d16 = static_cast<uint16_t *>(pin);
    float *d32 = static_cast<float *>(pin);
#define BRIE_SLAB(T, FN)                                                                 \
    {                                                                                    \
        const T *s = static_cast<const T *>(src);                                        \
        if (FN(s, ld, rows, cols, d16) == 0) return 0;                                   \
        cast_rows_f32<T>(s, ld, rows, cols, d32);                                        \
        return 1;                                                                        \
    }
    switch (dtype) {
        case BRIE_DT_F64: BRIE_SLAB(double, convert_rows_u16_f64)
        case BRIE_DT_I32: BRIE_SLAB(int32_t, convert_rows_u16_i32)
        case BRIE_DT_I64: BRIE_SLAB(int64_t, convert_rows_u16_i64)
        case BRIE_DT_U32: BRIE_SLAB(uint32_t, convert_rows_u16_u32)
        case BRIE_DT_I16: BRIE_SLAB(int16_t, convert_rows_u16_i16)
        case BRIE_DT_U16: BRIE_SLAB(uint16_t, convert_rows_u16_u16)
        case BRIE_DT_U8: BRIE_SLAB(uint8_t, convert_rows_u16_u8)
        default: {
            const float *s = static_cast<const float *>(src);
            if (convert_rows_u16(s, ld, rows, cols, d16) == 0) return 0;
            for (int64_t r = 0; r < rows; ++r)                       // fractional / huge / negative: the values themselves
                memcpy(d32 + r * cols, s + r * ld, static_cast<size_t>(cols) * sizeof(float));
            return 1;
        }
    }
#undef BRIE_SLAB
}

bool use_staged_ingest(const brie_handle *h, const void *src, int64_t elems) {
    const char *m = getenv("BRIE_INGEST");               // "direct" / "staged": force (A/B runs, tests)
    if (m && strcmp(m, "direct") == 0) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess) {
        if (at.type != hipMemoryTypeUnregistered) return false;      // device, managed or page-locked host memory: DMA as is
    } else {
        (void)hipGetLastError();                                      // plain pageable host memory
    }
    if (m && strcmp(m, "staged") == 0) return true;
    const char *me = getenv("BRIE_INGEST_MIN_ELEMS");
    const int64_t min_elems = me ? atoll(me) : (int64_t(1) << 24);   // below 64 MB the plain copy is as good
    (void)h;
    return elems >= min_elems;
}

// the lanes of the staged host <-> device pipelines (ingest and result export share them); caller holds g_ingest.mu
int ensure_lane_pool(int device, size_t slab_bytes, int T) {
    if (g_ingest.device == device && g_ingest.slab_bytes >= slab_bytes && static_cast<int>(g_ingest.lanes.size()) >= T)
        return BRIE_OK;
    g_ingest.release();
    g_ingest.lanes.resize(static_cast<size_t>(T));
    g_ingest.device = device;
    g_ingest.slab_bytes = slab_bytes;
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (IngestLane &ln : g_ingest.lanes) {
        // highest priority: the short kernels of a slab must get compute units as they free up, also next to a long
        // kernel on the handle's stream (the result export runs under the 500-draw loss_gene pass)
        hipError_t e = hipStreamCreateWithPriority(&ln.stream, hipStreamNonBlocking, prio_hi);
        for (int b = 0; b < 2 && e == hipSuccess; ++b) {
            e = hipHostMalloc(&ln.pin[b], slab_bytes, hipHostMallocDefault);
            if (e == hipSuccess) e = dev_alloc(&ln.dev[b], slab_bytes);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.ev[b], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            g_ingest.release();
            return fail(BRIE_ERR_HIP, "staging lanes: %s", hipGetErrorString(e));
        }
    }
    return BRIE_OK;
}

int staged_count_upload(brie_handle *h, float *dev, const void *src_any, int32_t dtype, int64_t ld) {
    const char *src = static_cast<const char *>(src_any);
    const size_t esz = dtype_size(dtype);
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    std::lock_guard<std::mutex> lock(g_ingest.mu);
    const char *se = getenv("BRIE_INGEST_SLAB_ELEMS");
    const int64_t slab_elems = std::max<int64_t>(Ng, se && atoll(se) > 0 ? atoll(se) : (int64_t(1) << 21));   // 8 MB of fp32
    const int64_t R = std::max<int64_t>(1, std::min<int64_t>(Nc, slab_elems / Ng));      // (small layers: small slabs, small pool)
    const int64_t n_slabs = (Nc + R - 1) / R;
    const int T = static_cast<int>(std::min<int64_t>(ingest_threads(), n_slabs));
    const size_t slab_bytes = static_cast<size_t>(R) * Ng * sizeof(float);
    {
        int rc_pool = ensure_lane_pool(h->p.device, slab_bytes, T);
        if (rc_pool != BRIE_OK) return rc_pool;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));             // earlier work on the layer (memset, a previous upload)
    std::atomic<int> err{0};
    const int device = h->p.device, gene_blocks = h->gene_blocks;
    const int64_t row_stride = h->row_stride, gb_stride = h->gb_stride;
    auto lane_fn = [&, device, gene_blocks, row_stride, gb_stride](int t) {
        IngestLane &ln = g_ingest.lanes[static_cast<size_t>(t)];
        hipError_t e = hipSetDevice(device);
        int64_t k = 0;
        for (int64_t s = t; s < n_slabs && e == hipSuccess && err.load() == 0; s += T, ++k) {
            const int b = static_cast<int>(k & 1);
            if (k >= 2) e = hipEventSynchronize(ln.ev[b]);            // the copy out of this page-locked slab is done
            if (e != hipSuccess) break;
            const int64_t r0 = s * R, rows = std::min(R, Nc - r0);
            const int bad = convert_slab(src + static_cast<size_t>(r0) * ld * esz, dtype, ld, rows, Ng, ln.pin[b]);
            const size_t bytes = static_cast<size_t>(rows) * Ng * (bad ? sizeof(float) : sizeof(uint16_t));
            e = hipMemcpyAsync(ln.dev[b], ln.pin[b], bytes, hipMemcpyHostToDevice, ln.stream);
            if (e == hipSuccess) e = hipEventRecord(ln.ev[b], ln.stream);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(brie::ingest_slab, dim3(grid_1d(static_cast<int64_t>(gene_blocks) * rows * brie::kWave)), dim3(256),
                               0, ln.stream, ln.dev[b], bad ? 1 : 0, dev, static_cast<int>(r0), static_cast<int>(rows),
                               static_cast<int>(Ng), gene_blocks, row_stride, gb_stride);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ln.stream);
        if (e != hipSuccess) { (void)hipGetLastError(); err.store(static_cast<int>(e)); }
    };
    std::vector<std::thread> pool;
    std::vector<int> inline_lanes{0};                     // lanes this thread walks itself: its own, and any whose thread
    for (int t = 1; t < T; ++t) {                         // could not be created (no exception may cross the C ABI)
        try { pool.emplace_back(lane_fn, t); } catch (...) { inline_lanes.push_back(t); }
    }
    for (int t : inline_lanes) lane_fn(t);
    for (std::thread &th : pool) th.join();
    (void)hipSetDevice(device);
    if (err.load() != 0)
        return fail(BRIE_ERR_HIP, "staged ingest: %s", hipGetErrorString(static_cast<hipError_t>(err.load())));
    return BRIE_OK;
}

int grid_1d(int64_t n) {
    int64_t g = (n + 255) / 256;
    return static_cast<int>(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

const char *brie_last_error(void) { return g_last_error.c_str(); }
int brie_abi_version(void) { return BRIE_AMD_ABI_VERSION; }

// a creation that failed half way: its arrays are released for real (brie_destroy alone would park them in the cache)
static void drop_failed_create(brie_handle *h) {
    brie_destroy(h);
    g_blocks.trim();
}

int brie_create(const brie_problem *p, brie_handle **out) {
    if (!p || !out) return fail(BRIE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->abi_version != BRIE_AMD_ABI_VERSION)
        return fail(BRIE_ERR_INVALID, "abi_version %d != %d", p->abi_version, BRIE_AMD_ABI_VERSION);
    if (p->Nc <= 0 || p->Ng <= 0 || p->Nc > INT32_MAX || p->Ng > INT32_MAX - 1024)
        return fail(BRIE_ERR_INVALID, "bad shape Nc=%lld Ng=%lld", (long long)p->Nc, (long long)p->Ng);
    if (p->Kc < 0 || p->Kc > BRIE_MAX_KC_PANELS)
        return fail(BRIE_ERR_UNSUPPORTED, "Kc=%d outside 0..%d", p->Kc, BRIE_MAX_KC_PANELS);
    if (p->Kg > BRIE_MAX_KG_PANELS)
        return fail(BRIE_ERR_UNSUPPORTED, "Kg=%d outside 0..%d", p->Kg, BRIE_MAX_KG_PANELS);
    if (p->Kg < 0) return fail(BRIE_ERR_INVALID, "Kg=%d", p->Kg);
    if (p->intercept_mode != 0 && p->intercept_mode != 1)
        return fail(BRIE_ERR_INVALID, "intercept_mode=%d (0 = gene, 1 = cell)", p->intercept_mode);
    if ((p->Kg > 0 || p->intercept_mode == 1) && p->gene_offset != 0 && p->sharded == 0)
        return fail(BRIE_ERR_UNSUPPORTED, "gene features / cell intercepts couple all genes: a gene shard "
                    "(gene_offset=%lld) must be created with sharded=1 and stepped with brie_step_begin/_end",
                    (long long)p->gene_offset);
    if (p->n_layers != 2 && p->n_layers != 3)
        return fail(BRIE_ERR_INVALID, "n_layers=%d (must be 2 or 3)", p->n_layers);
    if (p->n_layers == 3 && !p->has_efflen)
        return fail(BRIE_ERR_INVALID, "a third count layer is only used with effLen (model_TFProb.py:184)");
    if (p->gene_offset < 0 || p->gene_offset % 4 != 0)
        return fail(BRIE_ERR_INVALID, "gene_offset=%lld must be a non-negative multiple of 4", (long long)p->gene_offset);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (p->device < 0 || p->device >= ndev)
        return fail(BRIE_ERR_HIP, "device %d not present (%d visible)", p->device, ndev);

    brie_handle *h = new brie_handle();
    h->p = *p;
    h->ld = round_up(p->Ng, brie::kGenesPerBlock);
    {   // device layout of cell x gene matrices: gene-block-major tiles [gene block][cell][256 genes]
        const char *cst = getenv("BRIE_COUNT_STORAGE");
        h->allow_compact = !(cst && strcmp(cst, "f32") == 0);
        h->row_stride = brie::kGenesPerBlock;
        h->gb_stride = p->Nc * brie::kGenesPerBlock;
    }
    h->mode = !p->has_efflen ? brie::kLik2 : (p->n_layers == 3 ? brie::kLikEff3 : brie::kLikEff2);
    int rc = set_device(h);
    if (rc != BRIE_OK) { delete h; return rc; }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(BRIE_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    const size_t mat = static_cast<size_t>(p->Nc) * h->ld;
    const size_t vec = static_cast<size_t>(h->ld);
#define A(ptr, n)                                            \
    if ((rc = alloc_f32(&(ptr), (n), h->stream)) != BRIE_OK) { drop_failed_create(h); return rc; }
#define AM(ptr)                                              \
    if ((rc = alloc_mat(&(ptr), mat, p->device, h->stream)) != BRIE_OK) { drop_failed_create(h); return rc; }
    g_blocks.prepare(p->device, mat * sizeof(float));
    for (int l = 0; l < p->n_layers; ++l) AM(h->c[l]);
    AM(h->mu); AM(h->rho); AM(h->m_mu); AM(h->v_mu); AM(h->m_rho); AM(h->v_rho);
#undef AM
    A(h->Xc, static_cast<size_t>(p->Nc) * p->Kc);
    // (Wc_loc and Xg: 64 readable zero rows beyond the last feature, what the last panel of panel_prior_mean reads past kp)
    A(h->W, p->Kc ? vec * (p->Kc + BRIE_MAX_KC_WIDE) : 0); A(h->m_W, vec * p->Kc); A(h->v_W, vec * p->Kc);
    A(h->b, vec); A(h->m_b, vec); A(h->v_b, vec);
    A(h->lam, vec); A(h->m_lam, vec); A(h->v_lam, vec);
    A(h->effL, vec * 6);
    A(h->gene_tmp, vec);
    A(h->gene_active, vec);
    A(h->ring_kl, vec * brie::kLossRing);
    A(h->ring_ll, vec * brie::kLossRing);
    h->cell_mode = p->intercept_mode == 1;
    h->coupled = p->Kg > 0 || h->cell_mode;
    if (h->coupled) {
        const size_t nc = static_cast<size_t>(p->Nc);
        h->gwide = p->Kg > brie::kKgMax;
        h->kgp = h->gwide ? static_cast<int>(round_up(p->Kg, 4)) : brie::kKgMax;
        A(h->Xg, vec * (h->kgp + BRIE_MAX_KG_WIDE));
        A(h->Wg, nc * h->kgp); A(h->m_Wg, nc * h->kgp); A(h->v_Wg, nc * h->kgp);
        A(h->cb, nc); A(h->m_cb, nc); A(h->v_cb, nc);
        A(h->clam, nc); A(h->m_clam, nc); A(h->v_clam, nc);
        A(h->rowstat, nc * (h->kgp + 2));
        h->part_kgp = p->Kg > BRIE_MAX_KG_WIDE ? brie::kKgMax : h->kgp;
        if (p->Kg > BRIE_MAX_KG_WIDE) { A(h->Xg_zero, vec * brie::kKgMax); A(h->Wg_zero, nc * brie::kKgMax); A(h->XgT, vec * h->kgp); }
    }
#undef A
    setup_paths(h);                 // kernel family: register path, LDS-broadcast wide variants, or the MFMA tile kernel
    if (h->vwide && prepare_wide_panels(h->p.device) != BRIE_OK) { drop_failed_create(h); return BRIE_ERR_UNSUPPORTED; }
    configure_tiling(h);
    {
        const char *pk = getenv("BRIE_PACK_ACTIVE");
        h->allow_pack = !(pk && strcmp(pk, "0") == 0);
        h->perm.resize(static_cast<size_t>(h->ld / 4));
        for (size_t q = 0; q < h->perm.size(); ++q) h->perm[q] = static_cast<int32_t>(q);
        e = hipMalloc(reinterpret_cast<void **>(&h->quad_ids), h->perm.size() * sizeof(int32_t));
        if (e != hipSuccess) { drop_failed_create(h); return fail(BRIE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
        e = hipMalloc(reinterpret_cast<void **>(&h->block_active), h->gene_blocks * sizeof(int32_t));
        if (e != hipSuccess) { drop_failed_create(h); return fail(BRIE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
        if ((rc = brie_set_gene_mask(h, nullptr)) != BRIE_OK) { drop_failed_create(h); return rc; }
    }
    if (h->coupled) {
        float *rp = nullptr;
        if ((rc = alloc_f32(&rp, static_cast<size_t>(h->gene_blocks) * (h->part_kgp + 2) * p->Nc, h->stream)) != BRIE_OK) {
            drop_failed_create(h);
            return rc;
        }
        h->row_partials = rp;
    }
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { drop_failed_create(h); return fail(BRIE_ERR_HIP, "sync: %s", hipGetErrorString(e)); }
    *out = h;
    return BRIE_OK;
}

// The next model of a likelihood-ratio test on the SAME counts (model_wrap.py:155-187 builds a fresh BRIE2 per tested
// feature and hands it the same count layers): keep the count layers as they sit in HBM (uploaded, pseudo-counted,
// compacted once), replace everything that depends on the design width and the seed.
int brie_reconfigure(brie_handle *h, int32_t Kc, uint64_t seed, int32_t train_intercept, int32_t train_sigma) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (Kc < 0 || Kc > BRIE_MAX_KC_PANELS) return fail(BRIE_ERR_UNSUPPORTED, "Kc=%d outside 0..%d", Kc, BRIE_MAX_KC_PANELS);
    if (h->step_open) return fail(BRIE_ERR_STATE, "a step is open");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t vec = static_cast<size_t>(h->ld);
    float **drop[] = {&h->Xc, &h->W, &h->m_W, &h->v_W, &h->Rbuf, &h->Gpart, &h->Mbuf};
    for (float **q : drop) {
        if (*q) HIP_TRY(hipFree(*q));
        *q = nullptr;
    }
#define A(ptr, n) if ((rc = alloc_f32(&(ptr), (n), h->stream)) != BRIE_OK) return rc;
    A(h->Xc, static_cast<size_t>(h->p.Nc) * Kc);
    A(h->W, Kc ? vec * (Kc + BRIE_MAX_KC_WIDE) : 0); A(h->m_W, vec * Kc); A(h->v_W, vec * Kc);
#undef A
    h->gpart_elems = h->rbuf_elems = 0;
    h->p.Kc = Kc; h->p.seed = seed; h->p.train_intercept = train_intercept; h->p.train_sigma = train_sigma;
    setup_paths(h);
    if (h->vwide && (rc = prepare_wide_panels(h->p.device)) != BRIE_OK) return rc;
    configure_tiling(h);            // the rows per chunk follow the kernel family
    if (h->row_scratch) { HIP_TRY(hipFree(h->row_scratch)); h->row_scratch = nullptr; }   // sized by max(ring, Kc)
    h->have_xc = false;
    h->have_state = false;
    h->draw = 0; h->t = 0; h->ring_pos = 0;
    h->target = 0;
    HIP_TRY(hipMemsetAsync(h->ring_kl, 0, vec * brie::kLossRing * sizeof(float), h->stream));
    HIP_TRY(hipMemsetAsync(h->ring_ll, 0, vec * brie::kLossRing * sizeof(float), h->stream));
    return brie_set_gene_mask(h, nullptr);
}

int brie_destroy(brie_handle *h) {
    if (!h) return BRIE_OK;
    hipSetDevice(h->p.device);
    // a pending asynchronous export still launches export_slab kernels that read mu / rho (error paths: the caller's
    // loss_gene raised between brie_read_results_async and brie_read_wait, or a C client skipped the wait): let the
    // worker and its stream finish BEFORE anything is freed
    if (h->io_thread.joinable()) h->io_thread.join();
    if (h->io_stream) hipStreamSynchronize(h->io_stream);
    if (h->io_stream2) hipStreamSynchronize(h->io_stream2);
    h->io_pending = false;
    if (h->stream) hipStreamSynchronize(h->stream);
    {   // the cell x gene arrays go to the one-generation cache (next handle of this size), everything else is freed
        const size_t mat_bytes = static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float);
        float *mats[] = {h->mu, h->rho, h->m_mu, h->v_mu, h->m_rho, h->v_rho, h->c[0], h->c[1], h->c[2]};
        for (float *q : mats) g_blocks.give(h->p.device, mat_bytes, q);
    }
    float *ptrs[] = {h->Xc,
                     h->W, h->m_W, h->v_W, h->b, h->m_b, h->v_b, h->lam, h->m_lam, h->v_lam, h->effL,
                     h->gene_tmp, h->partials, h->Xg, h->Wg, h->m_Wg, h->v_Wg, h->cb, h->m_cb, h->v_cb, h->clam,
                     h->m_clam, h->v_clam, h->row_partials, h->rowstat, h->Mbuf, h->Rbuf, h->Gpart, h->gene_active,
                     h->ring_kl, h->ring_ll, h->Xg_zero, h->Wg_zero, h->XgT};
    for (float *q : ptrs)
        if (q) hipFree(q);
    if (h->loss_parts) hipFree(h->loss_parts);
    if (h->block_active) hipFree(h->block_active);
    if (h->quad_ids) hipFree(h->quad_ids);
    if (h->pack_scratch) hipFree(h->pack_scratch);
    if (h->row_scratch) hipFree(h->row_scratch);
    if (h->io_scratch) hipFree(h->io_scratch);
    if (h->persist_args) hipFree(h->persist_args);
    if (h->persist_alphas) hipFree(h->persist_alphas);
    if (h->persist_barrier) hipFree(h->persist_barrier);
    if (h->persist_flag_host) hipHostFree(h->persist_flag_host);
    if (h->persist_copy_event) hipEventDestroy(h->persist_copy_event);
    if (h->partials2) hipFree(h->partials2);
    if (h->win_scratch) hipFree(h->win_scratch);
    if (h->io_stream) hipStreamDestroy(h->io_stream);
    if (h->io_stream2) hipStreamDestroy(h->io_stream2);
    if (h->io_event) hipEventDestroy(h->io_event);
    if (h->io_slab) hipFree(h->io_slab);
    (void)free_tier_tables(h);
    for (void *q : h->cu)
        if (q) hipFree(q);
    for (hipEvent_t ev : h->ev_pool) hipEventDestroy(ev);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return BRIE_OK;
}

int brie_upload(brie_handle *h, int which, const float *src, int64_t rows, int64_t cols, int64_t ld) {
    if (!h || (!src && rows * cols > 0)) return fail(BRIE_ERR_INVALID, "null argument");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    if (ld < cols) return fail(BRIE_ERR_INVALID, "ld=%lld < cols=%lld", (long long)ld, (long long)cols);
    if ((rc = order_after_device_source(src)) != BRIE_OK) return rc;
    if (which == BRIE_EFFLEN) {
        if (!h->p.has_efflen) return fail(BRIE_ERR_INVALID, "problem was created without effLen");
        if (rows != h->p.Ng || cols != 6)
            return fail(BRIE_ERR_INVALID, "effLen must be (Ng=%lld, 6), got (%lld, %lld)", (long long)h->p.Ng,
                        (long long)rows, (long long)cols);
        // effLen[:, [0, 4, 5]] (model_TFProb.py:176) -> rows 0..2 of effL, logs in rows 3..5
        std::vector<float> tmp(static_cast<size_t>(rows) * 6), packed(static_cast<size_t>(3) * h->ld, 1.0f);
        HIP_TRY(hipMemcpy2D(tmp.data(), 6 * sizeof(float), src, ld * sizeof(float), 6 * sizeof(float), rows,
                            hipMemcpyDefault));
        const int sel[3] = {0, 4, 5};
        for (int64_t j = 0; j < rows; ++j)
            for (int s = 0; s < 3; ++s) packed[s * h->ld + j] = tmp[j * 6 + sel[s]];
        HIP_TRY(hipMemcpyAsync(h->effL, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(brie::log_rows, dim3((h->p.Ng + 255) / 256), dim3(256), 0, h->stream, h->effL, h->ld,
                           static_cast<int>(h->p.Ng));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->have_eff = true;
        return BRIE_OK;
    }
    if (which == BRIE_XG) {
        if (rows != h->p.Ng || cols != h->p.Kg)
            return fail(BRIE_ERR_INVALID, "Xg must be (Ng=%lld, Kg=%d), got (%lld, %lld)", (long long)h->p.Ng, h->p.Kg,
                        (long long)rows, (long long)cols);
        if (h->p.Kg == 0) return BRIE_OK;
        std::vector<float> tmp(static_cast<size_t>(rows) * cols), packed(static_cast<size_t>(h->kgp) * h->ld, 0.0f);
        HIP_TRY(hipMemcpy2D(tmp.data(), cols * sizeof(float), src, ld * sizeof(float), cols * sizeof(float), rows,
                            hipMemcpyDefault));
        for (int64_t j = 0; j < rows; ++j)
            for (int64_t k = 0; k < cols; ++k) packed[k * h->ld + j] = tmp[j * cols + k];
        HIP_TRY(hipMemcpyAsync(h->Xg, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->XgT) {                            // gene-major copy, zero beyond (Ng, Kg)
            std::fill(packed.begin(), packed.end(), 0.0f);
            for (int64_t j = 0; j < rows; ++j)
                for (int64_t k = 0; k < cols; ++k) packed[j * h->kgp + k] = tmp[j * cols + k];
            HIP_TRY(hipMemcpyAsync(h->XgT, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        h->have_xg = true;
        return BRIE_OK;
    }
    if (which >= BRIE_COUNT1 && which <= BRIE_COUNT3 && (h->cs != brie::kCountF32 || h->compact_tried)) {
        // a fresh layer restarts the storage decision: back to plain fp32 layers first
        if ((rc = expand_counts(h)) != BRIE_OK) return rc;
        h->compact_tried = false;
        const char *cst = getenv("BRIE_COUNT_STORAGE");
        h->allow_compact = !(cst && strcmp(cst, "f32") == 0);
    }
    float *dev = nullptr;
    int64_t R = 0, C = 0, ldd = 0;
    rc = matrix_target(h, which, &dev, &R, &C, &ldd);
    if (rc != BRIE_OK) return rc;
    if (rows != R || cols != C)
        return fail(BRIE_ERR_INVALID, "array %d must be (%lld, %lld), got (%lld, %lld)", which, (long long)R,
                    (long long)C, (long long)rows, (long long)cols);
    if (R * C > 0) {
        const bool cellgene = (which <= BRIE_COUNT3) || which == BRIE_Z_LOC || which == BRIE_Z_STD_LOG;
        if (which <= BRIE_COUNT3 && use_staged_ingest(h, src, R * C)) {
            if ((rc = staged_count_upload(h, dev, src, BRIE_DT_F32, ld)) != BRIE_OK) return rc;
        } else if (cellgene) {
            if ((rc = copy_cellgene(h, dev, src, nullptr, ld)) != BRIE_OK) return rc;
        } else {
            HIP_TRY(hipMemcpy2DAsync(dev, ldd * sizeof(float), src, ld * sizeof(float), C * sizeof(float), R,
                                     hipMemcpyDefault, h->stream));
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (which >= BRIE_COUNT1 && which <= BRIE_COUNT3) h->have_c[which - BRIE_COUNT1] = true;
    if (which == BRIE_XC) h->have_xc = true;
    if (which == BRIE_Z_LOC || which == BRIE_Z_STD_LOG) h->have_state = true;
    return BRIE_OK;
}

// A count layer in the element type the caller holds it in (the reference casts on the host: .astype(np.float32)).
// Always host memory; converted by the staged pipeline's threads (u16 slabs when integral, else the float32 cast).
int brie_upload_typed(brie_handle *h, int which, const void *src, int32_t dtype, int64_t rows, int64_t cols, int64_t ld) {
    if (!h || (!src && rows * cols > 0)) return fail(BRIE_ERR_INVALID, "null argument");
    if (which < BRIE_COUNT1 || which > BRIE_COUNT3 || which - BRIE_COUNT1 >= h->p.n_layers)
        return fail(BRIE_ERR_INVALID, "brie_upload_typed takes a count layer (got array %d)", which);
    if (dtype_size(dtype) == 0) return fail(BRIE_ERR_INVALID, "dtype %d (see brie_dtype)", dtype);
    if (rows != h->p.Nc || cols != h->p.Ng)
        return fail(BRIE_ERR_INVALID, "layer must be (%lld, %lld), got (%lld, %lld)", (long long)h->p.Nc,
                    (long long)h->p.Ng, (long long)rows, (long long)cols);
    if (ld < cols) return fail(BRIE_ERR_INVALID, "ld=%lld < cols=%lld", (long long)ld, (long long)cols);
    if (dtype == BRIE_DT_F32) return brie_upload(h, which, static_cast<const float *>(src), rows, cols, ld);
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess) {
        if (at.type == hipMemoryTypeDevice) return fail(BRIE_ERR_UNSUPPORTED, "typed layers are converted on the host: pass host memory");
    } else {
        (void)hipGetLastError();
    }
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    if (h->cs != brie::kCountF32 || h->compact_tried) {               // a fresh layer restarts the storage decision
        if ((rc = expand_counts(h)) != BRIE_OK) return rc;
        h->compact_tried = false;
        const char *cst = getenv("BRIE_COUNT_STORAGE");
        h->allow_compact = !(cst && strcmp(cst, "f32") == 0);
    }
    if ((rc = staged_count_upload(h, h->c[which - BRIE_COUNT1], src, dtype, ld)) != BRIE_OK) return rc;
    h->have_c[which - BRIE_COUNT1] = true;
    return BRIE_OK;
}

int brie_upload_sparse(brie_handle *h, int which, int32_t format, const int64_t *indptr, const int32_t *indices,
                       const float *data, int64_t nnz, int64_t rows, int64_t cols) {
    if (!h || !indptr || (nnz > 0 && (!indices || !data))) return fail(BRIE_ERR_INVALID, "null argument");
    if (which < BRIE_COUNT1 || which > BRIE_COUNT3 || which - BRIE_COUNT1 >= h->p.n_layers)
        return fail(BRIE_ERR_INVALID, "brie_upload_sparse takes a count layer (got array %d)", which);
    if (format != 0 && format != 1) return fail(BRIE_ERR_INVALID, "format %d (0 = CSC, 1 = CSR)", format);
    if (rows != h->p.Nc || cols != h->p.Ng)
        return fail(BRIE_ERR_INVALID, "layer must be (%lld, %lld), got (%lld, %lld)", (long long)h->p.Nc,
                    (long long)h->p.Ng, (long long)rows, (long long)cols);
    if (nnz < 0) return fail(BRIE_ERR_INVALID, "nnz=%lld", (long long)nnz);
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    if ((rc = order_after_device_source(data ? static_cast<const void *>(data) : static_cast<const void *>(indptr))) != BRIE_OK)
        return rc;
    if (h->cs != brie::kCountF32 || h->compact_tried) {
        if ((rc = expand_counts(h)) != BRIE_OK) return rc;
        h->compact_tried = false;
        const char *cst = getenv("BRIE_COUNT_STORAGE");
        h->allow_compact = !(cst && strcmp(cst, "f32") == 0);
    }
    const int64_t n_major = format == 1 ? rows : cols;
    int64_t *d_ptr = nullptr;
    int32_t *d_idx = nullptr;
    float *d_val = nullptr;
    auto cleanup = [&]() { hipFree(d_ptr); hipFree(d_idx); hipFree(d_val); };
    hipError_t e = dev_alloc(&d_ptr, (n_major + 1) * sizeof(int64_t));
    if (e == hipSuccess && nnz > 0) e = dev_alloc(&d_idx, nnz * sizeof(int32_t));
    if (e == hipSuccess && nnz > 0) e = dev_alloc(&d_val, nnz * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_ptr, indptr, (n_major + 1) * sizeof(int64_t), hipMemcpyDefault, h->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_idx, indices, nnz * sizeof(int32_t), hipMemcpyDefault, h->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_val, data, nnz * sizeof(float), hipMemcpyDefault, h->stream);
    float *dst = h->c[which - BRIE_COUNT1];
    if (e == hipSuccess) e = hipMemsetAsync(dst, 0, static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float), h->stream);
    if (e == hipSuccess && nnz > 0) {
        hipLaunchKernelGGL(brie::scatter_sparse, dim3(static_cast<unsigned>(n_major)), dim3(128), 0, h->stream, d_ptr,
                           d_idx, d_val, dst, n_major, format, h->row_stride, h->gb_stride);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "sparse upload: %s", hipGetErrorString(e));
    h->have_c[which - BRIE_COUNT1] = true;
    return BRIE_OK;
}

int brie_add_pseudo_count(brie_handle *h, float pc) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (!h->have_c[0] || !h->have_c[1]) return fail(BRIE_ERR_STATE, "count layers 1 and 2 not uploaded");
    for (int l = 0; l < h->p.n_layers; ++l)
        if (!h->have_c[l]) return fail(BRIE_ERR_STATE, "count layer %d not uploaded", l + 1);
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    if (h->cs != brie::kCountF32 && (rc = expand_counts(h)) != BRIE_OK) return rc;   // second pseudo-count: fp32
    if ((rc = try_compact_counts(h)) != BRIE_OK) return rc;
    if (h->cs != brie::kCountF32) {           // integer counts: keep them compact, add `pc` in registers
        h->pc = pc;
        return BRIE_OK;
    }
    const int64_t n4 = h->p.Nc * h->ld / 4;
    hipLaunchKernelGGL(brie::pseudo_count, dim3(grid_1d(n4)), dim3(256), 0, h->stream, h->c[0], h->c[1], n4, pc);
    HIP_TRY(hipGetLastError());
    h->compact_tried = true;                  // fractional values from now on
    return BRIE_OK;
}

int brie_reset_optimizer(brie_handle *h) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    const size_t mat = static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float), vec = h->ld * sizeof(float);
    HIP_TRY(hipMemsetAsync(h->m_mu, 0, mat, h->stream));
    HIP_TRY(hipMemsetAsync(h->v_mu, 0, mat, h->stream));
    HIP_TRY(hipMemsetAsync(h->m_rho, 0, mat, h->stream));
    HIP_TRY(hipMemsetAsync(h->v_rho, 0, mat, h->stream));
    if (h->p.Kc > 0) {
        HIP_TRY(hipMemsetAsync(h->m_W, 0, vec * h->p.Kc, h->stream));
        HIP_TRY(hipMemsetAsync(h->v_W, 0, vec * h->p.Kc, h->stream));
    }
    HIP_TRY(hipMemsetAsync(h->m_b, 0, vec, h->stream));
    HIP_TRY(hipMemsetAsync(h->v_b, 0, vec, h->stream));
    HIP_TRY(hipMemsetAsync(h->m_lam, 0, vec, h->stream));
    HIP_TRY(hipMemsetAsync(h->v_lam, 0, vec, h->stream));
    if (h->coupled) {
        const size_t nc = static_cast<size_t>(h->p.Nc) * sizeof(float);
        HIP_TRY(hipMemsetAsync(h->m_Wg, 0, nc * h->kgp, h->stream));
        HIP_TRY(hipMemsetAsync(h->v_Wg, 0, nc * h->kgp, h->stream));
        HIP_TRY(hipMemsetAsync(h->m_cb, 0, nc, h->stream));
        HIP_TRY(hipMemsetAsync(h->v_cb, 0, nc, h->stream));
        HIP_TRY(hipMemsetAsync(h->m_clam, 0, nc, h->stream));
        HIP_TRY(hipMemsetAsync(h->v_clam, 0, nc, h->stream));
    }
    h->t = 0;
    return BRIE_OK;
}

int brie_init_state(brie_handle *h, float intercept, float sigma) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    const uint32_t slo = static_cast<uint32_t>(h->p.seed & 0xFFFFFFFFull), shi = static_cast<uint32_t>(h->p.seed >> 32);
    const uint32_t qoff = static_cast<uint32_t>(h->p.gene_offset / 4);
    const int Nc = static_cast<int>(h->p.Nc), Ng = static_cast<int>(h->p.Ng);
    const int64_t quads = (Ng + 3) / 4;
    hipLaunchKernelGGL(brie::init_z, dim3(grid_1d(static_cast<int64_t>(h->gene_blocks) * Nc * brie::kWave)), dim3(256), 0,
                       h->stream, h->mu, h->rho, h->row_stride, h->gb_stride, h->gene_blocks, Nc, Ng, slo, shi, qoff);
    if (h->p.Kc > 0)
        hipLaunchKernelGGL(brie::init_gene_rows, dim3(grid_1d(h->p.Kc * quads)), dim3(256), 0, h->stream, h->W, h->ld,
                           h->p.Kc, Ng, 2u, slo, shi, qoff);
    if (std::isnan(intercept))
        hipLaunchKernelGGL(brie::init_gene_rows, dim3(grid_1d(quads)), dim3(256), 0, h->stream, h->b, h->ld, 1, Ng, 3u,
                           slo, shi, qoff);
    else
        hipLaunchKernelGGL(brie::fill_f32, dim3(grid_1d(Ng)), dim3(256), 0, h->stream, h->b, static_cast<int64_t>(Ng),
                           intercept);
    const float lam0 = std::isnan(sigma) ? 0.0f : logf(sigma);
    hipLaunchKernelGGL(brie::fill_f32, dim3(grid_1d(Ng)), dim3(256), 0, h->stream, h->lam, static_cast<int64_t>(Ng), lam0);
    if (h->coupled) {       // per-cell parameters (shapes of model_TFProb.py:53-55)
        const bool rnd_cb = h->cell_mode && std::isnan(intercept);
        hipLaunchKernelGGL(brie::init_cell_params, dim3((Nc + 255) / 256), dim3(256), 0, h->stream, h->Wg, h->cb, Nc,
                           h->p.Kg, h->kgp, rnd_cb ? 1 : 0, slo, shi);
        if (h->cell_mode && !rnd_cb)
            hipLaunchKernelGGL(brie::fill_f32, dim3(grid_1d(Nc)), dim3(256), 0, h->stream, h->cb, static_cast<int64_t>(Nc),
                               intercept);
        hipLaunchKernelGGL(brie::fill_f32, dim3(grid_1d(Nc)), dim3(256), 0, h->stream, h->clam, static_cast<int64_t>(Nc),
                           lam0);
    }
    HIP_TRY(hipGetLastError());
    h->have_state = true;
    return brie_reset_optimizer(h);
}

int brie_set_tiling(brie_handle *h, int32_t rows_per_chunk) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (rows_per_chunk < 0 || (rows_per_chunk > 0 && rows_per_chunk < brie::kWavesPerBlock))
        return fail(BRIE_ERR_INVALID, "rows_per_chunk=%d", rows_per_chunk);
    h->user_rows_per_chunk = rows_per_chunk;
    configure_tiling(h);
    return BRIE_OK;
}

int brie_set_gene_mask(brie_handle *h, const uint8_t *active) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if (active && h->coupled)
        return fail(BRIE_ERR_UNSUPPORTED, "per-gene freezing is meaningless for coupled fits (one joint problem)");
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;        // `active` is in original gene order
    std::vector<float> mask(static_cast<size_t>(h->ld), 0.0f);
    bool any_frozen = false;
    for (int64_t j = 0; j < h->p.Ng; ++j) {
        const bool on = active ? active[j] != 0 : true;
        mask[j] = on ? 1.0f : 0.0f;
        any_frozen |= !on;
    }
    HIP_TRY(hipMemcpyAsync(h->gene_active, mask.data(), mask.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->mask_host = mask;
    h->any_frozen = any_frozen;
    if (any_frozen && h->allow_pack && !h->wide_like) {
        // pack: quads with an active gene first (stable), fully frozen quads after them; whole 256-gene
        // blocks at the tail then hold no active gene and are skipped by the kernels
        const int nq = static_cast<int>(h->ld / 4);
        std::vector<int32_t> order;
        order.reserve(nq);
        auto quad_on = [&](int q) { return mask[4 * q] + mask[4 * q + 1] + mask[4 * q + 2] + mask[4 * q + 3] > 0.0f; };
        for (int q = 0; q < nq; ++q) if (quad_on(q)) order.push_back(q);
        const size_t n_on = order.size();
        for (int q = 0; q < nq; ++q) if (!quad_on(q)) order.push_back(q);
        if (n_on < static_cast<size_t>(nq)) {
            if ((rc = apply_quad_gather(h, order)) != BRIE_OK) return rc;
            h->perm = order;
            h->packed = true;
            std::vector<float> by_pos(static_cast<size_t>(h->ld));
            for (int p = 0; p < nq; ++p)
                for (int v = 0; v < 4; ++v) by_pos[4 * p + v] = mask[4 * order[p] + v];
            return upload_quad_ids_and_blocks(h, by_pos);
        }
    }
    return upload_quad_ids_and_blocks(h, mask);
}

int brie_read_loss_window(brie_handle *h, int32_t n_last, float *out) {
    if (!h || !out) return fail(BRIE_ERR_INVALID, "null argument");
    if (n_last < 1 || n_last > brie::kLossRing || n_last > h->ring_pos)
        return fail(BRIE_ERR_INVALID, "n_last=%d (ring holds %d steps, %lld taken)", n_last, brie::kLossRing,
                    (long long)h->ring_pos);
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    const size_t n = static_cast<size_t>(n_last) * h->p.Ng;
    if ((rc = ensure_f32(&h->win_scratch, &h->win_elems, n, h->stream)) != BRIE_OK) return rc;    // kept: one call per round
    hipLaunchKernelGGL(brie::loss_window, dim3(h->fin_blocks, n_last), dim3(brie::kBlock), 0, h->stream, h->ring_kl,
                       h->ring_ll, h->win_scratch, h->ld, static_cast<int>(h->p.Ng), n_last, h->ring_pos);
    hipError_t e = hipMemcpyAsync(out, h->win_scratch, n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "loss window: %s", hipGetErrorString(e));
    return BRIE_OK;
}

int brie_set_target(brie_handle *h, int32_t target) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (target != 0 && target != 1) return fail(BRIE_ERR_INVALID, "target %d (0 = ELBO, 1 = marginLik)", target);
    h->target = target;
    return BRIE_OK;
}

int brie_set_count_storage(brie_handle *h, int32_t mode) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (mode != 0 && mode != 1) return fail(BRIE_ERR_INVALID, "count storage mode %d (0 = auto, 1 = fp32)", mode);
    if (mode == 1) {
        int rc = set_device(h);
        if (rc != BRIE_OK) return rc;
        if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
        if ((rc = expand_counts(h)) != BRIE_OK) return rc;
    }
    h->allow_compact = mode == 0;
    return BRIE_OK;
}

int brie_get_count_storage(const brie_handle *h) { return h ? h->cs : -1; }

int64_t brie_step_storage_bytes(const brie_handle *h) {
    if (!h) return 0;
    // LDS-broadcast wide variants: residual r written by the step, read back by wide_design_grad (the tile kernel keeps it on chip)
    int64_t gemm_streams = (h->wide_like && !h->tile) ? 8 : 0;
    if (h->vwide) {
        // the prior-mean array: written by the first forward launch (the cell design's), read and written by the gene design's; read by the step, which writes the residual back (8);
        // the residual read once per backward launch: one per 256 cell features, one per 64 gene features
        const int64_t Kc = h->p.Kc;
        const int64_t fwd_launches = (Kc > 0 ? 1 : 0) + (h->vgwide ? 1 : 0);       // (designs of <= 64 features: one panel)
        const int64_t fwd = fwd_launches > 0 ? 4 + 8 * (fwd_launches - 1) : 0;
        const int64_t bwd_c = Kc > 0 ? (Kc + 255) / 256 : 0;
        const int64_t bwd_g = h->vgwide ? (h->kgp + BRIE_MAX_KG_WIDE - 1) / BRIE_MAX_KG_WIDE : 0;
        gemm_streams = fwd + 8 + 4 * (bwd_c + bwd_g);
    }
    if (h->cs == brie::kCountMixed) {                    // genes of u8 quads move 1 byte per count, of u16 quads 2
        int64_t genes16 = 0;
        for (int64_t q = 0; q * brie::kVec < h->p.Ng; ++q)
            if (h->q_esz_host[static_cast<size_t>(q)] == 2) genes16 += std::min<int64_t>(brie::kVec, h->p.Ng - q * brie::kVec);
        return h->p.Nc * (h->p.Ng * (48 + gemm_streams) + static_cast<int64_t>(h->p.n_layers) * (h->p.Ng + genes16));
    }
    const int64_t per_count = h->cs == brie::kCountU8 ? 1 : (h->cs == brie::kCountU16 ? 2 : 4);
    return h->p.Nc * h->p.Ng * (48 + per_count * static_cast<int64_t>(h->p.n_layers) + gemm_streams);
}

int64_t brie_step_algorithmic_bytes(const brie_handle *h) {
    if (!h) return 0;
    return h->p.Nc * h->p.Ng * (48 + 4 * static_cast<int64_t>(h->p.n_layers));     // SURVEY 8d fp32 model
}

}  // extern "C"

namespace {

// Wide designs (Kc = 9..64): G = Xc^T . r on the matrix cores, then Adam for Wc_loc.
int wide_backward(brie_handle *h, float alpha) {
    const dim3 grid(h->gene_blocks, h->n_gchunks), block(512);
    const int Kc = h->p.Kc;
    // one launch per 256 features (one for Kc <= 256): the residual is read once per launch, every feature keeps its own
    // accumulator, so the sums are those of the 64-feature launches of round 4 bit for bit
    constexpr int kGroup = 256;
    for (int k0 = 0; k0 < Kc; k0 += kGroup) {
        const int kp = std::min(Kc - k0, kGroup);
        const float *xp = h->Xc + k0;
        float *gp = h->Gpart + static_cast<size_t>(k0) * h->ld;
#define BRIE_WDG(NACC)                                                                                                  \
    hipLaunchKernelGGL((brie::wide_design_grad<NACC>), grid, block, 0, h->stream, xp, h->Rbuf, gp,                     \
                       static_cast<int>(h->p.Nc), kp, h->ld, h->gb_stride, h->gchunk_rows, Kc, Kc)
#define BRIE_WDG_LDS(NACC)                                                                                              \
    hipLaunchKernelGGL((brie::wide_design_grad_lds<NACC>), grid, block, 0, h->stream, xp, h->Rbuf, gp,                 \
                       static_cast<int>(h->p.Nc), kp, h->ld, h->gb_stride, h->gchunk_rows, Kc, Kc)
        if (kp <= 32) BRIE_WDG(1);
        else if (kp <= 64) BRIE_WDG(2);
        else if (kp <= 128) BRIE_WDG_LDS(4);           // design operands shared through LDS: 2.55 instead of 3.05 ms per 128
        else BRIE_WDG_LDS(8);                          // features at configs[2] (calls r5g, r5h)
#undef BRIE_WDG_LDS
#undef BRIE_WDG
    }
    const int64_t nW = static_cast<int64_t>(h->p.Kc) * h->ld;
    hipLaunchKernelGGL(brie::wide_w_adam, dim3(grid_1d(nW)), dim3(256), 0, h->stream, h->W, h->m_W, h->v_W, h->Gpart, nW,
                       h->n_gchunks, alpha, h->gene_active, h->ld);
    HIP_TRY(hipGetLastError());
    return BRIE_OK;
}

// dst (Nc, ld) tiled (+)= X . B over kp <= 64 features: a handful through the LDS-broadcast kernel, more on the matrix cores
void launch_panel(brie_handle *h, const float *X, int x_ld, const float *B, int kp, float *dst, bool accumulate) {
    if (kp <= 8) {
        hipLaunchKernelGGL(brie::wide_prior_mean, dim3(h->gene_blocks, h->n_chunks), dim3(brie::kBlock), 0, h->stream, X, B, dst,
                           static_cast<int>(h->p.Nc), static_cast<int>(h->p.Ng), kp, h->ld, h->row_stride, h->gb_stride,
                           h->rows_per_chunk, accumulate ? 1 : 0, x_ld);
        return;
    }
    const dim3 grid(h->gene_blocks, static_cast<unsigned>((h->p.Nc + brie::kWavesPerBlock * 32 - 1) / (brie::kWavesPerBlock * 32)));
    if (kp <= 32)
        hipLaunchKernelGGL((brie::panel_prior_mean<16>), grid, dim3(brie::kBlock), 0, h->stream, X, B, dst,
                           static_cast<int>(h->p.Nc), kp, h->ld, h->row_stride, h->gb_stride, accumulate ? 1 : 0, x_ld);
    else
        hipLaunchKernelGGL((brie::panel_prior_mean<32>), grid, dim3(brie::kBlock), 0, h->stream, X, B, dst,
                           static_cast<int>(h->p.Nc), kp, h->ld, h->row_stride, h->gb_stride, accumulate ? 1 : 0, x_ld);
}

// dst (Nc, ld) tiled (+)= X . B over any number of features: more than 64 in ONE launch of fused_prior_mean (dst read at
// most once, written once; round 4 launched once per 64-feature panel: 16.5 instead of 14.7 ms per step at Kc = 128, call r5m)
// fused_prior_mean keeps two stages of tiles in 140 KB of dynamic LDS: the attribute that allows it belongs to the DEVICE, so it
// is set once per device -- when the first handle that can need it is created (prepare_wide_panels: a device with less LDS is
// refused there with a message that says so, not by a generic launch error a fit later) -- together with the CU count the
// persistent grid is sized by.  g_panel_cus[d]: 0 = not asked yet, > 0 = compute units, < 0 = the device cannot run the kernel.
// compute units of a device (cached; the PERSIST launch must fit the device at once)
int device_cus(int device) {
    static std::atomic<int> cus[64];
    int n = cus[device & 63].load(std::memory_order_acquire);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        (void)hipGetLastError();
        cus[device & 63].store(n, std::memory_order_release);
    }
    return n;
}
std::atomic<int> g_panel_cus[64];
int panel_device_cus(int device) { return g_panel_cus[device & 63].load(std::memory_order_acquire); }
int prepare_wide_panels(int device) {
    int cur = panel_device_cus(device);
    if (cur == 0) {
        static std::mutex mu;
        std::lock_guard<std::mutex> l(mu);
        cur = panel_device_cus(device);
        if (cur == 0) {
            hipDeviceProp_t prop;
            const hipError_t e1 = hipGetDeviceProperties(&prop, device);
            const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(brie::fused_prior_mean),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, brie::kFpmLdsBytes);
            (void)hipGetLastError();
            cur = (e1 == hipSuccess && e2 == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : -1;
            g_panel_cus[device & 63].store(cur, std::memory_order_release);
        }
    }
    if (cur < 0)
        return fail(BRIE_ERR_UNSUPPORTED, "designs beyond 64 features need %d KB of LDS per workgroup (gfx950 has 160 KB): device %d "
                                          "does not grant it", brie::kFpmLdsBytes / 1024, device);
    return BRIE_OK;
}

void launch_panels(brie_handle *h, const float *X, int x_ld, const float *B, int K, float *dst, bool accumulate) {
    if (K > 64) {
        // persistent: one workgroup per CU walks the (gene block x 256-cell block) tiles round-robin
        const int n_cu = panel_device_cus(h->p.device);     // > 0: prepare_wide_panels ran at create and checked the device
        const int64_t n_tiles = static_cast<int64_t>(h->gene_blocks) * ((h->p.Nc + brie::kFpmWaves * 32 - 1) / (brie::kFpmWaves * 32));
        const dim3 grid(static_cast<unsigned>(std::min<int64_t>(n_tiles, n_cu > 0 ? n_cu : 256)));
        hipLaunchKernelGGL(brie::fused_prior_mean, grid, dim3(brie::kFpmWaves * brie::kWave), brie::kFpmLdsBytes, h->stream, X, B, dst,
                           static_cast<int>(h->p.Nc), K, h->ld, h->row_stride, h->gb_stride, accumulate ? 1 : 0, x_ld, h->gene_blocks);
        return;
    }
    launch_panel(h, X, x_ld, B, K, dst, accumulate);
}

// dst (Nc, ld) tiled = Xc . Wc_loc
void launch_xw_panels(brie_handle *h, float *dst) {
    if (h->p.Kc > 0) launch_panels(h, h->Xc, h->p.Kc, h->W, h->p.Kc, dst, false);
}

// Mbuf = Xc . Wc_loc for the forward-only loss_gene pass of a wide design
int cell_finalize_blocks(const brie_handle *h) {
    return static_cast<int>((static_cast<int64_t>(h->kgp + 2) * h->p.Nc + brie::kBlock - 1) / brie::kBlock);
}

// Kg > 4, forward only (loss_gene_eval reads it): Mbuf = Wg_loc . Xg^T -- the same (rows x K).(K x genes) product as
// the wide cell design's Xc . Wc_loc, so the same kernel with (Wg_loc, Xg) in the roles of (Xc, Wc_loc)
// dst (Nc, ld) tiled (+)= Wg_loc . Xg^T, in panels of at most 64 gene features (one panel for Kg <= 64)
void launch_gw_panels(brie_handle *h, float *dst, bool accumulate) {
    launch_panels(h, h->Wg, h->kgp, h->Xg, h->kgp, dst, accumulate);
}

int gwide_forward_mean(brie_handle *h, bool accumulate) {
    if (!h->Mbuf) {
        HIP_TRY(dev_alloc(&h->Mbuf, static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float)));
        HIP_TRY(hipMemsetAsync(h->Mbuf, 0, static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float), h->stream));
    }
    launch_gw_panels(h, h->Mbuf, accumulate);
    HIP_TRY(hipGetLastError());
    return BRIE_OK;
}

int wide_forward_mean(brie_handle *h) {
    if (!h->Mbuf) {
        HIP_TRY(dev_alloc(&h->Mbuf, static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float)));
        HIP_TRY(hipMemsetAsync(h->Mbuf, 0, static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float), h->stream));
    }
    launch_xw_panels(h, h->Mbuf);
    HIP_TRY(hipGetLastError());
    return BRIE_OK;
}

// ---- placement of the streamed arrays ---------------------------------------------------------------------------------
// The step kernel of one and the same handle runs at one of two speeds that differ by 10 - 20 % depending on where
// hipMalloc put its arrays; nothing a user-level program can see predicts it, so it is MEASURED:
// placement_probe (the step kernel's traffic without arithmetic or effect) is timed on the handle's arrays; while the
// rate is below `good_gbs` another set is allocated WHILE the current one is still held (other physical pages),
// filled by device-to-device copies, probed, and the faster set is kept.  A few launches and copies per try.
int probe_rate(brie_handle *h, int iters, double *gbs, bool two_per_cu) {
    brie::StepScalars a{};
    a.ld = h->ld; a.row_stride = h->row_stride; a.gb_stride = h->gb_stride;
    a.Nc = static_cast<int32_t>(h->p.Nc); a.Ng = static_cast<int32_t>(h->p.Ng);
    a.rows_per_chunk = h->rows_per_chunk;
    a.tt = h->tt;
    const bool compact = h->cs != brie::kCountF32;
    const void *c[3];
    for (int l = 0; l < 3; ++l) c[l] = compact ? static_cast<const void *>(h->cu[l]) : h->c[l];
    const bool l3 = h->mode == brie::kLikEff3;      // the layers the step READS: the third only with effLen (model_TFProb.py:184-185)
    const dim3 grid(h->gene_blocks, h->n_chunks), block(brie::kBlock);
    // the step's occupancy (brie_inst.hip): one workgroup per CU for MC_size 1 (2 x pad > 160 KB), two otherwise (3 x pad > 160 KB)
    const int pad = two_per_cu ? 54 * 1024 : 81 * 1024;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int it = -1; it < iters; ++it) {
        if (it == 0) (void)hipEventRecord(e0, h->stream);
#define BRIE_PROBE(CS, L3)                                                                                              \
    do {                                                                                                               \
        auto kern = brie::placement_probe<CS, L3>;                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pad); \
        hipLaunchKernelGGL(kern, grid, block, pad, h->stream, c[0], c[1], c[2], h->mu, h->rho, h->m_mu, h->v_mu,       \
                           h->m_rho, h->v_rho, a, 0u, static_cast<uint32_t *>(nullptr));                               \
    } while (0)
#define BRIE_PROBE_CS(CS) do { if (l3) BRIE_PROBE(CS, true); else BRIE_PROBE(CS, false); } while (0)
        switch (h->cs) {
            case brie::kCountU8: BRIE_PROBE_CS(brie::kCountU8); break;
            case brie::kCountU16: BRIE_PROBE_CS(brie::kCountU16); break;
            case brie::kCountMixed: BRIE_PROBE_CS(brie::kCountMixed); break;
            default: BRIE_PROBE_CS(brie::kCountF32); break;
        }
#undef BRIE_PROBE_CS
#undef BRIE_PROBE
    }
    (void)hipEventRecord(e1, h->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "placement probe: %s", hipGetErrorString(e));
    *gbs = static_cast<double>(brie_step_storage_bytes(h)) * iters / (static_cast<double>(ms) * 1e-3) / 1e9;
    return BRIE_OK;
}

// the arrays a step streams: the six state arrays and the count layers in their current storage
struct StreamedSet {
    void *p[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t bytes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int n = 0;
};
StreamedSet streamed_set(brie_handle *h) {
    StreamedSet s;
    const size_t mat_bytes = static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float);
    float *st[6] = {h->mu, h->rho, h->m_mu, h->v_mu, h->m_rho, h->v_rho};
    for (float *q : st) { s.p[s.n] = q; s.bytes[s.n++] = mat_bytes; }
    const bool compact = h->cs != brie::kCountF32;
    for (int l = 0; l < h->p.n_layers; ++l) {
        s.p[s.n] = compact ? h->cu[l] : static_cast<void *>(h->c[l]);
        s.bytes[s.n++] = compact ? compact_layer_bytes(h, h->cs) : mat_bytes;
    }
    return s;
}
void adopt_set(brie_handle *h, const StreamedSet &s) {
    h->mu = static_cast<float *>(s.p[0]); h->rho = static_cast<float *>(s.p[1]);
    h->m_mu = static_cast<float *>(s.p[2]); h->v_mu = static_cast<float *>(s.p[3]);
    h->m_rho = static_cast<float *>(s.p[4]); h->v_rho = static_cast<float *>(s.p[5]);
    const bool compact = h->cs != brie::kCountF32;
    for (int l = 0; l < h->p.n_layers; ++l) {
        if (compact) h->cu[l] = s.p[6 + l];
        else h->c[l] = static_cast<float *>(s.p[6 + l]);
    }
}

// rate a handle of this device has been seen to reach (GB/s of storage bytes): what "fast" means on this box
std::mutex g_place_mu;
double g_place_best[64][2] = {};           // [device][size class of the handle's arrays, see placement_class]
// Arrays below a gigabyte never reached the 6.0 - 6.3 TB/s of large ones in any process of round 4 (216 sets of the
// configs[1] shape: fast mode 5.85 - 6.0, none above 6.0; profiles/r4*_placement_auto_c2.jsonl, r4ae_auto_c2.jsonl): their own class
inline int placement_class(const brie_handle *h) {
    return static_cast<size_t>(h->p.Nc) * h->ld * sizeof(float) < (size_t(1) << 30) ? 1 : 0;
}

// One search: probe the arrays as they are; while the best rate is below `good_gbs`, allocate further candidate sets
// -- in ROUNDS, every earlier set still held so that the allocator has to reach into other physical stretches --, fill
// them by device-to-device copies, probe, and keep the fastest; everything else is freed when the search is over.
// BEST EFFORT: the search is an optimisation, so nothing that goes wrong inside it fails the caller's step --
// the fastest complete set is re-adopted, the candidates are freed, the HIP error is cleared, the reason is kept in
// placement_status / placement_note (brie_placement_status) and the function returns BRIE_OK.  Only an error of the
// caller's EARLIER work on the stream is returned.
//  * memory: before every round hipMemGetInfo is read afresh and the round takes at most `frac` (BRIE_PLACEMENT_HBM_FRACTION,
//    default 0.8) of what is free beyond a 2-GiB reserve; an allocation that fails anyway ends the search with what it has.
//    Peak transient holding = (sets held) x (streamed bytes), reported by brie_placement_status.
//  * time: no new round and no new candidate once `max_seconds` (BRIE_PLACEMENT_SECONDS, default 3) have passed.
//  * sets: at most max_sets <= BRIE_PLACEMENT_MAX_SETS (8) in all, the original included.
int tune_placement(brie_handle *h, int max_sets, double good_gbs, bool two_per_cu, double next_round_gbs = 0.0) {
    max_sets = std::max(1, std::min(max_sets, static_cast<int>(BRIE_PLACEMENT_MAX_SETS)));
    // BRIE_PLACEMENT_INTERLEAVE=0: candidate sets one after the other, each allocated as a block (A/B runs; the first version)
    static const bool interleave = [] { const char *e = getenv("BRIE_PLACEMENT_INTERLEAVE"); return !(e && e[0] == '0'); }();
    static const double env_seconds = [] { const char *e = getenv("BRIE_PLACEMENT_SECONDS"); return e ? atof(e) : 3.0; }();
    static const double env_frac = [] {
        const char *e = getenv("BRIE_PLACEMENT_HBM_FRACTION");
        const double f = e ? atof(e) : 0.8;
        return f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
    }();
    // per handle (brie_placement_configure) before the process environment: several ranks that share one GPU size their
    // rounds on the same hipMemGetInfo reading without knowing of each other -- their caller does, and divides the fraction
    const double max_seconds = h->placement_cfg_seconds > 0.0 ? h->placement_cfg_seconds : env_seconds;
    const double frac = h->placement_cfg_frac > 0.0 ? std::min(1.0, h->placement_cfg_frac) : env_frac;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    HIP_TRY(hipStreamSynchronize(h->stream));               // the caller's earlier work: its errors are its own
    h->placement_done = true;
    h->placement_tries = 0; h->placement_kept = 0; h->placement_peak_bytes = 0; h->placement_note[0] = '\0';
    for (double &g : h->placement_gbs) g = 0.0;
    auto note = [&](const char *what, const char *detail) {
        snprintf(h->placement_note, sizeof(h->placement_note), "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    };
    auto finish = [&](int status) {
        h->placement_status = status;
        (void)hipGetLastError();
        h->placement_seconds += elapsed();
        return BRIE_OK;
    };
    // tests (brie_debug_inject_placement_failure, an explicit call on ONE handle -- nothing in the environment can switch it on):
    // 1 the first probe, 2 a candidate allocation, 3 a candidate's probe, 4 a candidate copy fails in this search
    const int inject = h->placement_inject;
    h->placement_inject = 0;
    const int iters = 3;
    double best = 0.0;
    if (inject == 1 || probe_rate(h, iters, &best, two_per_cu) != BRIE_OK) {      // the arrays stay where they are
        if (inject == 1) (void)fail(BRIE_ERR_HIP, "injected failure");
        note("probe of the original set failed", brie_last_error());
        return finish(BRIE_PLACEMENT_STOPPED_ERROR);
    }
    h->placement_tries = 1; h->placement_gbs[0] = best;
    static const bool log_sets = getenv("BRIE_PLACEMENT_LOG") != nullptr;      // experiments: where every set lives
    auto log_set = [&](int t, double rate) {
        if (!log_sets) return;
        const StreamedSet s = streamed_set(h);
        fprintf(stderr, "[brie placement] set %d  %.1f GB/s ", t, rate);
        for (int i = 0; i < s.n; ++i) fprintf(stderr, " %p(+%zu)", s.p[i], s.bytes[i]);
        fprintf(stderr, "\n");
    };
    log_set(0, best);
    const StreamedSet first = streamed_set(h);
    size_t total = 0;
    for (int i = 0; i < first.n; ++i) total += first.bytes[i];
    auto drop = [](StreamedSet &s) { for (int i = 0; i < s.n; ++i) if (s.p[i]) { (void)hipFree(s.p[i]); s.p[i] = nullptr; } };
    // Candidate sets.  Arrays that lie next to each other in physical memory stream slower than arrays that lie far apart
    // (one slab, the same memory, calls r4k / r4l: packed 5.2 TB/s, the same arrays 18 GB apart 6.3 TB/s in every process),
    // and a block of fresh allocations is handed out back to back.  So the candidates of a round are allocated INTERLEAVED
    // -- array 0 of every candidate, then array 1 of every candidate, ... -- which puts the arrays of one set a few arrays
    // apart at no cost.  A set that is freed early is what the next hipMalloc hands out again (call r4a), so every set is
    // HELD until the search is over: a second round (round 5; the driver's round-4 run needed the last of four sets) has to
    // land somewhere else.  Arrays below a gigabyte are not spread by interleaving ("a few arrays apart" is then no distance,
    // call r4z): their candidates come one at a time (27 allocations and frees up front were most of 15 - 40 ms).
    const int round_size = placement_class(h) == 1 ? 1 : 3;
    std::vector<StreamedSet> held;                // candidate k is set k + 1
    int kept = -1, status = -1;
    // `good_gbs` ends the search at once; `next_round_gbs` <= good_gbs is what it takes NOT to start a further round once one
    // round of candidates has been probed: a first set between the two buys exactly one round (three candidates), not eight
    // sets (round 5: first sets at 6.09 - 6.12 TB/s where the kept ones of other processes read 6.16 - 6.33)
    if (next_round_gbs <= 0.0 || next_round_gbs > good_gbs) next_round_gbs = good_gbs;
    while (best < (held.empty() ? good_gbs : next_round_gbs) && 1 + static_cast<int>(held.size()) < max_sets) {
        if (elapsed() > max_seconds) { status = BRIE_PLACEMENT_STOPPED_TIME; break; }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { note("hipMemGetInfo failed", nullptr); status = BRIE_PLACEMENT_STOPPED_ERROR; break; }
        const size_t reserve = size_t(2) << 30;
        const size_t usable = free_b > reserve ? static_cast<size_t>(static_cast<double>(free_b - reserve) * frac) : 0;
        const int n_round = static_cast<int>(std::min<size_t>({static_cast<size_t>(round_size),
                                                               static_cast<size_t>(max_sets - 1) - held.size(), usable / total}));
        if (n_round <= 0) { status = BRIE_PLACEMENT_STOPPED_MEMORY; break; }
        const size_t base = held.size();
        held.resize(base + static_cast<size_t>(n_round));
        for (size_t k = base; k < held.size(); ++k) { held[k].n = first.n; for (int i = 0; i < first.n; ++i) held[k].bytes[i] = first.bytes[i]; }
        hipError_t e = hipSuccess;
        bool out_of_time = false;                  // the limit is looked at after EVERY allocation (a slow allocator: seconds each)
        auto alloc = [&](size_t k, int i) {
            e = hipMalloc(&held[k].p[i], first.bytes[i]);
            if (e == hipSuccess && elapsed() > max_seconds) out_of_time = true;
        };
        if (interleave) {
            for (int i = 0; i < first.n && e == hipSuccess && !out_of_time; ++i)
                for (size_t k = base; k < held.size() && e == hipSuccess && !out_of_time; ++k) alloc(k, i);
        } else {
            for (size_t k = base; k < held.size() && e == hipSuccess && !out_of_time; ++k)
                for (int i = 0; i < first.n && e == hipSuccess && !out_of_time; ++i) alloc(k, i);
        }
        if (inject == 2 && e == hipSuccess) e = hipErrorOutOfMemory;
        if (e != hipSuccess || out_of_time) {      // someone else took the memory in between, or the allocator is slow today:
            (void)hipGetLastError();               // search over, keep the best so far
            for (size_t k = base; k < held.size(); ++k) drop(held[k]);
            held.resize(base);
            if (e != hipSuccess) note("candidate allocation failed", hipGetErrorString(e));
            status = e != hipSuccess ? BRIE_PLACEMENT_STOPPED_MEMORY : BRIE_PLACEMENT_STOPPED_TIME;
            break;
        }
        h->placement_peak_bytes = std::max<int64_t>(h->placement_peak_bytes, static_cast<int64_t>(held.size() * total));
        for (size_t k = base; k < held.size() && status < 0; ++k) {
            if (best >= good_gbs) break;                            // good enough: the remaining candidates are not needed
            if (k > base && elapsed() > max_seconds) { status = BRIE_PLACEMENT_STOPPED_TIME; break; }
            for (int i = 0; i < first.n && e == hipSuccess; ++i)
                e = hipMemcpyAsync(held[k].p[i], first.p[i], first.bytes[i], hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess && k > base) {     // copies are asynchronous: the limit is looked at once they have run
                e = hipStreamSynchronize(h->stream);
                if (e == hipSuccess && elapsed() > max_seconds) { status = BRIE_PLACEMENT_STOPPED_TIME; break; }
            }
            if (inject == 4 && e == hipSuccess) e = hipErrorInvalidValue;
            if (e != hipSuccess) { note("placement copy failed", hipGetErrorString(e)); status = BRIE_PLACEMENT_STOPPED_ERROR; break; }
            adopt_set(h, held[k]);
            double r = 0.0;
            if (probe_rate(h, iters, &r, two_per_cu) != BRIE_OK || (inject == 3 && fail(BRIE_ERR_HIP, "injected failure") != BRIE_OK)) {
                note("probe of a candidate set failed", brie_last_error());
                status = BRIE_PLACEMENT_STOPPED_ERROR;
                break;
            }
            const int set_no = static_cast<int>(k) + 1;
            h->placement_gbs[set_no] = r;
            h->placement_tries = set_no + 1;
            log_set(set_no, r);
            if (r > best) { best = r; kept = static_cast<int>(k); h->placement_kept = set_no; }
        }
        if (status >= 0) break;
    }
    if (status < 0) status = best >= next_round_gbs ? BRIE_PLACEMENT_GOOD : BRIE_PLACEMENT_BEST_OF_ALL;
    adopt_set(h, kept < 0 ? first : held[static_cast<size_t>(kept)]);     // the fastest COMPLETE set (first is never written to)
    (void)hipStreamSynchronize(h->stream);
    for (size_t k = 0; k < held.size(); ++k)
        if (static_cast<int>(k) != kept) drop(held[k]);
    if (kept >= 0) { StreamedSet f = first; drop(f); }
    if (status == BRIE_PLACEMENT_BEST_OF_ALL || status == BRIE_PLACEMENT_STOPPED_MEMORY || status == BRIE_PLACEMENT_STOPPED_TIME) {
        if (!h->placement_note[0]) {
            char buf[96];
            snprintf(buf, sizeof(buf), "best of %d sets %.0f GB/s, below the stop rate %.0f GB/s", h->placement_tries, best, good_gbs);
            note(status == BRIE_PLACEMENT_BEST_OF_ALL ? "no set reached the stop rate"
                 : (status == BRIE_PLACEMENT_STOPPED_MEMORY ? "stopped by free HBM" : "stopped by the time limit"), buf);
        }
    }
    {
        std::lock_guard<std::mutex> l(g_place_mu);
        const int d = h->p.device & 63;
        double &seen = g_place_best[d][placement_class(h)];
        if (best > seen) seen = best;
    }
    return finish(status);
}

// automatic tuning before the first step of a handle whose step streams >= 256 MiB.  BRIE_PLACEMENT_TRIES = n (1 = off,
// default 8: the first set + up to seven candidates in rounds of three); BRIE_PLACEMENT_GOOD_GBS = the rate at which no
// further set is tried.  Default 6050: over 26 handles of
// configs[1] / configs[2] in four processes (profiles/r4a_placement_c{2,3}.jsonl) the probe -- whose rate equals the step
// kernel's to 1 %, correlation 0.999 at configs[2] -- read 4.93 - 5.30 TB/s in the slow mode, 5.49 - 5.94 in between and
// 6.0 - 6.24 in the fast one; or 0.97 x the best rate a handle of this process has reached on the device if higher.
constexpr double kPlacementGoodGBs = 6050.0;
// ... and a first set below 6150 still buys ONE round of candidates (call r5q: the bench handle stopped on a first set at 6.11
// TB/s, `frac` 0.855, where the sets kept by 30 other processes of the round read 6.16 - 6.33, 0.854 - 0.874)
constexpr double kPlacementStopAtOnceGBs = 6150.0;
// ... and for handles whose arrays are below a gigabyte 5850: the search used to run all four sets for them every time
// (6050 was out of reach) -- 14 - 44 ms of a 0.6-s configs[1] fit even when the first set was already in the fast mode
constexpr double kPlacementGoodSmallGBs = 5850.0;
int auto_placement(brie_handle *h, int mc_size) {
    if (h->placement_done) return BRIE_OK;
    h->placement_done = true;
    // sets in all: the caller's (brie_placement_configure), else BRIE_PLACEMENT_TRIES, else ONE round for arrays of a gigabyte
    // and more -- the first set + three candidates: up to 3 x the streamed bytes held for the search; every one of
    // round 5's 42 fresh processes ended `good` inside its first round -- and all eight for smaller arrays, whose candidates
    // come one at a time and cost 2.5 GB and 3 ms each at configs[1] (where most sets of a process read ~5.0 TB/s and one or
    // two of eight 5.6 - 6.0: calls r6a, r6b)
    static const int env_tries = [] { const char *e = getenv("BRIE_PLACEMENT_TRIES"); return e ? atoi(e) : 0; }();
    const int cls = placement_class(h);
    const int tries = h->placement_cfg_sets > 0 ? h->placement_cfg_sets
                                                : (env_tries > 0 ? env_tries : (cls == 1 ? static_cast<int>(BRIE_PLACEMENT_MAX_SETS) : 4));
    if (tries <= 1 || brie_step_storage_bytes(h) < (int64_t(256) << 20)) { h->placement_status = BRIE_PLACEMENT_OFF; return BRIE_OK; }
    static const double good_env = [] { const char *e = getenv("BRIE_PLACEMENT_GOOD_GBS"); return e ? atof(e) : 0.0; }();
    double good = good_env > 0.0 ? good_env : (cls == 1 ? kPlacementGoodSmallGBs : kPlacementGoodGBs);
    if (good_env <= 0.0) {
        std::lock_guard<std::mutex> l(g_place_mu);
        good = std::max(good, 0.97 * g_place_best[h->p.device & 63][cls]);
    }
    const double at_once = (good_env > 0.0 || cls == 1) ? good : std::max(good, kPlacementStopAtOnceGBs);
    return tune_placement(h, tries, at_once, mc_size > 1, good);     // the step runs two workgroups per CU for MC_size > 1 (brie_inst.hip)
}

// split = 0: n_steps complete steps.  split = 1 (n_steps == 1): everything up to the reduced per-cell
// statistics `rowstat`, which a gene-sharded coupled fit all-reduces across ranks before brie_step_end.
int run_steps(brie_handle *h, int32_t n_steps, float lr, int32_t mc_size, float *loss_trace, int split) {
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if (n_steps < 0 || mc_size < 1) return fail(BRIE_ERR_INVALID, "n_steps=%d mc_size=%d", n_steps, mc_size);
    if (h->step_open) return fail(BRIE_ERR_STATE, "brie_step_begin without brie_step_end");
    const bool lib_reduce = h->coupled && h->p.sharded != 0 && h->comm != nullptr;
    if (split == 0 && h->coupled && h->p.sharded != 0 && !lib_reduce)
        return fail(BRIE_ERR_STATE, "this handle is one gene shard of a coupled fit: attach a communicator "
                    "(brie_attach_comm) or use brie_step_begin / all-reduce brie_rowstat_buffer / brie_step_end");
    if (n_steps == 0) return BRIE_OK;
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    if ((rc = io_wait(h)) != BRIE_OK) return rc;
    if ((rc = ensure_partials(h)) != BRIE_OK) return rc;
    if ((rc = try_compact_counts(h)) != BRIE_OK) return rc;
    if ((rc = auto_placement(h, mc_size)) != BRIE_OK) return rc;
    const size_t lp_need = static_cast<size_t>(n_steps) * h->fin_blocks * 2;
    if (lp_need > h->loss_parts_elems) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->loss_parts) HIP_TRY(hipFree(h->loss_parts));
        h->loss_parts = nullptr;
        HIP_TRY(dev_alloc(&h->loss_parts, lp_need * sizeof(double)));
        h->loss_parts_elems = lp_need;
    }
    if (h->profiling) {
        while (h->ev_pool.size() < h->ev_used + 2 * static_cast<size_t>(n_steps)) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            h->ev_pool.push_back(ev);
        }
    }

    const bool u8 = h->cs != brie::kCountF32;
    brie::StepPointers q{};
    q.c1 = u8 ? static_cast<const void *>(h->cu[0]) : h->c[0];
    q.c2 = u8 ? static_cast<const void *>(h->cu[1]) : h->c[1];
    q.c3 = u8 ? static_cast<const void *>(h->cu[2]) : h->c[2];
    q.mu = h->mu; q.rho = h->rho; q.m_mu = h->m_mu; q.v_mu = h->v_mu; q.m_rho = h->m_rho; q.v_rho = h->v_rho;
    q.Xc = h->Xc; q.W = h->W; q.b = h->b; q.lam = h->lam; q.effL = h->effL; q.partials = h->partials;
    brie::StepScalars a{};
    a.ld = h->ld; a.row_stride = h->row_stride; a.gb_stride = h->gb_stride;
    a.Nc = static_cast<int32_t>(h->p.Nc); a.Ng = static_cast<int32_t>(h->p.Ng);
    a.rows_per_chunk = h->rows_per_chunk; a.mc = mc_size; a.inv_mc = 1.0f / static_cast<float>(mc_size);
    a.seed_lo = static_cast<uint32_t>(h->p.seed & 0xFFFFFFFFull); a.seed_hi = static_cast<uint32_t>(h->p.seed >> 32);
    a.quad_offset = static_cast<uint32_t>(h->p.gene_offset / 4);
    a.kc_wide = (h->wide_like && !h->vwide) ? h->p.Kc : 0;
    a.mean_in_rbuf = h->vwide ? 1 : 0;
    a.pc = h->pc;
    a.gene_active = h->gene_active; a.block_active = h->block_active; a.quad_ids = h->quad_ids;
    a.tt = h->tt;                    // tiers per gene quad (kCountMixed), one launch
    // wide designs: the MFMA tile kernel (ELBO target), else the LDS-broadcast variants + residual buffer
    const bool use_tile = h->tile && h->target == 0;
    if (h->wide_like && h->p.Kc > 0) {
        const size_t chunks = static_cast<size_t>(std::max(use_tile ? h->n_chunks : 0, h->n_gchunks));
        if ((rc = ensure_f32(&h->Gpart, &h->gpart_elems, chunks * h->p.Kc * h->ld, h->stream)) != BRIE_OK) return rc;
    }
    if (h->wide_like && !use_tile &&
        (rc = ensure_f32(&h->Rbuf, &h->rbuf_elems, static_cast<size_t>(h->p.Nc) * h->ld, h->stream)) != BRIE_OK)
        return rc;
    brie::LaunchCfg cfg{h->mode, h->cs, dim3(h->gene_blocks, h->n_chunks), h->stream, h->coupled ? 1 : 0};
    cfg.rbuf = h->Rbuf;
    brie::TileArgs ta{};
    ta.Xc = h->Xc; ta.W = h->W; ta.Xg = h->Xg; ta.Wg = h->Wg; ta.cb = h->cb; ta.clam = h->clam;
    ta.row_partials = h->row_partials; ta.Gpart = h->Gpart;
    ta.Kc = h->p.Kc; ta.Kg = h->p.Kg; ta.kgp = h->coupled ? h->kgp : 0; ta.cell_mode = h->cell_mode ? 1 : 0;
    // target="marginLik": uncoupled models with Kc <= 8 have their own light kernel, the others the MARGIN variants
    const bool simple_margin = h->target == 1 && !h->coupled && !h->wide_like;
    cfg.margin = (h->target == 1 && !simple_margin) ? 1 : 0;
    brie::CoupledArgs cp{};
    cp.Xg = h->Xg; cp.Wg = h->Wg; cp.cb = h->cb; cp.clam = h->clam; cp.row_partials = h->row_partials;
    cp.Kg = h->p.Kg; cp.cell_mode = h->cell_mode ? 1 : 0; cp.kgp = h->kgp;
    if (h->vgwide) {                 // Kg > 64: the gene design is in the prior mean the step reads and in gene_design_grad;
        cp.Xg = h->Xg_zero; cp.Wg = h->Wg_zero; cp.Kg = 0; cp.kgp = h->part_kgp;   // the step keeps the two per-cell sums
    }
    // Xg tile (kgp x 256 fp32), re-used by the cross-wave fold of the (Kc + 4) per-gene statistics
    cfg.gw_lds_bytes = (h->gwide && !h->vgwide) ? std::max(h->kgp, (brie::kWavesPerBlock - 1) * h->S) * brie::kGenesPerBlock *
                                                      static_cast<int>(sizeof(float)) : 0;
    brie::CellFinalizeArgs cf{};
    cf.row_partials = h->row_partials; cf.rowstat = h->rowstat; cf.Wg = h->Wg; cf.m_Wg = h->m_Wg; cf.v_Wg = h->v_Wg;
    cf.cb = h->cb; cf.m_cb = h->m_cb; cf.v_cb = h->v_cb; cf.clam = h->clam; cf.m_clam = h->m_clam; cf.v_clam = h->v_clam;
    cf.Nc = static_cast<int32_t>(h->p.Nc); cf.gene_blocks = h->gene_blocks; cf.Kg = h->p.Kg;
    cf.cell_mode = cp.cell_mode; cf.train_b = h->p.train_intercept; cf.train_lam = h->p.train_sigma;
    cf.phase = split ? 1 : 0;
    cf.kgp = h->kgp;
    cf.part_kgp = h->part_kgp;
    if (h->rowstat_ext) cf.rowstat = h->rowstat_ext;

    brie::FinalizeArgs f{};
    f.partials = h->partials; f.W = h->W; f.m_W = h->m_W; f.v_W = h->v_W; f.b = h->b; f.m_b = h->m_b; f.v_b = h->v_b;
    f.lam = h->lam; f.m_lam = h->m_lam; f.v_lam = h->v_lam; f.ld = h->ld;
    f.gene_active = h->gene_active; f.ring_kl = h->ring_kl; f.ring_ll = h->ring_ll;
    // gene POSITIONS the finalize walks: up to the end of the last real quad.  With the active quads packed to the front
    // a full quad can land in the last position, whose genes 1..3 sit at positions >= Ng when Ng % 4 != 0 -- bounded by
    // Ng, their carried losses were dropped from the trace and the loss ring (soak sequence 61 of call r3c).  The padding
    // genes this adds are frozen by construction (mask 0) and carry zeros.
    f.Ng = (a.Ng + brie::kVec - 1) / brie::kVec * brie::kVec; f.Kc = h->kernel_kc; f.n_chunks = h->n_chunks;
    f.train_b = h->cell_mode ? 0 : h->p.train_intercept;        // cell mode: the (1,Ng) vectors are not parameters
    f.train_lam = h->cell_mode ? 0 : h->p.train_sigma;

    auto adam_alpha = [lr](int64_t t) {
        const double tt = static_cast<double>(t);
        return static_cast<float>(static_cast<double>(lr) * std::sqrt(1.0 - std::pow(0.999, tt)) / (1.0 - std::pow(0.9, tt)));
    };
    // Small inputs: the n_steps steps as ONE launch (PERSIST variant; state, loss trace and loss ring as the loop below leaves
    // them, bit for bit).  Allowed for an uncoupled model with Kc <= 8, ELBO target, no frozen gene, MC_size 1 or 3 and a grid the
    // device holds at once; AUTOMATIC only where it was measured faster than two launches per step (calls r6k, r6l): at most 16
    // cell chunks (every workgroup of a gene block re-reads all of the block's chunk rows: the cost grows with their square)
    // and at most a fifth of the CUs busy (call r7d: 48 workgroups 9.1 against 11.1 us, 64: 11.4 / 12.2 at MC_size 1 but 14.4 / 13.9
    // at MC_size 3, 104: a tie, 140 and more: slower) -- configs[0]: 8.0 against 11.3 us per step; 300 x 2000: 15.7 against 12.3, so not there.
    // With at most 8 gene blocks the launcher gives every gene block an XCD of its own (persist_launch); there one launch also wins
    // with up to 8 x 16 workgroups while Kc <= 3 (calls r8r, r8s: 200 x 2000 8.4 - 10.9 against 10.6 - 12.3 us, MC_size 3 10.4 - 13.5
    // against 12.1 - 14.0; from Kc = 4 a tie at MC_size 3, from Kc = 6 a loss).
    int i_start = 0;
    {
        static const int env_mode = [] { const char *e = getenv("BRIE_FUSE_STEPS"); return e ? atoi(e) : -1; }();
        const int mode = h->persist_mode >= 0 ? h->persist_mode : env_mode;
        const bool can = !h->coupled && !h->wide_like && h->target == 0 && !split && !h->profiling && !h->any_frozen && !h->packed &&
                         (mc_size == 1 || mc_size == 3) && n_steps >= 2;
        const bool want = mode == 1 || (mode < 0 && h->n_chunks <= 16 &&
                                        (static_cast<int64_t>(h->n_chunks) * h->gene_blocks * 5 <= device_cus(h->p.device) ||
                                         (h->gene_blocks <= 8 && h->kernel_kc <= 3)));
        if (can && want) {
            const size_t pneed = static_cast<size_t>(h->n_chunks) * h->S * h->ld;
            if ((rc = ensure_f32(&h->partials2, &h->partials2_elems, pneed, h->stream)) != BRIE_OK) return rc;
            if ((rc = ensure_f32(&h->persist_alphas, &h->persist_alphas_elems, static_cast<size_t>(n_steps), h->stream)) != BRIE_OK) return rc;
            if (!h->persist_args) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->persist_args), sizeof(brie::PersistArgs)));
            if (!h->persist_barrier) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->persist_barrier), sizeof(uint32_t) * (h->gene_blocks + 1)));
            if (!h->persist_flag_host) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->persist_flag_host), sizeof(uint32_t), hipHostMallocDefault));
                *h->persist_flag_host = 0;
            }
            // host sources of the two small uploads live in the handle and are guarded by an event: the call only ENQUEUES (as
            // brie_step promises with loss_trace == NULL); it waits -- for the copies of the PREVIOUS fused call, long done --
            // before it overwrites them
            if (!h->persist_copy_event) HIP_TRY(hipEventCreateWithFlags(&h->persist_copy_event, hipEventDisableTiming));
            else HIP_TRY(hipEventSynchronize(h->persist_copy_event));
            std::vector<float> &alphas = h->persist_alphas_host;
            alphas.resize(static_cast<size_t>(n_steps));
            for (int i = 0; i < n_steps; ++i) alphas[i] = adam_alpha(h->t + 1 + i);
            brie::PersistArgs &pa = h->persist_args_host;
            pa = brie::PersistArgs{};
            pa.alphas = h->persist_alphas; pa.barrier = h->persist_barrier; pa.partials2 = h->partials2;
            pa.W = h->W; pa.m_W = h->m_W; pa.v_W = h->v_W; pa.b = h->b; pa.m_b = h->m_b; pa.v_b = h->v_b;
            pa.lam = h->lam; pa.m_lam = h->m_lam; pa.v_lam = h->v_lam; pa.ring_kl = h->ring_kl; pa.ring_ll = h->ring_ll;
            pa.loss_parts = h->loss_parts; pa.n_steps = n_steps;
            pa.ring_pos0 = static_cast<int32_t>(h->ring_pos % brie::kLossRing);
            pa.train_b = f.train_b; pa.train_lam = f.train_lam; pa.fin_Ng = f.Ng; pa.gene_blocks = h->gene_blocks;
            // One XCD per gene block where there are at most 8 and a column's workgroups fit one XCD's CUs: workgroups are dealt round
            // robin over the 8 XCDs by linear id, so with a grid of (8, chunks) column x runs on XCD x (HW_REG_XCC_ID:
            // profiles/micro/xcd_local_barrier.hip); columns without a gene block exit at once.  The hand-off protocol does not
            // depend on it (it is correct across XCDs); inside one XCD it is shorter (call r8q).  Handles and processes start at
            // different columns, so that small fits running side by side do not all queue for XCD 0.
            const int n_cus = device_cus(h->p.device);
            brie::LaunchCfg pcfg = cfg;
            if (h->gene_blocks <= 8 && h->n_chunks <= n_cus / 8) {
                static std::atomic<int> serial{0};
                if (h->persist_serial < 0) h->persist_serial = serial.fetch_add(1);
                pa.columns = 8;
                pa.col_offset = static_cast<int32_t>((static_cast<int64_t>(h->persist_serial) + static_cast<int64_t>(getpid())) * h->gene_blocks % 8);
                pcfg.grid.x = 8;
                pcfg.persist_columns = 8;
            }
            pa.debug = h->persist_debug;          // brie_debug_step_fusion (tests / experiments): an explicit call, no variable
            HIP_TRY(hipMemcpyAsync(h->persist_alphas, alphas.data(), alphas.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->persist_args, &pa, sizeof(pa), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemsetAsync(h->persist_barrier, 0, sizeof(uint32_t) * (h->gene_blocks + 1), h->stream));
            HIP_TRY(hipEventRecord(h->persist_copy_event, h->stream));
            a.alpha = alphas[0];
            a.draw = h->draw;
            if (h->fin_blocks == h->gene_blocks && launch_step_persist(h, pcfg, q, a, h->persist_args, n_cus) == 1) {
                h->t += n_steps;
                h->draw += static_cast<uint32_t>(n_steps);
                h->ring_pos += n_steps;
                h->persist_launches += 1;
                h->persist_steps += n_steps;
                i_start = n_steps;
                // the time-out word travels back behind the launch; check_ready of the next call that finds it set fails loudly
                HIP_TRY(hipMemcpyAsync(h->persist_flag_host, h->persist_barrier + h->gene_blocks, sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, h->stream));
            }
        }
    }
    for (int i = i_start; i < n_steps; ++i) {
        h->t += 1;
        const float alpha = adam_alpha(h->t);
        a.alpha = alpha; f.alpha = alpha; cf.alpha = alpha;
        a.draw = h->draw++;
        f.loss_parts = h->loss_parts + static_cast<size_t>(i) * h->fin_blocks * 2;
        f.ring_slot = static_cast<int32_t>(h->ring_pos % brie::kLossRing);
        f.ring_prev = static_cast<int32_t>((h->ring_pos + brie::kLossRing - 1) % brie::kLossRing);
        h->ring_pos += 1;
        if (h->profiling) HIP_TRY(hipEventRecord(h->ev_pool[h->ev_used++], h->stream));
        if (h->vwide) launch_xw_panels(h, h->Rbuf);   // Kc > 64: the prior mean's Xc . Wc_loc, panel by panel (timed with the step)
        if (h->vgwide) launch_gw_panels(h, h->Rbuf, h->p.Kc > 0);               // Kg > 64: + Wg_loc . Xg^T
        if (use_tile) launch_tile(h, cfg, q, a, ta);
        else if (simple_margin) launch_margin(h, cfg, q, a);
        else launch_step(h, cfg, q, a, cp);
        if (h->profiling) HIP_TRY(hipEventRecord(h->ev_pool[h->ev_used++], h->stream));
        hipLaunchKernelGGL(brie::gene_finalize, dim3(h->fin_blocks, h->S), dim3(brie::kBlock), 0, h->stream, f);
        if (use_tile && h->p.Kc > 0) {            // G = Xc^T . r was reduced inside the pass: Adam on Wc_loc
            const int64_t nW = static_cast<int64_t>(h->p.Kc) * h->ld;
            hipLaunchKernelGGL(brie::wide_w_adam, dim3(grid_1d(nW)), dim3(256), 0, h->stream, h->W, h->m_W, h->v_W, h->Gpart, nW,
                               h->n_chunks, alpha, h->gene_active, h->ld);
        } else if (h->wide_like && h->p.Kc > 0 && (rc = wide_backward(h, alpha)) != BRIE_OK)
            return rc;                            // residual buffer -> G = Xc^T . r (MFMA kernel), Adam on Wc_loc
        if (h->vgwide) {                          // residual buffer -> the Wg_loc gradient r . Xg (MFMA), panel by panel, into rowstat
            const dim3 ggrid(static_cast<unsigned>((h->p.Nc + 31) / 32));
            for (int k0 = 0; k0 < h->kgp; k0 += BRIE_MAX_KG_WIDE) {     // one launch per 64 gene features (4 accumulator sets per
                const int kp = std::min(h->kgp - k0, BRIE_MAX_KG_WIDE);    // wave need > 256 registers with this kernel's unrolling)
#define BRIE_GDG(NACC)                                                                                                  \
    hipLaunchKernelGGL((brie::gene_design_grad<NACC>), ggrid, dim3(brie::kBlock), 0, h->stream, h->XgT, h->Rbuf,       \
                       cf.rowstat, static_cast<int>(h->p.Nc), static_cast<int>(h->p.Ng), h->gene_blocks, kp, k0,       \
                       h->kgp, h->row_stride, h->gb_stride)
                if (kp <= 32) BRIE_GDG(1);
                else BRIE_GDG(2);
#undef BRIE_GDG
            }
        }
        if (h->coupled && lib_reduce && !split) {
            // gene shard of a coupled fit: local sums -> RCCL all-reduce on this stream -> Adam, all enqueued
            brie::CellFinalizeArgs c1 = cf, c2 = cf;
            c1.phase = 1; c2.phase = 2;
            hipLaunchKernelGGL(brie::cell_finalize, dim3(cell_finalize_blocks(h)), dim3(brie::kBlock), 0, h->stream, c1);
            if ((rc = brie::comm_allreduce_sum_f32_async(h->comm, cf.rowstat, static_cast<int64_t>(h->kgp + 2) * h->p.Nc,
                                                         h->stream)) != BRIE_OK) return rc;
            hipLaunchKernelGGL(brie::cell_finalize, dim3(cell_finalize_blocks(h)), dim3(brie::kBlock), 0, h->stream, c2);
        } else if (h->coupled)
            hipLaunchKernelGGL(brie::cell_finalize, dim3(cell_finalize_blocks(h)), dim3(brie::kBlock), 0, h->stream, cf);
    }
    HIP_TRY(hipGetLastError());
    if (split) {
        h->step_open = true;
        h->pending_cf = cf;
        return BRIE_OK;
    }
    if (loss_trace) {
        std::vector<double> parts(lp_need);
        HIP_TRY(hipMemcpyAsync(parts.data(), h->loss_parts, lp_need * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int i = 0; i < n_steps; ++i) {
            double kl = 0.0, ll = 0.0;
            for (int bk = 0; bk < h->fin_blocks; ++bk) {
                kl += parts[(static_cast<size_t>(i) * h->fin_blocks + bk) * 2 + 0];
                ll += parts[(static_cast<size_t>(i) * h->fin_blocks + bk) * 2 + 1];
            }
            loss_trace[i] = static_cast<float>(kl - ll);      // sum KL - sum ll (model_TFProb.py:208-211)
        }
    }
    return BRIE_OK;
}

}  // namespace

extern "C" {

int brie_step(brie_handle *h, int32_t n_steps, float lr, int32_t mc_size, float *loss_trace) {
    return run_steps(h, n_steps, lr, mc_size, loss_trace, 0);
}

int brie_step_begin(brie_handle *h, float lr, int32_t mc_size) { return run_steps(h, 1, lr, mc_size, nullptr, 1); }

int brie_attach_comm(brie_handle *h, brie_comm *c) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (h->step_open) return fail(BRIE_ERR_STATE, "a step is open");
    if (c && brie::comm_device(c) != h->p.device)
        return fail(BRIE_ERR_INVALID, "communicator lives on device %d, the handle on device %d", brie::comm_device(c),
                    h->p.device);
    h->comm = c;
    return BRIE_OK;
}

int brie_rowstat_buffer(brie_handle *h, float **dev, int64_t *n_floats) {
    if (!h || !dev || !n_floats) return fail(BRIE_ERR_INVALID, "null argument");
    if (!h->coupled) return fail(BRIE_ERR_STATE, "no per-cell statistics: Kg == 0 and intercept_mode 'gene'");
    *dev = h->rowstat_ext ? h->rowstat_ext : h->rowstat;
    *n_floats = static_cast<int64_t>(h->kgp + 2) * h->p.Nc;
    return BRIE_OK;
}

int brie_set_rowstat_buffer(brie_handle *h, float *dev) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (!h->coupled) return fail(BRIE_ERR_STATE, "no per-cell statistics: Kg == 0 and intercept_mode 'gene'");
    if (h->step_open) return fail(BRIE_ERR_STATE, "a step is open");
    h->rowstat_ext = dev;
    return BRIE_OK;
}

int brie_step_end(brie_handle *h, float *loss) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (!h->step_open) return fail(BRIE_ERR_STATE, "brie_step_end without brie_step_begin");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    h->step_open = false;
    if (h->coupled) {
        brie::CellFinalizeArgs cf = h->pending_cf;
        cf.phase = 2;
        if (h->comm && h->p.sharded != 0 &&
            (rc = brie::comm_allreduce_sum_f32_async(h->comm, cf.rowstat, static_cast<int64_t>(h->kgp + 2) * h->p.Nc,
                                                     h->stream)) != BRIE_OK)
            return rc;
        hipLaunchKernelGGL(brie::cell_finalize, dim3(cell_finalize_blocks(h)), dim3(brie::kBlock), 0, h->stream, cf);
        HIP_TRY(hipGetLastError());
    }
    if (loss) {
        std::vector<double> parts(static_cast<size_t>(h->fin_blocks) * 2);
        HIP_TRY(hipMemcpyAsync(parts.data(), h->loss_parts, parts.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        double kl = 0.0, ll = 0.0;
        for (int bk = 0; bk < h->fin_blocks; ++bk) { kl += parts[2 * bk]; ll += parts[2 * bk + 1]; }
        *loss = static_cast<float>(kl - ll);
    }
    return BRIE_OK;
}

int brie_loss_gene(brie_handle *h, int32_t n_repeats, float *out) {
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if (n_repeats < 1 || !out) return fail(BRIE_ERR_INVALID, "n_repeats=%d out=%p", n_repeats, (void *)out);
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    // This pass only READS the state, so it runs next to a pending asynchronous export (brie_read_results_async reads
    // mu / rho, writes its own slabs) -- that overlap is the point of the export.  Only a packed gene order has to be
    // undone first, and that moves the state: then the export is waited for (ensure_identity does).
    if (h->packed && (rc = ensure_identity(h)) != BRIE_OK) return rc;
    if ((rc = ensure_partials(h)) != BRIE_OK) return rc;
    if ((rc = try_compact_counts(h)) != BRIE_OK) return rc;
    const bool u8 = h->cs != brie::kCountF32;
    brie::LossGeneArgs a{};
    a.c1 = u8 ? static_cast<const void *>(h->cu[0]) : h->c[0];
    a.c2 = u8 ? static_cast<const void *>(h->cu[1]) : h->c[1];
    a.c3 = u8 ? static_cast<const void *>(h->cu[2]) : h->c[2];
    a.pc = h->pc;
    a.quad_ids = h->quad_ids;
    a.coupled = h->coupled ? 1 : 0;
    a.margin = h->target == 1 ? 1 : 0;
    a.mbuf = nullptr;
    const bool xw_in_mbuf = h->wide_like && h->p.Kc > 0;
    if (xw_in_mbuf) {
        if ((rc = wide_forward_mean(h)) != BRIE_OK) return rc;
        a.mbuf = h->Mbuf;
    }
    if (h->gwide) {          // after the wide cell design's Xc . Wc_loc, if any (same buffer)
        if ((rc = gwide_forward_mean(h, xw_in_mbuf)) != BRIE_OK) return rc;
        a.mbuf = h->Mbuf;
    }
    a.cp.Xg = h->Xg; a.cp.Wg = h->Wg; a.cp.cb = h->cb; a.cp.clam = h->clam; a.cp.row_partials = nullptr;
    a.cp.Kg = h->p.Kg; a.cp.cell_mode = h->cell_mode ? 1 : 0; a.cp.kgp = h->kgp;
    a.mu = h->mu; a.rho = h->rho; a.Xc = h->Xc; a.W = h->W; a.b = h->b;
    a.lam = h->lam; a.effL = h->effL; a.partials = h->partials; a.ld = h->ld;
    a.row_stride = h->row_stride; a.gb_stride = h->gb_stride;
    a.Nc = static_cast<int32_t>(h->p.Nc); a.Ng = static_cast<int32_t>(h->p.Ng);
    a.rows_per_chunk = h->rows_per_chunk; a.n_rep = n_repeats;
    a.seed_lo = static_cast<uint32_t>(h->p.seed & 0xFFFFFFFFull); a.seed_hi = static_cast<uint32_t>(h->p.seed >> 32);
    a.draw0 = h->draw; a.quad_offset = static_cast<uint32_t>(h->p.gene_offset / 4);
    h->draw += static_cast<uint32_t>(n_repeats);
    a.tt = h->tt;
    launch_loss_gene(h, brie::LaunchCfg{h->mode, h->cs, dim3(h->gene_blocks, h->n_chunks), h->stream, h->coupled ? 1 : 0}, a);
    hipLaunchKernelGGL(brie::loss_gene_reduce, dim3(h->fin_blocks), dim3(brie::kBlock), 0, h->stream, h->partials,
                       h->gene_tmp, h->ld, a.Ng, h->n_chunks, 1.0f / static_cast<float>(n_repeats));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->gene_tmp, h->p.Ng * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BRIE_OK;
}

int brie_read(brie_handle *h, int which, float *dst, int64_t rows, int64_t cols, int64_t ld) {
    if (!h || !dst) return fail(BRIE_ERR_INVALID, "null argument");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    if (ld < cols) return fail(BRIE_ERR_INVALID, "ld=%lld < cols=%lld", (long long)ld, (long long)cols);
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    if (which == BRIE_PSI || which == BRIE_Z_STD || which == BRIE_PSI95CI || which == BRIE_Z_LOC ||
        which == BRIE_Z_STD_LOG) {
        if (rows != Nc || cols != Ng)
            return fail(BRIE_ERR_INVALID, "array %d must be (%lld, %lld)", which, (long long)Nc, (long long)Ng);
        // kernel writes the array row-major and contiguous, then ONE copy crosses PCIe
        const int mode = which == BRIE_Z_LOC ? 3 : (which == BRIE_Z_STD_LOG ? 4 : which - BRIE_PSI);
        float *tmp = nullptr;
        if ((rc = io_buffer(h, &tmp)) != BRIE_OK) return rc;
        hipLaunchKernelGGL(brie::export_rowmajor, dim3(grid_1d(static_cast<int64_t>(h->gene_blocks) * Nc * brie::kWave)),
                           dim3(256), 0, h->stream, h->mu, h->rho, tmp, static_cast<int>(Nc), static_cast<int>(Ng),
                           h->gene_blocks, h->row_stride, h->gb_stride, mode);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) {
            if (ld == cols)
                e = hipMemcpyAsync(dst, tmp, static_cast<size_t>(Nc) * Ng * sizeof(float), hipMemcpyDefault, h->stream);
            else
                e = hipMemcpy2DAsync(dst, ld * sizeof(float), tmp, Ng * sizeof(float), Ng * sizeof(float), Nc,
                                     hipMemcpyDefault, h->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return fail(BRIE_ERR_HIP, "read array %d: %s", which, hipGetErrorString(e));
        return BRIE_OK;
    }
    if (which == BRIE_SIGMA) {
        if (h->cell_mode) {                 // (Nc, 1): exp of the per-cell log sigma, via rowstat as scratch
            if (rows != Nc || cols != 1) return fail(BRIE_ERR_INVALID, "sigma must be (%lld, 1) in cell mode", (long long)Nc);
            hipLaunchKernelGGL(brie::exp_vec, dim3((Nc + 255) / 256), dim3(256), 0, h->stream, h->clam, h->rowstat,
                               static_cast<int>(Nc));
            HIP_TRY(hipMemcpy2DAsync(dst, ld * sizeof(float), h->rowstat, sizeof(float), sizeof(float), Nc,
                                     hipMemcpyDefault, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            return BRIE_OK;
        }
        if (rows != 1 || cols != Ng) return fail(BRIE_ERR_INVALID, "sigma must be (1, %lld)", (long long)Ng);
        hipLaunchKernelGGL(brie::exp_vec, dim3((Ng + 255) / 256), dim3(256), 0, h->stream, h->lam, h->gene_tmp,
                           static_cast<int>(Ng));
        HIP_TRY(hipMemcpyAsync(dst, h->gene_tmp, Ng * sizeof(float), hipMemcpyDefault, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return BRIE_OK;
    }
    if (which >= BRIE_COUNT1 && which <= BRIE_COUNT3 && h->cs != brie::kCountF32) {
        const int l = which - BRIE_COUNT1;
        if (l >= h->p.n_layers) return fail(BRIE_ERR_INVALID, "count layer %d outside n_layers=%d", l + 1, h->p.n_layers);
        if (rows != Nc || cols != Ng)
            return fail(BRIE_ERR_INVALID, "array %d is (%lld, %lld)", which, (long long)Nc, (long long)Ng);
        float *tmp = nullptr;
        if ((rc = io_buffer(h, &tmp)) != BRIE_OK) return rc;
        launch_expand(h, l, tmp, h->pc, l < 2 ? 1 : 0);
        rc = copy_cellgene(h, tmp, nullptr, dst, ld);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (rc != BRIE_OK) return rc;
        if (e != hipSuccess) return fail(BRIE_ERR_HIP, "read compact counts: %s", hipGetErrorString(e));
        return BRIE_OK;
    }
    float *dev = nullptr;
    int64_t R = 0, C = 0, ldd = 0;
    rc = matrix_target(h, which, &dev, &R, &C, &ldd);
    if (rc != BRIE_OK) return rc;
    if (rows != R || cols != C)
        return fail(BRIE_ERR_INVALID, "array %d is (%lld, %lld), asked for (%lld, %lld)", which, (long long)R,
                    (long long)C, (long long)rows, (long long)cols);
    if (R * C > 0) {
        if (which == BRIE_Z_LOC || which == BRIE_Z_STD_LOG || which <= BRIE_COUNT3) {
            if ((rc = copy_cellgene(h, dev, nullptr, dst, ld)) != BRIE_OK) return rc;
        } else {
            HIP_TRY(hipMemcpy2DAsync(dst, ld * sizeof(float), dev, ldd * sizeof(float), C * sizeof(float), R,
                                     hipMemcpyDefault, h->stream));
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return BRIE_OK;
}

}  // extern "C"

namespace {
// Per-entry Monte-Carlo terms of the loss into row-major (Nc, Ng) DEVICE buffers: res_ll = mean over `size` samples
// z ~ q of the log-likelihood (target ELBO) or their log-mean-exp with z ~ prior (target marginLik); res_kl (may be
// null; ELBO only) = KL(q || prior).  by_draw: see brie::LogLikArgs.  An accessor, not the hot loop: compact count
// layers are expanded into fp32 temporaries, the prior mean goes through Mbuf.
int eval_elements(brie_handle *h, int32_t size, int by_draw, float *res_ll, float *res_kl) {
    int rc = BRIE_OK;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;
    if ((rc = try_compact_counts(h)) != BRIE_OK) return rc;
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    const size_t mat = static_cast<size_t>(Nc) * h->ld;
    // fp32 views of the count layers (pseudo-count included), expanded into temporaries when stored compactly
    float *tmp[3] = {nullptr, nullptr, nullptr};
    auto cleanup = [&]() { for (float *q : tmp) if (q) hipFree(q); };
    brie::LogLikArgs a{};
    const float *layers[3] = {nullptr, nullptr, nullptr};
    for (int l = 0; l < h->p.n_layers; ++l) {
        if (h->cs == brie::kCountF32) { layers[l] = h->c[l]; continue; }
        if (dev_alloc(&tmp[l], mat * sizeof(float)) != hipSuccess) {
            cleanup();
            return fail(BRIE_ERR_HIP, "hipMalloc count view");
        }
        launch_expand(h, l, tmp[l], h->pc, l < 2 ? 1 : 0);
        layers[l] = tmp[l];
    }
    a.c1 = layers[0]; a.c2 = layers[1]; a.c3 = layers[2];
    a.mu = h->mu; a.rho = h->rho; a.b = h->b; a.lam = h->lam; a.cb = h->cb; a.clam = h->clam; a.effL = h->effL;
    a.prior_m = nullptr;
    a.margin = h->target == 1 ? 1 : 0;
    if ((a.margin || res_kl) && (h->p.Kc > 0 || h->p.Kg > 0)) {   // prior mean Xc.Wc_loc (+ Wg_loc.Xg^T) into Mbuf
        if (h->p.Kc > 0 && (rc = wide_forward_mean(h)) != BRIE_OK) { cleanup(); return rc; }
        if (h->p.Kg > 0 && (rc = gwide_forward_mean(h, h->p.Kc > 0)) != BRIE_OK) { cleanup(); return rc; }
        a.prior_m = h->Mbuf;
    }
    a.out = res_ll; a.kl_out = a.margin ? nullptr : res_kl;
    a.ld = h->ld; a.row_stride = h->row_stride; a.gb_stride = h->gb_stride;
    a.Nc = static_cast<int32_t>(Nc); a.Ng = static_cast<int32_t>(Ng); a.gene_blocks = h->gene_blocks;
    a.mode = h->mode; a.cell_mode = h->cell_mode ? 1 : 0; a.n_mc = size; a.by_draw = by_draw;
    a.seed_lo = static_cast<uint32_t>(h->p.seed & 0xFFFFFFFFull); a.seed_hi = static_cast<uint32_t>(h->p.seed >> 32);
    // by_draw == 0: ONE draw id; sample k of the `size` samples is keyed (draw, k) like the MC samples of one step -- no
    // (draw, k) pair of this call is ever produced again by a later step or loss_gene (those use k < MC_size of LATER
    // draw ids).  by_draw == 1: `size` draw ids, sample k at (draw + k, 0).
    a.draw = h->draw; a.quad_offset = static_cast<uint32_t>(h->p.gene_offset / 4);
    h->draw += by_draw ? static_cast<uint32_t>(size) : 1u;
    hipLaunchKernelGGL(brie::loglik_mc_export, dim3(grid_1d(static_cast<int64_t>(h->gene_blocks) * Nc * brie::kWave)),
                       dim3(256), 0, h->stream, a);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "per-entry loss terms: %s", hipGetErrorString(e));
    return BRIE_OK;
}
}  // namespace

extern "C" {

int brie_loglik_mc(brie_handle *h, int32_t size, float *out, int64_t ld) {
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if (size < 1 || !out) return fail(BRIE_ERR_INVALID, "size=%d out=%p", size, (void *)out);
    if (ld < h->p.Ng) return fail(BRIE_ERR_INVALID, "ld=%lld < Ng", (long long)ld);
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    float *res = nullptr;
    if (dev_alloc(&res, static_cast<size_t>(Nc) * Ng * sizeof(float)) != hipSuccess)
        return fail(BRIE_ERR_HIP, "hipMalloc result");
    rc = eval_elements(h, size, 0, res, nullptr);
    hipError_t e = hipSuccess;
    if (rc == BRIE_OK) {
        e = hipMemcpy2DAsync(out, ld * sizeof(float), res, Ng * sizeof(float), Ng * sizeof(float), Nc, hipMemcpyDefault,
                             h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    hipFree(res);
    if (rc != BRIE_OK) return rc;
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "logLik_MC: %s", hipGetErrorString(e));
    return BRIE_OK;
}

int brie_get_loss(brie_handle *h, int32_t mc_size, int32_t axis, float *out) {
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if (mc_size < 1 || !out) return fail(BRIE_ERR_INVALID, "mc_size=%d out=%p", mc_size, (void *)out);
    if (axis != 0 && axis != 1) return fail(BRIE_ERR_INVALID, "axis=%d (0 = per gene, 1 = per cell)", axis);
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    const bool elbo = h->target == 0;
    const int rows_per_chunk = 256;
    const int n_chunks = static_cast<int>((Nc + rows_per_chunk - 1) / rows_per_chunk);
    float *ll = nullptr, *kl = nullptr, *res = nullptr;
    double *part = nullptr;
    auto cleanup = [&]() { hipFree(ll); hipFree(kl); hipFree(res); hipFree(part); };
    const size_t mat = static_cast<size_t>(Nc) * Ng * sizeof(float);
    hipError_t e = dev_alloc(&ll, mat);
    if (e == hipSuccess && elbo) e = dev_alloc(&kl, mat);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&res), static_cast<size_t>(axis == 0 ? Ng : Nc) * sizeof(float));
    if (e == hipSuccess && axis == 0)
        e = dev_alloc(&part, static_cast<size_t>(n_chunks) * 2 * Ng * sizeof(double));
    if (e != hipSuccess) { cleanup(); return fail(BRIE_ERR_HIP, "get_loss buffers: %s", hipGetErrorString(e)); }
    // ELBO: KL - mean_k ll(z_k), the samples at consecutive draw ids like brie_loss_gene(mc_size) (the loss is linear in
    // them); marginLik: -log-mean-exp over the samples of ONE draw id, like a step with this MC_size
    if ((rc = eval_elements(h, mc_size, elbo ? 1 : 0, ll, kl)) != BRIE_OK) { cleanup(); return rc; }
    if (axis == 0) {
        const dim3 grid(static_cast<unsigned>((Ng + 255) / 256), static_cast<unsigned>(n_chunks));
        hipLaunchKernelGGL(brie::loss_axis0_partial, grid, dim3(256), 0, h->stream, kl, ll, part, static_cast<int>(Nc),
                           static_cast<int>(Ng), rows_per_chunk);
        hipLaunchKernelGGL(brie::loss_axis0_final, dim3(grid.x), dim3(256), 0, h->stream, part, res, static_cast<int>(Ng),
                           n_chunks);
    } else {
        hipLaunchKernelGGL(brie::loss_axis1, dim3(static_cast<unsigned>(Nc)), dim3(brie::kBlock), 0, h->stream, kl, ll, res,
                           static_cast<int>(Nc), static_cast<int>(Ng));
    }
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, res, static_cast<size_t>(axis == 0 ? Ng : Nc) * sizeof(float), hipMemcpyDefault, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "get_loss: %s", hipGetErrorString(e));
    return BRIE_OK;
}

int brie_read_results_async(brie_handle *h, float *psi, float *z_std, float *psi95ci, float *z_loc, int64_t ld) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (!psi && !z_std && !psi95ci && !z_loc) return BRIE_OK;
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    if ((rc = ensure_identity(h)) != BRIE_OK) return rc;        // also waits for an earlier asynchronous read
    const int64_t Nc = h->p.Nc, Ng = h->p.Ng;
    if (ld < Ng) return fail(BRIE_ERR_INVALID, "ld=%lld < Ng=%lld", (long long)ld, (long long)Ng);
    if (!h->io_stream) {
        // highest priority: the short export kernels must get compute units as they free up, not after the 500-draw
        // loss_gene kernel on the main stream has dispatched its last workgroup (which serialised the two: 0.46 s of
        // loss_gene + 0.33 s of exports and copies instead of max of the two)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        const char *pe = getenv("BRIE_IO_STREAM_PRIORITY");         // "0": default priority (A/B runs)
        if (pe && pe[0] == '0') {
            HIP_TRY(hipStreamCreateWithFlags(&h->io_stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&h->io_stream2, hipStreamNonBlocking));
        } else {
            HIP_TRY(hipStreamCreateWithPriority(&h->io_stream, hipStreamNonBlocking, prio_hi));
            HIP_TRY(hipStreamCreateWithPriority(&h->io_stream2, hipStreamNonBlocking, prio_hi));
        }
        HIP_TRY(hipEventCreateWithFlags(&h->io_event, hipEventDisableTiming));
    }
    // slabs of ~64 M elements per output: the export kernel of a slab (microseconds to milliseconds) and its copies
    // are enqueued in order on the i/o stream, the copy engine streams while the main stream keeps computing
    const char *se = getenv("BRIE_IO_SLAB_ELEMS");
    const int64_t slab_target = se && atoll(se) > 0 ? atoll(se) : (int64_t(1) << 26);
    const int64_t slab_rows = std::max<int64_t>(1, std::min<int64_t>(Nc, slab_target / Ng));
    const size_t need = static_cast<size_t>(slab_rows) * Ng * 4 * 2;            // 4 outputs, double buffered
    if (need > h->io_slab_elems) {
        if (h->io_slab) HIP_TRY(hipFree(h->io_slab));
        h->io_slab = nullptr;
        h->io_slab_elems = 0;
        HIP_TRY(dev_alloc(&h->io_slab, need * sizeof(float)));
        h->io_slab_elems = need;
    }
    // the state is final once everything enqueued so far on the main stream has run
    HIP_TRY(hipEventRecord(h->io_event, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->io_stream, h->io_event, 0));
    HIP_TRY(hipStreamWaitEvent(h->io_stream2, h->io_event, 0));
    h->io_pending = true;
    h->io_rc = BRIE_OK;
    const int device = h->p.device;
    auto body = [h, Nc, Ng, ld, slab_rows, device, psi, z_std, psi95ci, z_loc]() {
        float *outs[4] = {psi, z_std, psi95ci, z_loc};
        auto bad = [h](const char *what, hipError_t e) {
            h->io_rc = BRIE_ERR_HIP;
            h->io_err = std::string(what) + ": " + hipGetErrorString(e);
            (void)hipGetLastError();
        };
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { bad("hipSetDevice", e); return; }
        // Slab k lives in buffer k & 1 and on stream k & 1.  The export kernel of slab k + 1 is enqueued BEFORE the copies
        // of slab k start: a copy into pageable memory blocks this thread, and next to the 500-draw loss_gene pass the
        // short export kernel waits milliseconds for compute units -- with kernel and copies of all slabs on ONE stream the
        // copy engine idled through every one of those waits (tail behind loss_gene 0.22 s; profiles/history/r3g_*).  Stream order
        // still protects the buffers: kernel k + 2 follows the copies of slab k on their common stream.
        const bool one_stream = getenv("BRIE_IO_ONE_STREAM") != nullptr;             // A/B runs: round 2's order
        const int64_t n_slabs = (Nc + slab_rows - 1) / slab_rows;
        auto slab_ptr = [&](int64_t k, int i) {
            return h->io_slab + static_cast<size_t>(k & 1) * slab_rows * Ng * 4 + static_cast<size_t>(i) * slab_rows * Ng;
        };
        auto stream_of = [&](int64_t k) { return (one_stream || (k & 1) == 0) ? h->io_stream : h->io_stream2; };
        auto launch = [&](int64_t k) {
            const int64_t r0 = k * slab_rows, rows = std::min(slab_rows, Nc - r0);
            brie::ExportSlabArgs a{};
            a.mu = h->mu; a.rho = h->rho;
            a.psi = outs[0] ? slab_ptr(k, 0) : nullptr; a.zstd = outs[1] ? slab_ptr(k, 1) : nullptr;
            a.ci = outs[2] ? slab_ptr(k, 2) : nullptr; a.zloc = outs[3] ? slab_ptr(k, 3) : nullptr;
            a.row_stride = h->row_stride; a.gb_stride = h->gb_stride;
            a.Ng = static_cast<int32_t>(Ng); a.gene_blocks = h->gene_blocks;
            a.r0 = static_cast<int32_t>(r0); a.rows = static_cast<int32_t>(rows);
            hipLaunchKernelGGL(brie::export_slab, dim3(grid_1d(static_cast<int64_t>(h->gene_blocks) * rows * brie::kWave)),
                               dim3(256), 0, stream_of(k), a);
            return hipGetLastError();
        };
        if ((e = launch(0)) != hipSuccess) { bad("export_slab", e); return; }
        for (int64_t k = 0; k < n_slabs; ++k) {
            if (!one_stream && k + 1 < n_slabs && (e = launch(k + 1)) != hipSuccess) { bad("export_slab", e); return; }
            const int64_t r0 = k * slab_rows, rows = std::min(slab_rows, Nc - r0);
            for (int i = 0; i < 4; ++i) {
                if (!outs[i]) continue;
                if (ld == Ng)
                    e = hipMemcpyAsync(outs[i] + r0 * ld, slab_ptr(k, i), static_cast<size_t>(rows) * Ng * sizeof(float),
                                       hipMemcpyDefault, stream_of(k));
                else
                    e = hipMemcpy2DAsync(outs[i] + r0 * ld, ld * sizeof(float), slab_ptr(k, i), Ng * sizeof(float),
                                         Ng * sizeof(float), rows, hipMemcpyDefault, stream_of(k));
                if (e != hipSuccess) { bad("copy", e); return; }
            }
            if (one_stream && k + 1 < n_slabs && (e = launch(k + 1)) != hipSuccess) { bad("export_slab", e); return; }
        }
    };
    try {
        h->io_thread = std::thread(body);
    } catch (...) {               // no thread to be had: export on the caller's thread (no exception may cross the C ABI)
        body();
    }
    return BRIE_OK;
}

int brie_read_wait(brie_handle *h) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    return io_wait(h);
}

// the host half of the staged ingest on its own (no GPU involved): tests and host-bandwidth measurements
int brie_host_convert_u16(const float *src, int64_t rows, int64_t cols, int64_t ld, uint16_t *dst, int32_t *not_integral) {
    if (!not_integral || rows < 0 || cols < 0 || ld < cols) return fail(BRIE_ERR_INVALID, "bad argument");
    *not_integral = 0;
    if (rows * cols == 0) return BRIE_OK;
    if (!src || !dst) return fail(BRIE_ERR_INVALID, "null argument");
    const int T = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), rows)));
    std::vector<uint32_t> bad(static_cast<size_t>(T), 0u);
    std::vector<std::thread> pool;
    auto part = [&](int t) {
        const int64_t a = rows * t / T, b = rows * (t + 1) / T;
        bad[static_cast<size_t>(t)] = convert_rows_u16(src + a * ld, ld, b - a, cols, dst + a * cols);
    };
    std::vector<int> inline_parts{0};
    for (int t = 1; t < T; ++t) {
        try { pool.emplace_back(part, t); } catch (...) { inline_parts.push_back(t); }
    }
    for (int t : inline_parts) part(t);
    for (std::thread &th : pool) th.join();
    uint32_t any = 0;
    for (uint32_t b : bad) any |= b;
    *not_integral = any ? 1 : 0;
    return BRIE_OK;
}

// one slab of the typed ingest on the host alone (tests): dst receives rows x cols u16 (*is_f32 = 0) or float32 (= 1)
int brie_host_convert_slab(const void *src, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, void *dst, int32_t *is_f32) {
    if (!is_f32 || rows < 0 || cols < 0 || ld < cols || dtype_size(dtype) == 0) return fail(BRIE_ERR_INVALID, "bad argument");
    *is_f32 = 0;
    if (rows * cols == 0) return BRIE_OK;
    if (!src || !dst) return fail(BRIE_ERR_INVALID, "null argument");
    *is_f32 = convert_slab(src, dtype, ld, rows, cols, dst);
    return BRIE_OK;
}

int brie_host_register(void *ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return fail(BRIE_ERR_INVALID, "bad argument");
    HIP_TRY(hipHostRegister(ptr, static_cast<size_t>(bytes), hipHostRegisterPortable));
    return BRIE_OK;
}

int brie_host_unregister(void *ptr) {
    if (!ptr) return fail(BRIE_ERR_INVALID, "null argument");
    HIP_TRY(hipHostUnregister(ptr));
    return BRIE_OK;
}

// measurement aid: device address of a cell x gene array (0..2 count layers as stored, 8 Z_loc, 9 Z_std_log,
// 20..23 the Adam moments m/v of Z_loc, m/v of Z_std_log)
int brie_debug_address(brie_handle *h, int which, uint64_t *addr) {
    if (!h || !addr) return fail(BRIE_ERR_INVALID, "null argument");
    const void *p = nullptr;
    switch (which) {
        case 0: case 1: case 2: p = h->cs == brie::kCountF32 ? static_cast<const void *>(h->c[which]) : h->cu[which]; break;
        case BRIE_Z_LOC: p = h->mu; break;
        case BRIE_Z_STD_LOG: p = h->rho; break;
        case 20: p = h->m_mu; break;
        case 21: p = h->v_mu; break;
        case 22: p = h->m_rho; break;
        case 23: p = h->v_rho; break;
        default: return fail(BRIE_ERR_INVALID, "array %d", which);
    }
    *addr = reinterpret_cast<uint64_t>(p);
    return BRIE_OK;
}

int brie_get_draw(brie_handle *h, uint32_t *draw) {
    if (!h || !draw) return fail(BRIE_ERR_INVALID, "null argument");
    *draw = h->draw;
    return BRIE_OK;
}
int brie_set_draw(brie_handle *h, uint32_t draw) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    h->draw = draw;
    return BRIE_OK;
}

int brie_synchronize(brie_handle *h) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BRIE_OK;
}

int brie_profile_enable(brie_handle *h, int32_t enable) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    h->profiling = enable != 0;
    h->ev_used = 0;
    h->prof_ms = 0.0;
    h->prof_launches = 0;
    return BRIE_OK;
}

int brie_profile_read(brie_handle *h, double *kernel_ms_total, int64_t *n_launches) {
    if (!h || !kernel_ms_total || !n_launches) return fail(BRIE_ERR_INVALID, "null argument");
    int rc = set_device(h);
    if (rc != BRIE_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[i], h->ev_pool[i + 1]));
        h->prof_ms += ms;
        h->prof_launches += 1;
    }
    h->ev_used = 0;
    *kernel_ms_total = h->prof_ms;
    *n_launches = h->prof_launches;
    return BRIE_OK;
}

int brie_placement_probe(brie_handle *h, int32_t iters, double *gbs) {
    if (!h || !gbs || iters < 1) return fail(BRIE_ERR_INVALID, "bad argument");
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    if ((rc = io_wait(h)) != BRIE_OK) return rc;
    if ((rc = try_compact_counts(h)) != BRIE_OK) return rc;
    return probe_rate(h, iters, gbs, false);
}

int brie_placement_tune(brie_handle *h, int32_t max_tries, double good_gbs) {
    if (!h || max_tries < 1) return fail(BRIE_ERR_INVALID, "bad argument");
    int rc = check_ready(h);
    if (rc != BRIE_OK) return rc;
    if (h->step_open) return fail(BRIE_ERR_STATE, "a step is open");
    if ((rc = set_device(h)) != BRIE_OK) return rc;
    if ((rc = io_wait(h)) != BRIE_OK) return rc;
    if ((rc = try_compact_counts(h)) != BRIE_OK) return rc;
    return tune_placement(h, max_tries, good_gbs, false);
}

int brie_placement_info(const brie_handle *h, int32_t *tries, int32_t *kept, double *gbs, int32_t n_gbs, double *seconds) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (gbs && n_gbs < 0) return fail(BRIE_ERR_INVALID, "n_gbs=%d", n_gbs);
    if (tries) *tries = h->placement_tries;
    if (kept) *kept = h->placement_kept;
    if (gbs) for (int i = 0; i < n_gbs; ++i) gbs[i] = i < BRIE_PLACEMENT_MAX_SETS ? h->placement_gbs[i] : 0.0;   // the caller's capacity
    if (seconds) *seconds = h->placement_seconds;
    return BRIE_OK;
}

int brie_set_step_fusion(brie_handle *h, int32_t mode) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (mode < -1 || mode > 1) return fail(BRIE_ERR_INVALID, "mode=%d (-1 automatic, 0 off, 1 on)", mode);
    h->persist_mode = mode;
    return BRIE_OK;
}

int brie_debug_step_fusion(brie_handle *h, int32_t flags) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (flags < 0) return fail(BRIE_ERR_INVALID, "flags=%d", flags);
    h->persist_debug = flags;
    return BRIE_OK;
}

int brie_step_fusion_info(const brie_handle *h, int64_t *launches, int64_t *steps) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (launches) *launches = h->persist_launches;
    if (steps) *steps = h->persist_steps;
    return BRIE_OK;
}

int brie_placement_configure(brie_handle *h, int32_t max_sets, double hbm_fraction, double max_seconds) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (max_sets > BRIE_PLACEMENT_MAX_SETS || hbm_fraction > 1.0)
        return fail(BRIE_ERR_INVALID, "max_sets=%d (at most %d) hbm_fraction=%g (at most 1)", max_sets, BRIE_PLACEMENT_MAX_SETS, hbm_fraction);
    h->placement_cfg_sets = max_sets > 0 ? max_sets : 0;
    h->placement_cfg_frac = hbm_fraction > 0.0 ? hbm_fraction : 0.0;
    h->placement_cfg_seconds = max_seconds > 0.0 ? max_seconds : 0.0;
    return BRIE_OK;
}

int brie_debug_inject_placement_failure(brie_handle *h, int32_t point) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (point < 0 || point > 4) return fail(BRIE_ERR_INVALID, "point=%d (0 = none, 1 .. 4)", point);
    h->placement_inject = point;
    return BRIE_OK;
}

int brie_placement_status(const brie_handle *h, int32_t *status, int64_t *peak_bytes, char *note, int32_t note_len) {
    if (!h) return fail(BRIE_ERR_INVALID, "null handle");
    if (status) *status = h->placement_status;
    if (peak_bytes) *peak_bytes = h->placement_peak_bytes;
    if (note && note_len > 0) snprintf(note, static_cast<size_t>(note_len), "%s", h->placement_note);
    return BRIE_OK;
}

// Experiment aid (no reference counterpart): ONE slab, the eight streamed arrays of a 2-layer u8 problem carved out of it at
// caller-given byte offsets, the placement probe timed for each of `n_layouts` offset sets -- does the RELATIVE spacing of the
// arrays decide the streaming rate, the physical memory being the same?  offsets: n_layouts x 8 (six state arrays, two u8
// count layers), each + its array inside slab_bytes.  gbs: n_layouts rates (storage bytes of a step / probe time).
int brie_probe_layouts(int32_t device, int64_t Nc, int64_t Ng, int64_t slab_bytes, int32_t n_layouts, const int64_t *offsets,
                       int32_t iters, double *gbs) {
    if (!offsets || !gbs || n_layouts < 1 || iters < 1 || Nc <= 0 || Ng <= 0) return fail(BRIE_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const int64_t ld = round_up(Ng, brie::kGenesPerBlock);
    const int64_t mat = Nc * ld * 4, cnt = Nc * ld;
    for (int i = 0; i < n_layouts * 8; ++i) {
        const int64_t need = offsets[i] + ((i % 8) < 6 ? mat : cnt);
        if (offsets[i] < 0 || offsets[i] % 16 != 0 || need > slab_bytes) return fail(BRIE_ERR_INVALID, "offset %d outside the slab", i);
    }
    char *slab = nullptr;
    HIP_TRY(dev_alloc(&slab, static_cast<size_t>(slab_bytes)));
    hipError_t e = hipMemset(slab, 0, static_cast<size_t>(slab_bytes));
    brie::StepScalars a{};
    a.ld = ld; a.row_stride = brie::kGenesPerBlock; a.gb_stride = Nc * brie::kGenesPerBlock;
    a.Nc = static_cast<int32_t>(Nc); a.Ng = static_cast<int32_t>(Ng);
    int rpc = 256;
    while (rpc > 16 && (Nc + rpc - 1) / rpc < 128) rpc /= 2;
    a.rows_per_chunk = rpc;
    const dim3 grid(static_cast<unsigned>(ld / brie::kGenesPerBlock), static_cast<unsigned>((Nc + rpc - 1) / rpc)), block(brie::kBlock);
    const int pad = 81 * 1024;
    auto kern = brie::placement_probe<brie::kCountU8, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pad);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int l = 0; l < n_layouts && e == hipSuccess; ++l) {
        const int64_t *o = offsets + 8 * l;
        float *st[6];
        for (int i = 0; i < 6; ++i) st[i] = reinterpret_cast<float *>(slab + o[i]);
        for (int it = -1; it < iters; ++it) {
            if (it == 0) hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(kern, grid, block, pad, nullptr, slab + o[6], slab + o[7], static_cast<const void *>(nullptr), st[0],
                               st[1], st[2], st[3], st[4], st[5], a, 0u, static_cast<uint32_t *>(nullptr));
        }
        hipEventRecord(e1, nullptr);
        e = hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        gbs[l] = static_cast<double>(Nc) * Ng * 50.0 * iters / (ms * 1e-3) / 1e9;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(slab);
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "layout probe: %s", hipGetErrorString(e));
    return BRIE_OK;
}

// Experiment aid (round 6): the streamed arrays of a u8-count problem of (Nc, Ng) built from the HIP virtual-
// memory API -- one address range per array, backed by physical chunks of `chunk_bytes` that are created in a chosen ORDER --
// and the placement probe timed on them.  order 0: array after array (what hipMalloc does, chunk by chunk); order 1: round
// robin -- chunk i of array 0, chunk i of array 1, ... -- so that the arrays of the set are interleaved in the order the
// driver hands out physical memory; order 2: plain hipMalloc per array (the baseline).  n_layers 2 or 3.  Per layout:
// gbs = storage bytes of a step / probe time, seconds = what building the set took.  Nothing is kept.
int brie_probe_vmm(int32_t device, int64_t Nc, int64_t Ng, int32_t n_layers, int32_t n_layouts, const int64_t *chunk_bytes,
                   const int32_t *order, int32_t iters, double *gbs, double *seconds) {
    if (!chunk_bytes || !order || !gbs || !seconds || n_layouts < 1 || iters < 1 || Nc <= 0 || Ng <= 0 || n_layers < 2 || n_layers > 3)
        return fail(BRIE_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const int64_t ld = round_up(Ng, brie::kGenesPerBlock);
    const int n_arr = 6 + n_layers;
    int64_t bytes[9];
    for (int i = 0; i < n_arr; ++i) bytes[i] = i < 6 ? Nc * ld * 4 : Nc * ld;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) return fail(BRIE_ERR_HIP, "allocation granularity 0");
    hipMemAccessDesc ad{};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    brie::StepScalars a{};
    a.ld = ld; a.row_stride = brie::kGenesPerBlock; a.gb_stride = Nc * brie::kGenesPerBlock;
    a.Nc = static_cast<int32_t>(Nc); a.Ng = static_cast<int32_t>(Ng);
    int rpc = 256;
    while (rpc > 16 && (Nc + rpc - 1) / rpc < 128) rpc /= 2;
    a.rows_per_chunk = rpc;
    const dim3 grid(static_cast<unsigned>(ld / brie::kGenesPerBlock), static_cast<unsigned>((Nc + rpc - 1) / rpc)), block(brie::kBlock);
    const int pad = 81 * 1024;
    auto k2 = brie::placement_probe<brie::kCountU8, false>;
    auto k3 = brie::placement_probe<brie::kCountU8, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, pad);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, pad);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = BRIE_OK;
    for (int l = 0; l < n_layouts && rc == BRIE_OK; ++l) {
        const auto t0 = std::chrono::steady_clock::now();
        char *va[9] = {};
        size_t vsz[9] = {};
        std::vector<hipMemGenericAllocationHandle_t> handles;
        hipError_t e = hipSuccess;
        if (order[l] == 2) {
            for (int i = 0; i < n_arr && e == hipSuccess; ++i) e = hipMalloc(reinterpret_cast<void **>(&va[i]), static_cast<size_t>(bytes[i]));
        } else {
            const size_t chunk = static_cast<size_t>(round_up(std::max<int64_t>(chunk_bytes[l], 1), static_cast<int64_t>(gran)));
            size_t n_chunks[9], most = 0;
            for (int i = 0; i < n_arr && e == hipSuccess; ++i) {
                n_chunks[i] = (static_cast<size_t>(bytes[i]) + chunk - 1) / chunk;
                most = std::max(most, n_chunks[i]);
                vsz[i] = n_chunks[i] * chunk;
                e = hipMemAddressReserve(reinterpret_cast<void **>(&va[i]), vsz[i], gran, nullptr, 0);
            }
            auto map_one = [&](int i, size_t c) {
                hipMemGenericAllocationHandle_t hd;
                e = hipMemCreate(&hd, chunk, &prop, 0);
                if (e != hipSuccess) return;
                handles.push_back(hd);
                e = hipMemMap(va[i] + c * chunk, chunk, 0, hd, 0);
            };
            if (order[l] == 3 || order[l] == 4) {
                // 3 "spread": every array ONE physical allocation, a spacer of chunk_bytes (physical memory that is created, never
                // mapped, and released before the probe) between consecutive arrays; 4 "skip": one such spacer first, then the
                // arrays back to back -- where in physical memory a packed set sits, if the driver hands memory out in order
                std::vector<hipMemGenericAllocationHandle_t> spacers;
                auto spacer = [&]() {
                    hipMemGenericAllocationHandle_t hd;
                    e = hipMemCreate(&hd, chunk, &prop, 0);
                    if (e == hipSuccess) spacers.push_back(hd);
                };
                if (order[l] == 4) spacer();
                for (int i = 0; i < n_arr && e == hipSuccess; ++i) {
                    (void)hipMemAddressFree(va[i], vsz[i]);
                    vsz[i] = static_cast<size_t>(round_up(bytes[i], static_cast<int64_t>(gran)));
                    e = hipMemAddressReserve(reinterpret_cast<void **>(&va[i]), vsz[i], gran, nullptr, 0);
                    hipMemGenericAllocationHandle_t hd;
                    if (e == hipSuccess) e = hipMemCreate(&hd, vsz[i], &prop, 0);
                    if (e != hipSuccess) break;
                    handles.push_back(hd);
                    e = hipMemMap(va[i], vsz[i], 0, hd, 0);
                    if (order[l] == 3 && i + 1 < n_arr && e == hipSuccess) spacer();
                }
                for (auto hd : spacers) (void)hipMemRelease(hd);
            } else if (order[l] == 0) {
                for (int i = 0; i < n_arr && e == hipSuccess; ++i)
                    for (size_t c = 0; c < n_chunks[i] && e == hipSuccess; ++c) map_one(i, c);
            } else {
                for (size_t c = 0; c < most && e == hipSuccess; ++c)
                    for (int i = 0; i < n_arr && e == hipSuccess; ++i)
                        if (c < n_chunks[i]) map_one(i, c);
            }
            for (int i = 0; i < n_arr && e == hipSuccess; ++i) e = hipMemSetAccess(va[i], vsz[i], &ad, 1);
        }
        for (int i = 0; i < n_arr && e == hipSuccess; ++i) e = hipMemsetAsync(va[i], 0, static_cast<size_t>(bytes[i]), nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        seconds[l] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (e == hipSuccess) {
            float *st[6];
            for (int i = 0; i < 6; ++i) st[i] = reinterpret_cast<float *>(va[i]);
            for (int it = -2; it < iters; ++it) {
                if (it == 0) hipEventRecord(e0, nullptr);
                if (n_layers == 3)
                    hipLaunchKernelGGL(k3, grid, block, pad, nullptr, va[6], va[7], va[8], st[0], st[1], st[2], st[3], st[4], st[5], a, 0u,
                                       static_cast<uint32_t *>(nullptr));
                else
                    hipLaunchKernelGGL(k2, grid, block, pad, nullptr, va[6], va[7], static_cast<const void *>(nullptr), st[0], st[1],
                                       st[2], st[3], st[4], st[5], a, 0u, static_cast<uint32_t *>(nullptr));
            }
            hipEventRecord(e1, nullptr);
            e = hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            gbs[l] = static_cast<double>(Nc) * Ng * (48.0 + n_layers) * iters / (ms * 1e-3) / 1e9;
        }
        if (e != hipSuccess) rc = fail(BRIE_ERR_HIP, "vmm probe, layout %d: %s", l, hipGetErrorString(e));
        (void)hipGetLastError();
        if (order[l] == 2) {
            for (int i = 0; i < n_arr; ++i) if (va[i]) (void)hipFree(va[i]);
        } else {
            for (int i = 0; i < n_arr; ++i) if (va[i]) { (void)hipMemUnmap(va[i], vsz[i]); }
            for (auto hd : handles) (void)hipMemRelease(hd);
            for (int i = 0; i < n_arr; ++i) if (va[i]) (void)hipMemAddressFree(va[i], vsz[i]);
        }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}

// Measure the HBM rate of `n_read` read streams + `n_write` write streams of `bytes_per_stream`
// each (no arithmetic) -> GB/s.  Supported mixes: (1,1) copy, (8,6) and (9,6) = elbo_adam_step with
// 2 / 3 count layers.  `lds_bytes_per_block` > 0 reserves dynamic LDS per 256-thread block to cap the
// occupancy (160 KiB per CU: 80 KiB -> 2 blocks = 8 waves per CU), i.e. the bytes in flight per CU;
// a negative value -(lds+1) selects non-temporal loads/stores (what elbo_adam_step uses).
int brie_calibrate_stream(int32_t device, int32_t n_read, int32_t n_write, int64_t bytes_per_stream, int32_t iters,
                          int32_t lds_bytes_per_block, double *gbps) {
    if (!gbps || iters < 1 || bytes_per_stream < 4096) return fail(BRIE_ERR_INVALID, "bad argument");
    if (!((n_read == 1 && n_write == 1) || (n_read == 8 && n_write == 6) || (n_read == 9 && n_write == 6)))
        return fail(BRIE_ERR_INVALID, "unsupported mix %d/%d", n_read, n_write);
    HIP_TRY(hipSetDevice(device));
    brie::StreamArgs a{};
    a.n4 = bytes_per_stream / 16;
    std::vector<void *> bufs;
    auto cleanup = [&]() { for (void *q : bufs) hipFree(q); };
    for (int i = 0; i < n_read + n_write; ++i) {
        void *q = nullptr;
        if (dev_alloc(&q, a.n4 * 16) != hipSuccess) { cleanup(); return fail(BRIE_ERR_HIP, "hipMalloc calibration buffer"); }
        hipMemset(q, 0, a.n4 * 16);
        bufs.push_back(q);
        if (i < n_read) a.in[i] = static_cast<const float *>(q); else a.out[i - n_read] = static_cast<float *>(q);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(8192), block(brie::kBlock);
    const bool nt = lds_bytes_per_block < 0;          // negative: non-temporal variant, |value| = LDS bytes
    const int lds = lds_bytes_per_block < 0 ? -lds_bytes_per_block - 1 : lds_bytes_per_block;
    // three depths of loads in flight per thread (1, 2, 4 vectors x n_read streams); the best one is reported
    float ms = 0.f;
    hipError_t e = hipSuccess;
    for (int unroll = 1; unroll <= 4 && e == hipSuccess; unroll *= 2) {
        for (int it = -2; it < iters; ++it) {
            if (it == 0) hipEventRecord(e0, nullptr);
#define BRIE_SM2(NR, NW, U)                                                                               \
    do {                                                                                                 \
        if (nt) hipLaunchKernelGGL((brie::stream_mix<NR, NW, true, U>), grid, block, lds, nullptr, a);   \
        else hipLaunchKernelGGL((brie::stream_mix<NR, NW, false, U>), grid, block, lds, nullptr, a);     \
    } while (0)
#define BRIE_SM(NR, NW)                                  \
    do {                                                 \
        if (unroll == 1) BRIE_SM2(NR, NW, 1);            \
        else if (unroll == 2) BRIE_SM2(NR, NW, 2);       \
        else BRIE_SM2(NR, NW, 4);                        \
    } while (0)
            if (n_read == 1) BRIE_SM(1, 1);
            else if (n_read == 8) BRIE_SM(8, 6);
            else BRIE_SM(9, 6);
#undef BRIE_SM
#undef BRIE_SM2
        }
        hipEventRecord(e1, nullptr);
        e = hipEventSynchronize(e1);
        float t = 0.f;
        hipEventElapsedTime(&t, e0, e1);
        if (e == hipSuccess && (ms == 0.f || t < ms)) ms = t;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    cleanup();
    if (e != hipSuccess) return fail(BRIE_ERR_HIP, "calibration: %s", hipGetErrorString(e));
    *gbps = static_cast<double>(n_read + n_write) * a.n4 * 16.0 * iters / (ms * 1e-3) / 1e9;
    return BRIE_OK;
}

// ---- count simulator: stateless, row-major host or device arrays, processed in row slabs -------------
namespace {
struct SlabBuffers {
    std::vector<void *> bufs;
    ~SlabBuffers() { for (void *q : bufs) hipFree(q); }
    float *get(size_t elems) {
        void *q = nullptr;
        if (elems == 0 || dev_alloc(&q, elems * sizeof(float)) != hipSuccess) return nullptr;
        bufs.push_back(q);
        return static_cast<float *>(q);
    }
};
// elements per row slab staged through HBM (5 fp32 slabs live at once); BRIE_SIM_SLAB_ELEMS overrides (tests)
int64_t sim_slab_elems() {
    const char *e = getenv("BRIE_SIM_SLAB_ELEMS");
    const int64_t v = e ? atoll(e) : 0;
    return v > 0 ? v : (int64_t(1) << 27);
}
}  // namespace

int brie_trim_memory(void) {
    g_blocks.trim();
    return BRIE_OK;
}

int brie_device_memory(int32_t device, int64_t *free_bytes, int64_t *total_bytes) {
    if (!free_bytes || !total_bytes) return fail(BRIE_ERR_INVALID, "null argument");
    g_blocks.trim();                       // what the last handle left behind counts as free
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = static_cast<int64_t>(f);
    *total_bytes = static_cast<int64_t>(t);
    return BRIE_OK;
}

int brie_simulate_psi(int32_t device, int64_t Nc, int64_t Ng, int64_t gene_offset, uint64_t seed,
                      const float *mean_logit, const float *sigma, float *psi_out) {
    if (!mean_logit || !sigma || !psi_out) return fail(BRIE_ERR_INVALID, "null argument");
    if (Nc <= 0 || Ng <= 0 || Nc > INT32_MAX || Ng > INT32_MAX || gene_offset < 0 || gene_offset % 4 != 0)
        return fail(BRIE_ERR_INVALID, "bad shape / gene_offset (Nc=%lld Ng=%lld gene_offset=%lld)", (long long)Nc,
                    (long long)Ng, (long long)gene_offset);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    const int64_t slab = std::max<int64_t>(1, std::min<int64_t>(Nc, sim_slab_elems() / Ng));
    SlabBuffers sb;
    float *d_mean = sb.get(slab * Ng), *d_out = sb.get(slab * Ng), *d_sig = sb.get(Ng);
    if (!d_mean || !d_out || !d_sig) return fail(BRIE_ERR_HIP, "hipMalloc simulator slab");
    HIP_TRY(hipMemcpy(d_sig, sigma, Ng * sizeof(float), hipMemcpyDefault));
    for (int64_t r0 = 0; r0 < Nc; r0 += slab) {
        const int64_t rows = std::min(slab, Nc - r0);
        HIP_TRY(hipMemcpy(d_mean, mean_logit + r0 * Ng, rows * Ng * sizeof(float), hipMemcpyDefault));
        hipLaunchKernelGGL(brie::sim_psi, dim3(grid_1d(rows * ((Ng + 3) / 4))), dim3(256), 0, nullptr, d_mean, d_sig, d_out,
                           rows, Ng, r0, gene_offset, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(psi_out + r0 * Ng, d_out, rows * Ng * sizeof(float), hipMemcpyDefault));
    }
    return BRIE_OK;
}

int brie_simulate_counts(int32_t device, int64_t Nc, int64_t Ng, int64_t gene_offset, uint64_t seed, const float *psi,
                         const float *total, const float *effLen, float *out1, float *out2, float *out3) {
    if (!psi || !total || !out1 || !out2 || (effLen && !out3)) return fail(BRIE_ERR_INVALID, "null argument");
    if (Nc <= 0 || Ng <= 0 || Nc > INT32_MAX || Ng > INT32_MAX || gene_offset < 0)
        return fail(BRIE_ERR_INVALID, "bad shape / gene_offset (Nc=%lld Ng=%lld gene_offset=%lld)", (long long)Nc,
                    (long long)Ng, (long long)gene_offset);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    const int64_t slab = std::max<int64_t>(1, std::min<int64_t>(Nc, sim_slab_elems() / Ng));
    SlabBuffers sb;
    float *d_psi = sb.get(slab * Ng), *d_tot = sb.get(slab * Ng), *d_o1 = sb.get(slab * Ng), *d_o2 = sb.get(slab * Ng);
    float *d_o3 = effLen ? sb.get(slab * Ng) : nullptr, *d_eff = effLen ? sb.get(Ng * 6) : nullptr;
    if (!d_psi || !d_tot || !d_o1 || !d_o2 || (effLen && (!d_o3 || !d_eff)))
        return fail(BRIE_ERR_HIP, "hipMalloc simulator slab");
    if (effLen) HIP_TRY(hipMemcpy(d_eff, effLen, Ng * 6 * sizeof(float), hipMemcpyDefault));
    for (int64_t r0 = 0; r0 < Nc; r0 += slab) {
        const int64_t rows = std::min(slab, Nc - r0);
        const size_t bytes = static_cast<size_t>(rows) * Ng * sizeof(float);
        HIP_TRY(hipMemcpy(d_psi, psi + r0 * Ng, bytes, hipMemcpyDefault));
        HIP_TRY(hipMemcpy(d_tot, total + r0 * Ng, bytes, hipMemcpyDefault));
        hipLaunchKernelGGL(brie::sim_counts, dim3(grid_1d(rows * Ng)), dim3(256), 0, nullptr, d_psi, d_tot, d_eff, d_o1, d_o2,
                           d_o3, rows, Ng, r0, gene_offset, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out1 + r0 * Ng, d_o1, bytes, hipMemcpyDefault));
        HIP_TRY(hipMemcpy(out2 + r0 * Ng, d_o2, bytes, hipMemcpyDefault));
        if (effLen) HIP_TRY(hipMemcpy(out3 + r0 * Ng, d_o3, bytes, hipMemcpyDefault));
    }
    return BRIE_OK;
}

}  // extern "C"

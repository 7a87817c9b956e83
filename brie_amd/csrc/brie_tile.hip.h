// brie_tile.hip.h -- the step kernel for WIDE designs: cell features Kc = 9..64 and / or gene features Kg = 5..64.
//
// These are the only places of the path with a genuine dense contraction (2 Nc Ng K FLOP per product):
//   forward   M = Xc . Wc_loc + Wg_loc . Xg^T                 (model_TFProb.py:122-125)
//   backward  G = Xc^T . R   (gradient of Wc_loc, over cells)  and  P = R . Xg (gradient of Wg_loc, over genes)
// with R the residual (mu - m) / sigma^2.  All three run on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// fma chains) INSIDE the streaming pass, so nothing but the state and the counts touches HBM:
//
//   a workgroup (4 waves) owns one 256-gene block and one cell chunk and walks it in tiles of 32 cells:
//     A  forward: every wave computes 2 of the 8 32-gene column blocks of the 32 x 256 prior-mean tile with MFMAs
//        (B operands from the Wc_loc / Xg tiles of the gene block in LDS, A operands = the tile's design rows
//        straight from L2) and drops it into the LDS tile T;
//     B  stream: wave w takes rows w, w+4, ... of the tile exactly like elbo_adam_step (lane owns 4 genes, 16-B
//        non-temporal loads, next row prefetched under the current row's arithmetic), reads its prior mean from T
//        (one ds_read_b128) and writes the residual back to the same place;
//     C  backward: G += Xc^T . T on the wave's 2 column blocks (accumulators live in registers for the whole
//        chunk); P = T . Xg in 16 x 16 output blocks (v_mfma_f32_16x16x4_f32), each wave contracting its blocks over
//        all 256 genes itself -- no cross-wave fold -- and storing them straight into the per-cell statistics.
//
// Replaces (measured in DESIGN 4.5): the v_readlane + ds_read_b128 + 4 FMA per feature of the LDS-broadcast
// variants, the 8 B/element residual round trip through HBM and the separate wide_design_grad launch.
#pragma once
#include "brie_kernels.hip.h"

namespace brie {

constexpr int kTileRows = 32;
constexpr int kTileStride = kGenesPerBlock + 4;      // floats per tile row: 16-B aligned rows, column reads 4-way banked
constexpr int kXgStride = kGenesPerBlock + 4;        // Xg tile rows: 16-B aligned; 16 lanes walking features at one gene offset hit
                                                     // 16 different bank quads (ds_read_b128 in the R.Xg product)

__host__ __device__ constexpr int tile_a_stride(int ka) { return ka | 1; }

struct TileArgs {
    const float *Xc;            // (Nc, Kc)
    const float *W;             // (Kc, ld)
    const float *Xg;            // (kgp, ld) transposed gene features, zero beyond Kg / Ng
    const float *Wg;            // (Nc, kgp)
    const float *cb, *clam;     // (Nc) per-cell intercept / log sigma (cell mode)
    float *row_partials;        // (gene_blocks, (kgp + 2) * Nc)
    float *Gpart;               // (n_chunks, Kc, ld) partial sums of Xc^T . R
    int32_t Kc, Kg, kgp, cell_mode;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DBRIE_TILE_PROF=1 (tuning builds only): every wave sums the cycles it spends in the phases of a tile; wave 0 lane 0 of
// each half adds them to tile_prof[] (read with brie_debug_tile_prof, profiles/tile_phases.py).
#ifndef BRIE_TILE_PROF
#define BRIE_TILE_PROF 0
#endif
#if BRIE_TILE_PROF
static __device__ unsigned long long tile_prof[16];
#define BRIE_PROF_MARK(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_[slot] += now_ - prof_t_; prof_t_ = now_; } while (0)
#else
#define BRIE_PROF_MARK(slot) do { } while (0)
#endif

// NACC = 32-feature accumulator sets for Xc^T.R (0: Kc == 0, 1: Kc <= 32, 2: Kc <= 64)
// NJT  = 32-feature output tiles of R.Xg        (0: Kg == 0, 1: kgp <= 32, 2: kgp <= 64)
// NH   = independent 4-wave halves per workgroup.  1: a plain 4-wave workgroup, two of them per CU while the LDS tiles
//        stay under 80 KB.  2: when they do not, ONE 8-wave workgroup per CU whose two halves are two virtual workgroups:
//        each owns its own cell chunk, its own T tile and its own barriers (an LDS counter the 4 waves spin on -- gfx950
//        has no named barriers), only the read-only Wc_loc / Xg tiles are shared.  While one half is in its MFMA phases
//        the other keeps streaming, exactly as two real workgroups would, and per-chunk results are the same bit for bit.
template <int NH>
__device__ __forceinline__ void tile_sync(int *ctr, int &arrived) {
    if constexpr (NH == 1) {
        __syncthreads();
    } else {                                   // barrier of ONE half (4 waves): monotone arrival counter in LDS
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        arrived += kWavesPerBlock;
        if ((threadIdx.x & (kWave - 1)) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < arrived) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// KCR  = 4: gene-feature models with at most 4 cell features (the usual kind: a few covariates next to Xg).  A 32-feature
//        MFMA tile would contract 3 real rows of 32 -- 8 % of a tile's cycles in X^T.R alone (profiles/history/r03q_tile_phases.log)
//        -- so the cell design is handled the way elbo_adam_step does it: weights and X^T.R accumulators in registers,
//        the cell's design row through scalar loads, 2 x 4 FMAs per feature and row.  NACC must be 0.
template <int MODE, int CS, int NACC, int NJT, int NH, int KCR = 0>
__global__ __launch_bounds__(NH * kBlock, 2) void elbo_adam_step_tile(   // 2 waves/SIMD: <= 256 registers, no spills
    const void *__restrict__ c1p, const void *__restrict__ c2p, const void *__restrict__ c3p,
    float *__restrict__ mu_p, float *__restrict__ rho_p, float *__restrict__ mmu_p,
    float *__restrict__ vmu_p, float *__restrict__ mrho_p, float *__restrict__ vrho_p,
    const float *__restrict__ bp, const float *__restrict__ lamp, const float *__restrict__ effL,
    float *__restrict__ partials, const StepScalars a, const TileArgs t) {
    constexpr int S = 4;
    constexpr int NW = kWavesPerBlock;      // waves per half
    constexpr int CB = 8 / NW;              // 32-gene column blocks per wave
    constexpr int NT = NH * kBlock;         // threads per workgroup
    constexpr bool CPL = NJT > 0;           // per-cell statistics are only produced with gene features / cell mode
    constexpr bool REGK = KCR > 0;          // cell features in registers, none on the matrix cores
    static_assert(!REGK || NACC == 0, "KCR > 0 replaces the MFMA path of the cell design");
    constexpr int KR = REGK ? KCR : 1;
    // dynamic LDS: NH x [T tile 32 x 260][W tile Kc x 256][Xg tile kgp x 257] NH x [At 32 x (Kc + kgp | 1)]; the cross-wave folds reuse T
    extern __shared__ __align__(16) float lds[];
    __shared__ int bar_ctr[2];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hf = wave / NW;               // which half (virtual workgroup)
    const int w = wave - hf * NW;           // wave within the half
    const int tid = threadIdx.x - hf * kBlock;      // thread within the half
    float *T = lds + hf * (kTileRows * kTileStride);
    float *wl = lds + NH * (kTileRows * kTileStride);
    const int kcm = REGK ? 0 : t.Kc;        // cell features that go through the matrix cores
    float *xl = wl + kcm * kGenesPerBlock;
    // A-operand tile of the half: the design rows [Xc | Wg_loc] of the tile's 32 cells (odd row stride: conflict-free when
    // the 32 lanes of a half-wave walk cells)
    const int KA = kcm + t.kgp, AS = tile_a_stride(KA);
    float *At = xl + t.kgp * kXgStride + hf * (kTileRows * AS);
    int *ctr = bar_ctr + hf;
    int arrived = 0;
    if (threadIdx.x < 2) bar_ctr[threadIdx.x] = 0;
    const int half = lane >> 5, l31 = lane & 31;
    const int gb = static_cast<int>(blockIdx.x);
    const int quad = gb * kWave + lane;
    const int j0 = quad * kVec;
    const bool active = j0 < a.Ng;
    const int chunk_id = blockIdx.y * NH + hf;             // every half owns one cell chunk
    const int row0 = chunk_id * a.rows_per_chunk;
    const int row_end = min(row0 + a.rows_per_chunk, a.Nc);
    if (a.block_active[gb] == 0) return;
    const bool cell = CPL && t.cell_mode != 0;

    for (int i = threadIdx.x; i < kcm * kGenesPerBlock; i += NT)
        wl[i] = t.W[static_cast<int64_t>(i / kGenesPerBlock) * a.ld + gb * kGenesPerBlock + (i % kGenesPerBlock)];
    if constexpr (NJT > 0) {
        for (int i = threadIdx.x; i < t.kgp * kGenesPerBlock; i += NT)
            xl[(i / kGenesPerBlock) * kXgStride + (i % kGenesPerBlock)] =
                t.Xg[static_cast<int64_t>(i / kGenesPerBlock) * a.ld + gb * kGenesPerBlock + (i % kGenesPerBlock)];
    }

    float acc[S][kVec];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int v = 0; v < kVec; ++v) acc[s][v] = 0.0f;
    float accW[KR][kVec];                   // REGK: sum over the chunk's cells of x_k * residual
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int v = 0; v < kVec; ++v) accW[k][v] = 0.0f;
    f32x16 G[NACC > 0 ? NACC : 1][CB];
#pragma unroll
    for (int n = 0; n < (NACC > 0 ? NACC : 1); ++n)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) G[n][c][q] = 0.0f;

    // Per-gene parameters of the lane's 4 genes.  They are only needed while rows stream (phase B), so every tile fetches
    // them again (L2 hits, 3 - 9 float4 per lane against the tile's 410 KB) instead of holding 12 - 36 registers through the
    // MFMA phases, whose operand double buffers need them; the laundered index keeps the loads inside the tile loop.
    float bj[kVec], lamj[kVec], isig2[kVec];
    float L0[kVec], L4[kVec], L5[kVec], lL0[kVec], lL4[kVec], lL5[kVec];
    float Wk[KR][kVec];
    auto load_gene_params = [&]() {
        int jj = j0;
        asm volatile("" : "+v"(jj));
        const F4 tb = ld4(bp + jj), tl = ld4(lamp + jj);
        if constexpr (REGK) {
#pragma unroll
            for (int k = 0; k < KCR; ++k) {
                F4 wk = {{0.f, 0.f, 0.f, 0.f}};
                if (k < t.Kc) wk = ld4(t.W + static_cast<int64_t>(k) * a.ld + jj);
#pragma unroll
                for (int v = 0; v < kVec; ++v) Wk[k][v] = wk.v[v];
            }
        }
#pragma unroll
        for (int v = 0; v < kVec; ++v) {
            bj[v] = tb.v[v];
            lamj[v] = tl.v[v];
            isig2[v] = f_exp(-2.0f * tl.v[v]);
        }
        if (MODE != kLik2) {
            const F4 t0 = ld4(effL + 0 * a.ld + jj), t1 = ld4(effL + 1 * a.ld + jj), t2 = ld4(effL + 2 * a.ld + jj),
                     t3 = ld4(effL + 3 * a.ld + jj), t4 = ld4(effL + 4 * a.ld + jj), t5 = ld4(effL + 5 * a.ld + jj);
#pragma unroll
            for (int v = 0; v < kVec; ++v) {
                L0[v] = t0.v[v]; L4[v] = t1.v[v]; L5[v] = t2.v[v];
                lL0[v] = t3.v[v]; lL4[v] = t4.v[v]; lL5[v] = t5.v[v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < kVec; ++v) { L0[v] = L4[v] = L5[v] = lL0[v] = lL4[v] = lL5[v] = 0.0f; }
        }
    };
    const uint32_t gquad = a.quad_offset + static_cast<uint32_t>(a.quad_ids[quad]);
    const int64_t mbase = static_cast<int64_t>(gb) * a.gb_stride + lane * kVec;
    const int esz = CS == kCountMixed ? a.tt.q_esz[quad] : 0;
    const int64_t cbase = CS == kCountMixed ? a.tt.blk_base[gb] + a.tt.q_off[quad] : mbase;
    const int64_t crow = CS == kCountMixed ? a.tt.row_bytes[gb] : a.row_stride;
    bool on[kVec], real[kVec];
    {
        const F4 tt = ld4(a.gene_active + j0);
#pragma unroll
        for (int v = 0; v < kVec; ++v) { on[v] = tt.v[v] != 0.0f; real[v] = j0 + v < a.Ng; }
    }
    __syncthreads();                                       // W / Xg tiles and the barrier counters complete (whole workgroup)
    if (row0 >= a.Nc) return;                              // odd number of chunks: this half has none (no later WG-wide barrier)

    auto load_row = [&](int r, RowRegs<CS> &R) {
        const int64_t off = mbase + static_cast<int64_t>(r) * a.row_stride;
        load_counts<CS, MODE>(c1p, c2p, c3p, cbase + static_cast<int64_t>(r) * crow, R.cnt, esz);
        R.mu = ld4s(mu_p + off);
        R.rho = ld4s(rho_p + off);
        R.mm = ld4s(mmu_p + off);
        R.vm = ld4s(vmu_p + off);
        R.mr = ld4s(mrho_p + off);
        R.vr = ld4s(vrho_p + off);
        if constexpr (REGK) {               // the cell's design row (wave-uniform: scalar loads)
#pragma unroll
            for (int k = 0; k < KCR; ++k) R.mp.v[k] = k < t.Kc ? t.Xc[static_cast<int64_t>(r) * t.Kc + k] : 0.0f;
        }
    };

    // one streamed row: prior mean from T[ti], residual back to T[ti]
    auto process_row = [&](int r, int ti, RowRegs<CS> &R) {
        const int64_t off = mbase + static_cast<int64_t>(r) * a.row_stride;
        float *trow = T + ti * kTileStride + lane * kVec;
        const F4 mt = ld4(trow);
        const float cbr = cell ? t.cb[r] : 0.0f, clamr = cell ? t.clam[r] : 0.0f;
        const float row_isig2 = cell ? f_exp(-2.0f * clamr) : 0.0f;
        F4 c1, c2, c3;
        decode_counts<CS>(R.cnt, a.pc, c1, c2, c3);
        float gbar[kVec] = {0.f, 0.f, 0.f, 0.f}, gse[kVec] = {0.f, 0.f, 0.f, 0.f}, ll[kVec] = {0.f, 0.f, 0.f, 0.f}, s[kVec];
#pragma unroll
        for (int v = 0; v < kVec; ++v) s[v] = f_exp(R.rho.v[v]);
        for (int k = 0; k < a.mc; ++k) {
            float e[kVec];
            normal4(gquad, static_cast<uint32_t>(r), a.draw, static_cast<uint32_t>(k), a.seed_lo, a.seed_hi, e);
#pragma unroll
            for (int v = 0; v < kVec; ++v) {
                const float z = fmaf(s[v], e[v], R.mu.v[v]);
                float l, g;
                loglik<MODE>(z, c1.v[v], c2.v[v], c3.v[v], L0[v], L4[v], L5[v], lL0[v], lL4[v], lL5[v], l, g);
                ll[v] += l;
                gbar[v] += g;
                gse[v] = fmaf(g, e[v], gse[v]);
            }
        }
        F4 res;
        float srow = 0.0f, lrow = 0.0f;                    // per-cell sums over the lane's real genes
#pragma unroll
        for (int v = 0; v < kVec; ++v) {
            float m = mt.v[v] + (cell ? cbr : bj[v]);
            if constexpr (REGK) {
#pragma unroll
                for (int k = 0; k < KCR; ++k) m = fmaf(R.mp.v[k], Wk[k][v], m);
            }
            const float is2 = cell ? row_isig2 : isig2[v];
            const float d = R.mu.v[v] - m;
            const float rr = d * is2;
            const float s2r = s[v] * s[v] * is2;
            const float dl = R.rho.v[v] - (cell ? clamr : lamj[v]);
            const float kl = 0.5f * d * rr + 0.5f * (s2r - 1.0f) - dl;
            const float lamstat = 1.0f - d * rr - s2r;
            const float g_mu = rr - gbar[v] * a.inv_mc;
            const float g_rho = s2r - 1.0f - gse[v] * s[v] * a.inv_mc;
            const float n_mm = R.mm.v[v] + (g_mu - R.mm.v[v]) * kOneMinusB1;
            const float n_vm = R.vm.v[v] + (g_mu * g_mu - R.vm.v[v]) * kOneMinusB2;
            const float n_mr = R.mr.v[v] + (g_rho - R.mr.v[v]) * kOneMinusB1;
            const float n_vr = R.vr.v[v] + (g_rho * g_rho - R.vr.v[v]) * kOneMinusB2;
            float nmu = adam_update(R.mu.v[v], n_mm, n_vm, a.alpha);
            nmu = fminf(fmaxf(nmu, -9.0f), 9.0f);
            const float nrho = adam_update(R.rho.v[v], n_mr, n_vr, a.alpha);
            R.mm.v[v] = on[v] ? n_mm : R.mm.v[v];
            R.vm.v[v] = on[v] ? n_vm : R.vm.v[v];
            R.mr.v[v] = on[v] ? n_mr : R.mr.v[v];
            R.vr.v[v] = on[v] ? n_vr : R.vr.v[v];
            R.mu.v[v] = on[v] ? nmu : R.mu.v[v];
            R.rho.v[v] = on[v] ? nrho : R.rho.v[v];
            res.v[v] = real[v] ? rr : 0.0f;                // padding genes are not part of any contraction
            srow += real[v] ? rr : 0.0f;
            lrow += real[v] ? lamstat : 0.0f;
            if constexpr (REGK) {
#pragma unroll
                for (int k = 0; k < KCR; ++k) accW[k][v] = fmaf(R.mp.v[k], rr, accW[k][v]);
            }
            acc[0][v] += rr;
            acc[1][v] += lamstat;
            acc[2][v] += kl;
            acc[3][v] += ll[v] * a.inv_mc;
        }
        st4(trow, res);
        if (active) {
            st4s(mu_p + off, R.mu);
            st4s(rho_p + off, R.rho);
            st4s(mmu_p + off, R.mm);
            st4s(vmu_p + off, R.vm);
            st4s(mrho_p + off, R.mr);
            st4s(vrho_p + off, R.vr);
        }
        if constexpr (CPL) {        // cell mode only: sum_j r and sum_j (1 - d r - s2r) of this cell over the gene block
            if (cell) {             // (wave-uniform; the per-cell intercept / sigma are not parameters otherwise)
                float *chunk = t.row_partials + static_cast<int64_t>(gb) * (t.kgp + 2) * a.Nc;
                const float ts = wave_sum(active ? srow : 0.0f), tl2 = wave_sum(active ? lrow : 0.0f);
                if (lane == 0) {
                    chunk[static_cast<int64_t>(t.kgp) * a.Nc + r] = ts;
                    chunk[static_cast<int64_t>(t.kgp + 1) * a.Nc + r] = tl2;
                }
            }
        }
    };

    // Addresses that only depend on the lane are tile-invariant; hoisted out of the tile loop they would sit in ~20
    // VGPRs through the streaming phase (the kernel's register peak).  fresh() hides the lane index from that hoisting:
    // each MFMA phase recomputes its handful of addresses.
    auto fresh = [](int x) { asm volatile("" : "+v"(x)); return x; };
    RowRegs<CS> cur;
    load_row(min(row0 + w, row_end - 1), cur);             // (a wave without rows in this chunk loads one and drops it)
#if BRIE_TILE_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t_ = __builtin_readcyclecounter();
#endif

    // The design rows of a tile are two contiguous runs in memory (32 x Kc of Xc, 32 x kgp of Wg_loc): the half fetches
    // them with coalesced loads into registers while the previous tile finishes, and drops them into At once that tile's
    // readers are through.  (Each lane fetching its own A operand inside the MFMA loop -- one 4-byte load per lane, row
    // and k-step -- cost 64 cache lines per instruction and one L2 round trip per k-step: 17 000 cycles for a forward
    // product whose MFMAs take 4 400, profiles/history/r03j_tile_phases.log.)
    constexpr int NXL = 4 * NACC, NGL = 4 * NJT;            // loads per thread: 32 x 32 NACC (NJT) floats over 256 threads
    float pre_x[NXL > 0 ? NXL : 1], pre_g[NGL > 0 ? NGL : 1];
    const float inv_kc = t.Kc > 0 ? 1.0f / static_cast<float>(t.Kc) : 0.0f;
    const float inv_kg = t.kgp > 0 ? 1.0f / static_cast<float>(t.kgp) : 0.0f;
    // (the empty asm keeps the per-thread element indices -- tile-invariant -- from being hoisted out of the tile loop,
    // where 16 addresses + 16 predicates would stay live across the streaming phase, the kernel's register peak)
    auto fetch_design = [&](int tr0) {
        const int n_ok = min(kTileRows, row_end - tr0);    // rows of the tile inside the chunk
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            int e = tid + kBlock * i;
            asm volatile("" : "+v"(e));
            pre_x[i] = e < n_ok * t.Kc ? t.Xc[static_cast<int64_t>(tr0) * t.Kc + e] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NGL; ++i) {
            int e = tid + kBlock * i;
            asm volatile("" : "+v"(e));
            pre_g[i] = e < n_ok * t.kgp ? t.Wg[static_cast<int64_t>(tr0) * t.kgp + e] : 0.0f;
        }
    };
    auto store_design = [&]() {
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            int e = tid + kBlock * i;
            asm volatile("" : "+v"(e));
            const int row = static_cast<int>((static_cast<float>(e) + 0.5f) * inv_kc);        // e / Kc (exact: e < 4096)
            if (e < kTileRows * t.Kc) At[row * AS + (e - row * t.Kc)] = pre_x[i];
        }
#pragma unroll
        for (int i = 0; i < NGL; ++i) {
            int e = tid + kBlock * i;
            asm volatile("" : "+v"(e));
            const int row = static_cast<int>((static_cast<float>(e) + 0.5f) * inv_kg);
            if (e < kTileRows * t.kgp) At[row * AS + kcm + (e - row * t.kgp)] = pre_g[i];
        }
    };
    fetch_design(row0);

    for (int tr0 = row0; tr0 < row_end; tr0 += kTileRows) {
        store_design();
        tile_sync<NH>(ctr, arrived);
        BRIE_PROF_MARK(6);
        // ---- A: prior-mean tile on the matrix cores: T[i][gene] = sum_k X[i][k] W[k][gene] + sum_k Wg[i][k] Xg[gene][k]
        {
            f32x16 D[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) D[c][q] = 0.0f;
            // Operands are read from LDS four k-steps at a time, one batch ahead of the MFMAs that use them (a
            // read-wait-MFMA sequence per step would expose one LDS latency per 64 MFMA cycles); k beyond K reads row
            // K - 1 and contributes a zero A operand -- no branch in the loop.  (Pinning "reads of the next batch, then
            // the MFMAs" with sched_group_barrier was measured and is slower: profiles/history/r03r_tile_phases.log.)
            const int l31 = fresh(lane) & 31, half = fresh(lane) >> 5;
            const float *arow_l = At + l31 * AS;            // the lane's A row (cell); rows past the chunk hold zeros
            auto forward = [&](const float *arow, const float *btile, int bstride, int K) {
                if (K <= 0) return;
                const float *bl = btile + (CB * w) * 32 + l31;
                auto fetch = [&](int k0, float (&av)[4], float (&bv)[4][CB]) {   // MFMA step j contracts k0 + 2 j + {0, 1}
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + 2 * j + half, kc = min(k, K - 1);
                        const float x = arow[kc];
                        av[j] = k < K ? x : 0.0f;
#pragma unroll
                        for (int c = 0; c < CB; ++c) bv[j][c] = bl[kc * bstride + c * 32];
                    }
                };
                float av[4], bv[4][CB];
                fetch(0, av, bv);
#pragma unroll 1
                for (int k0 = 0; k0 < K; k0 += 8) {
                    float nav[4], nbv[4][CB];
                    fetch(k0 + 8, nav, nbv);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < CB; ++c) D[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j][c], D[c], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        av[j] = nav[j];
#pragma unroll
                        for (int c = 0; c < CB; ++c) bv[j][c] = nbv[j][c];
                    }
                }
            };
            forward(arow_l, wl, kGenesPerBlock, kcm);
            if constexpr (NJT > 0) forward(arow_l + kcm, xl, kXgStride, t.kgp);
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
                    T[i * kTileStride + (CB * w + c) * 32 + l31] = D[c][q];
                }
        }
        BRIE_PROF_MARK(0);
        tile_sync<NH>(ctr, arrived);
        BRIE_PROF_MARK(1);

        // ---- B: stream the wave's rows of this tile (software-pipelined: next row's loads under this row's math)
        {
            load_gene_params();
            const int t_end = min(tr0 + kTileRows, row_end);
            int r = tr0 + w;
            while (r < t_end) {
                const int rn = r + NW;                      // next row of this wave (may belong to the next tile)
                RowRegs<CS> nxt;
                // branch-free prefetch (a guarded load would put a join -- and a full vmcnt wait -- right behind it):
                // past the chunk's end the wave's own row is fetched again, an L2 hit
                load_row(rn < row_end ? rn : r, nxt);
                process_row(r, r - tr0, cur);
                cur = nxt;
                r = rn;
            }
        }
        BRIE_PROF_MARK(2);
        tile_sync<NH>(ctr, arrived);
        BRIE_PROF_MARK(3);

        // ---- C: backward contractions of the residual tile
        if constexpr (NACC > 0) {           // G[feature][gene] += sum_cells X[cell][feature] T[cell][gene]
            const int l31 = fresh(lane) & 31, half = fresh(lane) >> 5;
            const float *tl = T + (CB * w) * 32 + l31;
            const int n_ok = min(kTileRows, row_end - tr0);      // rows of the tile inside the chunk (At is zero beyond)
            auto fetch = [&](int k0, float (&av)[4][NACC], float (&bv)[4][CB]) {     // four MFMA steps (cell pairs)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ti = (k0 + 2 * j + half) & (kTileRows - 1);
#pragma unroll
                    for (int n = 0; n < NACC; ++n) {
                        const int f = l31 + 32 * n;
                        const float x = At[ti * AS + min(f, t.Kc - 1)];
                        av[j][n] = f < t.Kc ? x : 0.0f;
                    }
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        const float y = tl[ti * kTileStride + c * 32];
                        bv[j][c] = ti < n_ok ? y : 0.0f;        // (T rows past the chunk hold the forward product, not a residual)
                    }
                }
            };
            float av[4][NACC], bv[4][CB];
            fetch(0, av, bv);
#pragma unroll 1
            for (int k0 = 0; k0 < kTileRows; k0 += 8) {
                float nav[4][NACC], nbv[4][CB];
                fetch(k0 + 8, nav, nbv);                        // (the last batch reads batch 0 again and drops it)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < CB; ++c)
#pragma unroll
                        for (int n = 0; n < NACC; ++n)
                            G[n][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][n], bv[j][c], G[n][c], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int n = 0; n < NACC; ++n) av[j][n] = nav[j][n];
#pragma unroll
                    for (int c = 0; c < CB; ++c) bv[j][c] = nbv[j][c];
                }
            }
        }
        BRIE_PROF_MARK(4);
        if constexpr (NJT > 0) {
            // P[cell][feature] = sum_genes T[cell][gene] Xg[gene][feature] (the Wg_loc gradient of the tile's 32 cells over
            // this gene block).  v_mfma_f32_16x16x4_f32: the 32 x kgp output is cut into 16 x 16 blocks, wave w owns cell
            // half (w & 1) and feature blocks (w >> 1) + 2 n -- every wave contracts over all 256 genes itself, so there
            // is no cross-wave fold and the result goes straight to the per-cell statistics in HBM.
            // Lane (l15, kq) contracts genes 16 G + 4 kq + j in MFMA step j of gene group G, so its A operands of four
            // steps are ONE ds_read_b128 of the residual row and its B operands one ds_read_b128 of the Xg row.
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int l15 = fresh(lane) & 15, kq = fresh(lane) >> 4;
            const int ch = w & 1;
            const float *arow = T + (ch * 16 + l15) * kTileStride + 4 * kq;      // A[i = cell][k = gene]
            f32x4 P[NJT][2];
            const float *bcol[NJT];
            bool fok[NJT];
#pragma unroll
            for (int n = 0; n < NJT; ++n) {
                const int f = ((w >> 1) + 2 * n) * 16 + l15;
                fok[n] = f < t.kgp;
                bcol[n] = xl + (fok[n] ? f : 0) * kXgStride + 4 * kq;            // B[k = gene][j = feature]
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) P[n][u][q] = 0.0f;
            }
            // (a feature block past kgp reads feature 0: its columns of P are garbage of its own and are not stored)
            auto fetch = [&](int g, F4 (&av)[2], F4 (&bv)[2][NJT]) {             // two gene groups per batch of LDS reads
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    av[u] = ld4(arow + g + 16 * u);
#pragma unroll
                    for (int n = 0; n < NJT; ++n) bv[u][n] = ld4(bcol[n] + g + 16 * u);
                }
            };
            F4 av[2], bv[2][NJT];
            fetch(0, av, bv);
#pragma unroll 1
            for (int g = 0; g < kGenesPerBlock; g += 32) {
                F4 nav[2], nbv[2][NJT];
                fetch((g + 32) & (kGenesPerBlock - 1), nav, nbv);                // (the last batch reads batch 0 again and drops it)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < NJT; ++n)            // two accumulator chains per block hide the MFMA latency
                            P[n][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].v[j], bv[u][n].v[j], P[n][j & 1], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    av[u] = nav[u];
#pragma unroll
                    for (int n = 0; n < NJT; ++n) bv[u][n] = nbv[u][n];
                }
            }
            float *chunk = t.row_partials + static_cast<int64_t>(gb) * (t.kgp + 2) * a.Nc;
#pragma unroll
            for (int n = 0; n < NJT; ++n) {
                const int f = ((w >> 1) + 2 * n) * 16 + l15;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                    // D[i][j]: j = lane & 15, i = 4 (lane >> 4) + q
                    const int cellr = tr0 + ch * 16 + 4 * kq + q;
                    if (fok[n] && cellr < row_end) chunk[static_cast<int64_t>(cellr) * t.kgp + f] = P[n][0][q] + P[n][1][q];
                }
            }
        }
        if (tr0 + kTileRows < row_end) fetch_design(tr0 + kTileRows);    // lands while the half gathers at the barrier
        BRIE_PROF_MARK(5);
        tile_sync<NH>(ctr, arrived);                    // T and At are free for the next tile
    }
#if BRIE_TILE_PROF
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&tile_prof[w == 0 ? i : 8 + i], prof_[i]);
        if (w == 0) atomicAdd(&tile_prof[7], static_cast<unsigned long long>((row_end - row0 + kTileRows - 1) / kTileRows));
    }
#endif

    // G accumulators -> this chunk's partial sums (summed over chunks in fp64 by wide_w_adam)
    if constexpr (NACC > 0) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int f = (q & 3) + 8 * (q >> 2) + 4 * half + 32 * n;
                    if (f < t.Kc)
                        t.Gpart[(static_cast<int64_t>(chunk_id) * t.Kc + f) * a.ld + gb * kGenesPerBlock + (CB * w + c) * 32 + l31] =
                            G[n][c][q];
                }
    }

    // fold the 4 waves' per-gene partials through LDS (T is free), wave 0 writes the chunk row
    constexpr int SF = S + (REGK ? KCR : 0);        // REGK: the X^T.R rows ride along
    if (w > 0) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int v = 0; v < kVec; ++v) T[((w - 1) * SF + s) * kGenesPerBlock + v * kWave + lane] = acc[s][v];
        if constexpr (REGK) {
#pragma unroll
            for (int k = 0; k < KCR; ++k)
#pragma unroll
                for (int v = 0; v < kVec; ++v) T[((w - 1) * SF + S + k) * kGenesPerBlock + v * kWave + lane] = accW[k][v];
        }
    }
    tile_sync<NH>(ctr, arrived);
    if (w == 0 && active) {
        float *dst = partials + (static_cast<int64_t>(chunk_id) * S) * a.ld + j0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            F4 o;
#pragma unroll
            for (int v = 0; v < kVec; ++v) {
                float tt = acc[s][v];
#pragma unroll
                for (int ww = 0; ww < NW - 1; ++ww) tt += T[(ww * SF + s) * kGenesPerBlock + v * kWave + lane];
                o.v[v] = tt;
            }
            st4(dst + s * a.ld, o);
        }
        if constexpr (REGK) {               // same place and meaning as the MFMA path's G: this chunk's X^T.R rows
#pragma unroll
            for (int k = 0; k < KCR; ++k) {
                if (k >= t.Kc) break;
                F4 o;
#pragma unroll
                for (int v = 0; v < kVec; ++v) {
                    float tt = accW[k][v];
#pragma unroll
                    for (int ww = 0; ww < NW - 1; ++ww) tt += T[(ww * SF + S + k) * kGenesPerBlock + v * kWave + lane];
                    o.v[v] = tt;
                }
                st4(t.Gpart + (static_cast<int64_t>(chunk_id) * t.Kc + k) * a.ld + j0, o);
            }
        }
    }
}

}  // namespace brie

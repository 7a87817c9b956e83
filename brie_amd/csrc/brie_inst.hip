// brie_inst.hip -- instantiates elbo_adam_step / loss_gene_eval for ONE cell-feature count
// (-DBRIE_KC=N) over likelihood mode x MC_size {1, 3, run-time} x count storage {fp32, u8, u16}.
#include <algorithm>
#include <cstdlib>
#include "brie_launch.h"

#ifndef BRIE_KC
#error "compile with -DBRIE_KC=<0..8>"
#endif
#define BRIE_CAT2(a, b) a##b
#define BRIE_CAT(a, b) BRIE_CAT2(a, b)

namespace brie {

namespace {

// ONE workgroup per CU (4 waves) streams best: the step kernel keeps a whole row of every array in flight per wave, and
// more waves only add concurrent streams to a memory system that is already saturated.  Same handle, occupancy toggled
// every ten steps at the C3 shape (profiles/occ_pad_sweep.py): 3 per CU (what 166 VGPRs allow) 8.26 ms, 2 per CU 8.17 -
// 8.21 ms, 1 per CU 8.07 - 8.11 ms on a fast box; 9.37 -> 9.12 ms on a box in its slow mode (r04j / r04k logs).  Capping
// through the register allocator (amdgpu_waves_per_eu) re-schedules every instantiation (MC_size 3 with effLen: 13 %
// slower, profiles/history/r03n_ab_max_waves.log), so the launch simply carries enough unused dynamic LDS that a second
// workgroup no longer fits the CU's 160 KB.  BRIE_STEP_OCCUPANCY_CAP=0: hardware occupancy, =1 / =2: one / two per CU (A/B runs;
// read once, or at every launch when BRIE_STEP_OCCUPANCY_CAP_DYNAMIC is set).
struct OccupancyPads { int one = 0, two = 0; };
template <typename Kern>
OccupancyPads occupancy_pads(Kern kern) {
    OccupancyPads p;
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(kern)) != hipSuccess) return p;
    const int stat = static_cast<int>(at.sharedSizeBytes);
    p.one = std::max(0, 81 * 1024 - stat);               // 2 x (static + pad) > 160 KB
    p.two = std::max(0, 54 * 1024 - stat);               // 3 x (static + pad) > 160 KB
    return p;
}
inline int occupancy_cap() {
    static const bool dynamic = getenv("BRIE_STEP_OCCUPANCY_CAP_DYNAMIC") != nullptr;
    auto read = []() { const char *e = getenv("BRIE_STEP_OCCUPANCY_CAP"); return e ? atoi(e) : -1; };     // -1: automatic
    static const int fixed = read();
    return dynamic ? read() : fixed;
}

template <int MODE, int MC, int CS, bool CPL>
void step_launch(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    auto kern = elbo_adam_step<BRIE_KC, MODE, MC, CS, CPL>;
    static const OccupancyPads pads = occupancy_pads(kern);
    // automatic: one per CU for one Monte-Carlo sample per step; with more samples the row body keeps the SIMD's VALU
    // busy most of the time and a second wave per SIMD is what hides the memory latency (MC_size 3: one per CU is 4 - 15 %
    // SLOWER than two for Kc = 1, 3, 8; profiles/history/r04o_occ_matrix.log)
    int cap = occupancy_cap();
    if (cap < 0) cap = MC == 1 ? 1 : 2;
    const int pad = cap == 1 ? pads.one : (cap == 2 ? pads.two : 0);
    if (pad > 64 * 1024)        // per launch, like the tile kernel: the attribute belongs to the current device
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pad);
    hipLaunchKernelGGL(kern, c.grid, dim3(kBlock), pad, c.stream, q.c1, q.c2,
                       q.c3, q.mu, q.rho, q.m_mu, q.v_mu, q.m_rho, q.v_rho, q.Xc, q.W, q.b, q.lam, q.effL,
                       q.partials, a, cp, static_cast<float *>(nullptr));
}

// Kg > 4: coupled variant with the gene block's Xg tile in dynamic LDS (up to 64 KiB on top of the static arrays)
template <int MODE, int CS>
void step_launch_gw(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    auto kern = elbo_adam_step<BRIE_KC, MODE, 0, CS, true, false, true>;
    // per device and cheap; the coupled variants are not launch-bound
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kKgWideMax * kGenesPerBlock * static_cast<int>(sizeof(float)));
    hipLaunchKernelGGL(kern, c.grid, dim3(kBlock), c.gw_lds_bytes, c.stream, q.c1, q.c2, q.c3, q.mu, q.rho, q.m_mu, q.v_mu,
                       q.m_rho, q.v_rho, q.Xc, q.W, q.b, q.lam, q.effL, q.partials, a, cp, nullptr);
}

// target="marginLik" for the coupled variants (the uncoupled ones use margin_step)
template <int MODE, int CS>
void step_launch_margin(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    if (c.gw_lds_bytes > 0) {
        auto kern = elbo_adam_step<BRIE_KC, MODE, 0, CS, true, false, true, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kKgWideMax * kGenesPerBlock * static_cast<int>(sizeof(float)));
        hipLaunchKernelGGL(kern, c.grid, dim3(kBlock), c.gw_lds_bytes, c.stream, q.c1, q.c2, q.c3, q.mu, q.rho, q.m_mu,
                           q.v_mu, q.m_rho, q.v_rho, q.Xc, q.W, q.b, q.lam, q.effL, q.partials, a, cp, nullptr);
    } else {
        hipLaunchKernelGGL((elbo_adam_step<BRIE_KC, MODE, 0, CS, true, false, false, true>), c.grid, dim3(kBlock), 0,
                           c.stream, q.c1, q.c2, q.c3, q.mu, q.rho, q.m_mu, q.v_mu, q.m_rho, q.v_rho, q.Xc, q.W, q.b, q.lam,
                           q.effL, q.partials, a, cp, nullptr);
    }
}

template <int MODE, int CS>
void step_mc(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    if (c.coupled && c.margin) { step_launch_margin<MODE, CS>(c, q, a, cp); return; }
    // coupled modes (gene features / per-cell intercept): one variant with run-time MC_size
    if (c.coupled && c.gw_lds_bytes > 0) { step_launch_gw<MODE, CS>(c, q, a, cp); return; }
    if (c.coupled) { step_launch<MODE, 0, CS, true>(c, q, a, cp); return; }
    // MC_size 1 = API default (model_TFProb.py:130), 3 = CLI default (bin/quant.py:173)
    if (a.mc == 1) step_launch<MODE, 1, CS, false>(c, q, a, cp);
    else if (a.mc == 3) step_launch<MODE, 3, CS, false>(c, q, a, cp);
    else step_launch<MODE, 0, CS, false>(c, q, a, cp);
}

template <int MODE>
void step_cs(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    if (c.cs == kCountU8) step_mc<MODE, kCountU8>(c, q, a, cp);
    else if (c.cs == kCountMixed) step_mc<MODE, kCountMixed>(c, q, a, cp);
    else if (c.cs == kCountU16) step_mc<MODE, kCountU16>(c, q, a, cp);
    else step_mc<MODE, kCountF32>(c, q, a, cp);
}

// Many steps per launch (PERSIST variant, brie_kernels.hip.h): every workgroup of the grid must be resident -- they wait for
// each other -- so the launch is made only when the occupancy the runtime reports for this instantiation holds the whole grid
// (returns 0 otherwise and the caller takes the two-launch path).  MC_size 1 and 3 (the API's and brie-quant's defaults).
template <int MODE, int MC, int CS>
int persist_launch(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const PersistArgs *dev_args, int n_cus) {
    auto kern = elbo_adam_fused_steps<BRIE_KC, MODE, MC, CS>;
    static const int per_cu = [&] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), kBlock, 0) != hipSuccess) nb = 0;
        (void)hipGetLastError();
        return nb;
    }();
    // (c.grid.x may be 8 columns for fewer gene blocks -- one XCD per gene block, brie_capi.hip run_steps: a column's workgroups
    // must then fit the CUs of ONE XCD)
    if (static_cast<int64_t>(c.grid.x) * c.grid.y > static_cast<int64_t>(per_cu) * n_cus) return 0;
    if (c.persist_columns > 0 && static_cast<int64_t>(c.grid.y) > static_cast<int64_t>(n_cus / c.persist_columns) * per_cu) return 0;
    hipLaunchKernelGGL(kern, c.grid, dim3(kBlock), 0, c.stream, q.c1, q.c2, q.c3, q.mu, q.rho, q.m_mu, q.v_mu, q.m_rho, q.v_rho,
                       q.Xc, q.W, q.b, q.lam, q.effL, q.partials, a, CoupledArgs{},
                       reinterpret_cast<float *>(const_cast<PersistArgs *>(dev_args)));
    return 1;
}
template <int MODE, int CS>
int persist_mc(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const PersistArgs *d, int n_cus) {
    if (a.mc == 1) return persist_launch<MODE, 1, CS>(c, q, a, d, n_cus);
    if (a.mc == 3) return persist_launch<MODE, 3, CS>(c, q, a, d, n_cus);
    return 0;
}
template <int MODE>
int persist_cs(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const PersistArgs *d, int n_cus) {
    if (c.cs == kCountU8) return persist_mc<MODE, kCountU8>(c, q, a, d, n_cus);
    if (c.cs == kCountMixed) return persist_mc<MODE, kCountMixed>(c, q, a, d, n_cus);
    if (c.cs == kCountU16) return 0;            // (every quad with counts > 255: not instantiated, the two-launch path)
    return persist_mc<MODE, kCountF32>(c, q, a, d, n_cus);
}

template <int MODE, int CS>
void lg_launch(const LaunchCfg &c, const LossGeneArgs &a) {
    hipLaunchKernelGGL((loss_gene_eval<BRIE_KC, MODE, CS>), c.grid, dim3(kBlock), 0, c.stream, a);
}

template <int MODE>
void lg_cs(const LaunchCfg &c, const LossGeneArgs &a) {
    if (c.cs == kCountU8) lg_launch<MODE, kCountU8>(c, a);
    else if (c.cs == kCountMixed) lg_launch<MODE, kCountMixed>(c, a);
    else if (c.cs == kCountU16) lg_launch<MODE, kCountU16>(c, a);
    else lg_launch<MODE, kCountF32>(c, a);
}

template <int MODE, int CS>
void margin_launch(const LaunchCfg &c, const StepPointers &q, const StepScalars &a) {
    hipLaunchKernelGGL((margin_step<BRIE_KC, MODE, CS>), c.grid, dim3(kBlock), 0, c.stream, q.c1, q.c2, q.c3, q.Xc, q.W,
                       q.b, q.lam, q.effL, q.partials, a);
}

template <int MODE>
void margin_cs(const LaunchCfg &c, const StepPointers &q, const StepScalars &a) {
    if (c.cs == kCountU8) margin_launch<MODE, kCountU8>(c, q, a);
    else if (c.cs == kCountMixed) margin_launch<MODE, kCountMixed>(c, q, a);
    else if (c.cs == kCountU16) margin_launch<MODE, kCountU16>(c, q, a);
    else margin_launch<MODE, kCountF32>(c, q, a);
}

}  // namespace

void BRIE_CAT(launch_margin_kc, BRIE_KC)(const LaunchCfg &c, const StepPointers &q, const StepScalars &a) {
    switch (c.mode) {
        case kLik2: margin_cs<kLik2>(c, q, a); break;
        case kLikEff2: margin_cs<kLikEff2>(c, q, a); break;
        default: margin_cs<kLikEff3>(c, q, a); break;
    }
}

void BRIE_CAT(launch_step_kc, BRIE_KC)(const LaunchCfg &c, const StepPointers &q, const StepScalars &a,
                                       const CoupledArgs &cp) {
    switch (c.mode) {
        case kLik2: step_cs<kLik2>(c, q, a, cp); break;
        case kLikEff2: step_cs<kLikEff2>(c, q, a, cp); break;
        default: step_cs<kLikEff3>(c, q, a, cp); break;
    }
}

int BRIE_CAT(launch_step_persist_kc, BRIE_KC)(const LaunchCfg &c, const StepPointers &q, const StepScalars &a,
                                              const PersistArgs *dev_args, int n_cus) {
    switch (c.mode) {
        case kLik2: return persist_cs<kLik2>(c, q, a, dev_args, n_cus);
        case kLikEff2: return persist_cs<kLikEff2>(c, q, a, dev_args, n_cus);
        default: return persist_cs<kLikEff3>(c, q, a, dev_args, n_cus);
    }
}

void BRIE_CAT(launch_loss_gene_kc, BRIE_KC)(const LaunchCfg &c, const LossGeneArgs &a) {
    switch (c.mode) {
        case kLik2: lg_cs<kLik2>(c, a); break;
        case kLikEff2: lg_cs<kLikEff2>(c, a); break;
        default: lg_cs<kLikEff3>(c, a); break;
    }
}

#if BRIE_KC == 0
namespace {
template <int MODE, int CS>
void wide_mc(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
#define BRIE_WIDE_M(MC, CPL, GW, MRG, LDS)                                                                         \
    hipLaunchKernelGGL((elbo_adam_step<0, MODE, MC, CS, CPL, true, GW, MRG>), c.grid, dim3(kBlock), LDS, c.stream, q.c1, \
                       q.c2, q.c3, q.mu, q.rho, q.m_mu, q.v_mu, q.m_rho, q.v_rho, q.Xc, q.W, q.b, q.lam, q.effL, \
                       q.partials, a, cp, c.rbuf)
#define BRIE_WIDE(MC, CPL, GW, LDS) BRIE_WIDE_M(MC, CPL, GW, false, LDS)
    if (c.margin) {                                  // target="marginLik": prior samples, residual q into rbuf
        if (c.coupled && c.gw_lds_bytes > 0) {
            auto kern = elbo_adam_step<0, MODE, 0, CS, true, true, true, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kKgWideMax * kGenesPerBlock * static_cast<int>(sizeof(float)));
            BRIE_WIDE_M(0, true, true, true, c.gw_lds_bytes);
        } else if (c.coupled) BRIE_WIDE_M(0, true, false, true, 0);
        else BRIE_WIDE_M(0, false, false, true, 0);
        return;
    }
    if (c.coupled && c.gw_lds_bytes > 0) {          // wide cell design + Kg > 4: W tile (static) + Xg tile (dynamic) in LDS
        auto kern = elbo_adam_step<0, MODE, 0, CS, true, true, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kKgWideMax * kGenesPerBlock * static_cast<int>(sizeof(float)));
        BRIE_WIDE(0, true, true, c.gw_lds_bytes);
    } else if (c.coupled) BRIE_WIDE(0, true, false, 0);
    else if (a.mc == 1) BRIE_WIDE(1, false, false, 0);
    else if (a.mc == 3) BRIE_WIDE(3, false, false, 0);
    else BRIE_WIDE(0, false, false, 0);
#undef BRIE_WIDE
#undef BRIE_WIDE_M
}
template <int MODE>
void wide_cs(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    if (c.cs == kCountU8) wide_mc<MODE, kCountU8>(c, q, a, cp);
    else if (c.cs == kCountMixed) wide_mc<MODE, kCountMixed>(c, q, a, cp);
    else if (c.cs == kCountU16) wide_mc<MODE, kCountU16>(c, q, a, cp);
    else wide_mc<MODE, kCountF32>(c, q, a, cp);
}
}  // namespace

void launch_step_wide(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const CoupledArgs &cp) {
    switch (c.mode) {
        case kLik2: wide_cs<kLik2>(c, q, a, cp); break;
        case kLikEff2: wide_cs<kLikEff2>(c, q, a, cp); break;
        default: wide_cs<kLikEff3>(c, q, a, cp); break;
    }
}
#endif

}  // namespace brie

// brie_tile_inst.hip -- instantiates elbo_adam_step_tile for ONE likelihood mode (-DBRIE_TILE_MODE=0..2) over
// count storage {fp32, u8, u16} x accumulator sets NACC {0,1,2} x per-cell output tiles NJT {0,1,2}.
#include "brie_launch.h"
#include "brie_tile.hip.h"

#ifndef BRIE_TILE_MODE
#error "compile with -DBRIE_TILE_MODE=<0..2>"
#endif
#define BRIE_CAT2(a, b) a##b
#define BRIE_CAT(a, b) BRIE_CAT2(a, b)

namespace brie {
namespace {

template <int CS, int NACC, int NJT, int NH, int KCR = 0>
void tile_launch(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const TileArgs &t, int lds_bytes) {
    auto kern = elbo_adam_step_tile<BRIE_TILE_MODE, CS, NACC, NJT, NH, KCR>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    dim3 grid = c.grid;
    grid.y = (grid.y + NH - 1) / NH;           // every half of a workgroup owns one cell chunk
    hipLaunchKernelGGL(kern, grid, dim3(NH * kBlock), lds_bytes, c.stream, q.c1, q.c2, q.c3, q.mu, q.rho, q.m_mu, q.v_mu,
                       q.m_rho, q.v_rho, q.b, q.lam, q.effL, q.partials, a, t);
}

template <int CS, int NACC, int NJT, int KCR = 0>
void tile_nw(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const TileArgs &t, int nw, int lds) {
    if (nw == 2) tile_launch<CS, NACC, NJT, 2, KCR>(c, q, a, t, lds);
    else tile_launch<CS, NACC, NJT, 1, KCR>(c, q, a, t, lds);
}

template <int CS, int NACC>
void tile_njt(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const TileArgs &t, int njt, int nw, int kcr,
              int lds) {
    if (njt == 0) {
        if constexpr (NACC > 0) tile_nw<CS, NACC, 0>(c, q, a, t, nw, lds);
    } else if constexpr (NACC == 0) {       // gene features; <= 4 cell features in registers (kcr), else none
        if (kcr) {
            if (njt == 1) tile_nw<CS, 0, 1, 4>(c, q, a, t, nw, lds);
            else tile_nw<CS, 0, 2, 4>(c, q, a, t, nw, lds);
        } else if (njt == 1) tile_nw<CS, 0, 1>(c, q, a, t, nw, lds);
        else tile_nw<CS, 0, 2>(c, q, a, t, nw, lds);
    } else if (njt == 1) tile_nw<CS, NACC, 1>(c, q, a, t, nw, lds);
    else tile_nw<CS, NACC, 2>(c, q, a, t, nw, lds);
}

template <int CS>
void tile_nacc(const LaunchCfg &c, const StepPointers &q, const StepScalars &a, const TileArgs &t, int nacc, int njt, int nw,
               int kcr, int lds) {
    if (nacc == 0) tile_njt<CS, 0>(c, q, a, t, njt, nw, kcr, lds);
    else if (nacc == 1) tile_njt<CS, 1>(c, q, a, t, njt, nw, 0, lds);
    else tile_njt<CS, 2>(c, q, a, t, njt, nw, 0, lds);
}

}  // namespace

void BRIE_CAT(launch_tile_mode, BRIE_TILE_MODE)(const LaunchCfg &c, const StepPointers &q, const StepScalars &a,
                                                const TileArgs &t, int nacc, int njt, int nw, int kcr, int lds_bytes) {
    if (c.cs == kCountU8) tile_nacc<kCountU8>(c, q, a, t, nacc, njt, nw, kcr, lds_bytes);
    else if (c.cs == kCountMixed) tile_nacc<kCountMixed>(c, q, a, t, nacc, njt, nw, kcr, lds_bytes);
    else if (c.cs == kCountU16) tile_nacc<kCountU16>(c, q, a, t, nacc, njt, nw, kcr, lds_bytes);
    else tile_nacc<kCountF32>(c, q, a, t, nacc, njt, nw, kcr, lds_bytes);
}

}  // namespace brie

#if BRIE_TILE_PROF && BRIE_TILE_MODE == 0
// tuning builds only: cycles per phase summed over the waves 0 (slots 0..6, slot 7 = tiles) and the other waves (8..14)
extern "C" int brie_debug_tile_prof(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(brie::tile_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(brie::tile_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

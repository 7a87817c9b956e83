// brie_launch.h -- launch entry points of the two template-heavy kernels.  The (KC, MODE, MC,
// count-storage) instantiations are split over one translation unit per KC (brie_inst.hip built
// with -DBRIE_KC=0..8) so that hipcc compiles them in parallel; brie_capi.hip only dispatches.
#pragma once
#include <hip/hip_runtime.h>
#include "brie_kernels.hip.h"

namespace brie {

struct StepPointers {
    const void *c1, *c2, *c3;                          // fp32 or u8 count layers
    float *mu, *rho, *m_mu, *v_mu, *m_rho, *v_rho;
    const float *Xc, *W, *b, *lam, *effL;
    float *partials;
};

struct LaunchCfg {
    int mode;      // kLik2 / kLikEff2 / kLikEff3
    int cs;        // kCountF32 / kCountU8 / kCountU16
    dim3 grid;
    hipStream_t stream;
    int coupled;   // 1: gene features and/or per-cell intercept (CoupledArgs valid)
    float *rbuf = nullptr;         // wide designs (Kc > 8): residual buffer read back by wide_design_grad
    int gw_lds_bytes = 0;          // > 0: Kg > 4, GW variant with an Xg tile of this many bytes in dynamic LDS
    int margin = 0;                // 1: target="marginLik" through the MARGIN variants (coupled / wide models)
    int persist_columns = 0;       // many steps per launch: > 0 = grid.x is this many columns (PersistArgs::columns), not gene blocks
};

#define BRIE_DECLARE_KC(N)                                                                          \
    void launch_step_kc##N(const LaunchCfg &, const StepPointers &, const StepScalars &,           \
                           const CoupledArgs &);                                                    \
    int launch_step_persist_kc##N(const LaunchCfg &, const StepPointers &, const StepScalars &,    \
                                  const PersistArgs *dev_args, int n_cus);                          \
    void launch_loss_gene_kc##N(const LaunchCfg &, const LossGeneArgs &);                           \
    void launch_margin_kc##N(const LaunchCfg &, const StepPointers &, const StepScalars &);
BRIE_DECLARE_KC(0) BRIE_DECLARE_KC(1) BRIE_DECLARE_KC(2) BRIE_DECLARE_KC(3) BRIE_DECLARE_KC(4)
BRIE_DECLARE_KC(5) BRIE_DECLARE_KC(6) BRIE_DECLARE_KC(7) BRIE_DECLARE_KC(8)
#undef BRIE_DECLARE_KC
// wide designs on the matrix cores (brie_tile.hip.h), one translation unit per likelihood mode
struct TileArgs;
void launch_tile_mode0(const LaunchCfg &, const StepPointers &, const StepScalars &, const TileArgs &, int nacc, int njt, int nw,
                       int kcr, int lds);
void launch_tile_mode1(const LaunchCfg &, const StepPointers &, const StepScalars &, const TileArgs &, int nacc, int njt, int nw,
                       int kcr, int lds);
void launch_tile_mode2(const LaunchCfg &, const StepPointers &, const StepScalars &, const TileArgs &, int nacc, int njt, int nw,
                       int kcr, int lds);
// wide cell designs: KC == 0 kernels with the W tile in LDS, writing the residual buffer (defined in the kc0 unit)
void launch_step_wide(const LaunchCfg &, const StepPointers &, const StepScalars &, const CoupledArgs &);

}  // namespace brie

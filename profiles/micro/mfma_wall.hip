// Wall-clock rate of v_mfma_f32_32x32x2_f32 over the WHOLE chip, bare (8 independent accumulator sets per wave, 2 waves per
// SIMD, no memory traffic), with one ds_read_b32 per MFMA, and with the barrier / LDS-write / global-load pattern of a stage
// of fused_prior_mean (156 / 144 / 145 / 141 / 128 TFLOP/s, profiles/r5/r5w_mfma_wall.jsonl) -- the ceiling the panel kernels of DESIGN 4.4 are held against:
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_wall.hip -o /tmp/mfma_wall && /tmp/mfma_wall
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// MODE 0: bare; 1: one ds_read_b32 per MFMA; 2: that + a workgroup barrier every 128 MFMAs per wave (a stage of
// fused_prior_mean); 3: that + 20 LDS writes per thread before every barrier (the stage's tiles being stashed)
// 4: that + the stage's global loads (4 x 16 B + 16 x 4 B per lane, issued at the start of the stage, written to LDS at its end)
template <int MODE>
__global__ __launch_bounds__(512) void k(float *o, int n, float a, float b, const float *src) {
    constexpr bool LDS = MODE >= 1;
    __shared__ float sh[64 * 288];
    for (int i = threadIdx.x; i < 64 * 288; i += 512) sh[i] = b;
    __syncthreads();
    f32x16 D[8];
    for (int c = 0; c < 8; ++c) for (int q = 0; q < 16; ++q) D[c][q] = 0.f;
    f32x4v g4[4] = {};
    float g1[16] = {};
    const float *bl = sh + (threadIdx.x & 63);
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) D[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, LDS ? bl[((i & 7) * 8 + c) * 288] : b, D[c], 0, 0, 0);
        if (MODE >= 4 && (i & 15) == 0) {
            const unsigned base = (blockIdx.x * 977u + (i >> 4) * 131u) % 4096u;
#pragma unroll
            for (int j = 0; j < 4; ++j) g4[j] = *reinterpret_cast<const f32x4v *>(src + ((base + j) * 2048u + threadIdx.x * 4u));
#pragma unroll
            for (int j = 0; j < 16; ++j) g1[j] = src[(base + 8 + j) * 2048u + (threadIdx.x >> 5) * 129u + (threadIdx.x & 31)];
        }
        if (MODE >= 2 && (i & 15) == 15) {
            if (MODE >= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v *>(sh + (j * 512 + threadIdx.x) * 4) = g4[j];
#pragma unroll
                for (int j = 0; j < 16; ++j) sh[8192 + j * 512 + threadIdx.x] = g1[j];
            } else if (MODE >= 3)
                for (int j = 0; j < 20; ++j) sh[(j * 512 + threadIdx.x) % (64 * 288)] = b + D[0][0] * 0.0f;
            __syncthreads();
        }
    }
    float s = 0;
    for (int c = 0; c < 8; ++c) for (int q = 0; q < 16; ++q) s += D[c][q];
    o[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float *o;
    hipMalloc(&o, 4096 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float *src;
    hipMalloc(&src, 4200u * 2048u * sizeof(float));
    hipMemset(src, 0, 4200u * 2048u * sizeof(float));
    for (int lds = 0; lds < 5; ++lds)
        for (int blocks : {256, 2048}) {
            for (int n : {8000, 32000}) {               // ~3.5 ms .. ~14 ms per launch at 256 blocks
                float best = 1e30f, last = 0;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    if (lds == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, o, n, 1.0f, 0.5f, src);
                    else if (lds == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, o, n, 1.0f, 0.5f, src);
                    else if (lds == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, o, n, 1.0f, 0.5f, src);
                    else if (lds == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, o, n, 1.0f, 0.5f, src);
                    else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, o, n, 1.0f, 0.5f, src);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&last, e0, e1);
                    if (last < best) best = last;
                }
                const double flop = 2.0 * 32 * 32 * 2 * 8.0 * n * 8 /*waves*/ * blocks;
                printf("{\"mode\": %d, \"blocks\": %d, \"mfma_per_wave\": %d, \"ms_best\": %.3f, \"ms_last\": %.3f, \"TFLOPs\": %.1f}\n",
                       lds, blocks, 8 * n, best, last, flop / (best * 1e-3) / 1e12);
            }
        }
    return 0;
}

// Issue rate of the fp32 MFMAs the tile kernel uses, with 1 / 2 / 4 independent accumulator chains per wave and 1 / 2 waves
// per SIMD:   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void k32(float *o, int n, float a, float b) {
    f32x16 D[CH];
    for (int c = 0; c < CH; ++c) for (int q = 0; q < 16; ++q) D[c][q] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < CH; ++c) D[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, D[c], 0, 0, 0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CH; ++c) for (int q = 0; q < 16; ++q) s += D[c][q];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) o[0] = static_cast<float>(t1 - t0) / (n * CH);
}
template <int CH>
__global__ void k16(float *o, int n, float a, float b) {
    f32x4 D[CH];
    for (int c = 0; c < CH; ++c) for (int q = 0; q < 4; ++q) D[c][q] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < CH; ++c) D[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, D[c], 0, 0, 0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CH; ++c) for (int q = 0; q < 4; ++q) s += D[c][q];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) o[0] = static_cast<float>(t1 - t0) / (n * CH);
}
template <typename K>
void run(const char *name, K kern, int waves_per_simd, float *o) {
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * waves_per_simd), 0, 0, o, n, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms, c; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&c, o, 4, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD %d: %.1f counter ticks per MFMA (wave 0), kernel %.3f ms\n", name, waves_per_simd, c, ms);
}
int main() {
    float *o; hipMalloc(&o, 256 * 512 * 4);
    for (int w = 1; w <= 2; ++w) {
        run("32x32x2 f32, 1 chain", k32<1>, w, o);
        run("32x32x2 f32, 2 chains", k32<2>, w, o);
        run("32x32x2 f32, 4 chains", k32<4>, w, o);
        run("16x16x4 f32, 1 chain", k16<1>, w, o);
        run("16x16x4 f32, 2 chains", k16<2>, w, o);
        run("16x16x4 f32, 4 chains", k16<4>, w, o);
    }
    return 0;
}

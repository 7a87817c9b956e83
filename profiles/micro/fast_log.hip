// Which bare-v_log_f32 formula returns the same fp32 value as __logf(x) for every normal x?  (f_log in brie_kernels.hip.h)
// a: v_log_f32(x) * ln2;  b: the extended-precision product y*c + (fma(y, c, -y*c) + y*cc) the compiler's own lowering of
// logf uses (c = 0x1.62e42ep-1, cc = 0x1.efa39ep-25), without its denormal-input rescue and infinity check.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/fast_log.hip -o /tmp/fast_log && /tmp/fast_log
// Round 5, ROCm 7.2.0 hipcc on gfx950: a differs for 688 358 784 and b for 690 802 480 of the 2 130 706 432 positive normal values,
// each by one ulp: the compiler contracts the last add of its own lowering into fma(y, c, .) (one rounding), formula b keeps
// the two roundings the lowering is written with.  b is therefore "within one ulp of __logf", not identical to it (the header
// of round 3 said identical; corrected in brie_kernels.hip.h::f_log_sel).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float log_b(float x) {
#pragma clang fp contract(off)
    const float y = __builtin_amdgcn_logf(x);
    const float c = 0x1.62e42ep-1f, cc = 0x1.efa39ep-25f;
    const float r = y * c;
    return r + __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -r));
}
__global__ void cmp(unsigned long long *bad, uint32_t lo, uint32_t hi) {
    // every fp32 bit pattern in [lo, hi) -- all positive normal numbers
    for (uint64_t b = lo + blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; b < hi;
         b += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float x = __uint_as_float(static_cast<uint32_t>(b));
        const float a = __logf(x), r = __builtin_amdgcn_logf(x) * 0.6931471805599453f;
        if (__float_as_uint(a) != __float_as_uint(r)) atomicAdd(bad, 1ull);
        if (__float_as_uint(a) != __float_as_uint(log_b(x))) atomicAdd(bad + 1, 1ull);
    }
}
int main() {
    unsigned long long *bad, h[2] = {0, 0};
    hipMalloc(&bad, 16);
    hipMemset(bad, 0, 16);
    hipLaunchKernelGGL(cmp, dim3(4096), dim3(256), 0, 0, bad, 0x00800000u, 0x7F800000u);
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("positive normal fp32 values where __logf(x) differs from  a: %llu  b: %llu  of %u\n", h[0], h[1], 0x7F800000u - 0x00800000u);
    return h[1] != 0;
}

// Micro-benchmark: Philox round multiplies as v_mul_hi_u32 + v_mul_lo_u32 versus one v_mad_u64_u32.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/philox_mul.hip -o gpurun_out/philox_mul && gpurun_out/philox_mul
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int WIDE>
__global__ void rounds(uint32_t *o, uint32_t a, int n) {
    uint32_t c0 = threadIdx.x + blockIdx.x * blockDim.x, c1 = 1, c2 = a, c3 = 3, k0 = 5, k1 = 7;
    for (int i = 0; i < n; ++i) {
        uint32_t hi0, lo0, hi1, lo1;
        if (WIDE) {
            const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
            hi0 = static_cast<uint32_t>(p0 >> 32); lo0 = static_cast<uint32_t>(p0);
            hi1 = static_cast<uint32_t>(p1 >> 32); lo1 = static_cast<uint32_t>(p1);
        } else {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        }
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[threadIdx.x + blockIdx.x * blockDim.x] = c0 ^ c1 ^ c2 ^ c3;
}

int main() {
    uint32_t *o;
    const int blocks = 256 * 8, threads = 256, n = 20000;
    hipMalloc(&o, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t h[2][4];
    for (int rep = 0; rep < 3; ++rep)
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (w) hipLaunchKernelGGL(rounds<1>, dim3(blocks), dim3(threads), 0, 0, o, 2u, n);
            else hipLaunchKernelGGL(rounds<0>, dim3(blocks), dim3(threads), 0, 0, o, 2u, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h[w], o, 16, hipMemcpyDeviceToHost);
            // per SIMD: blocks*threads/64 waves over 1024 SIMDs, n rounds each
            const double cyc = ms * 1e-3 * 2.4e9 / (double(blocks) * threads / 64 / 1024 * n);
            printf("%s  %.3f ms  ~%.1f cycles / wave-round @2.4GHz  out %08x\n", w ? "mad_u64_u32" : "mul_hi+mul_lo", ms, cyc, h[w][1]);
        }
    printf("identical: %d\n", h[0][0] == h[1][0] && h[0][1] == h[1][1] && h[0][3] == h[1][3]);
    return 0;
}

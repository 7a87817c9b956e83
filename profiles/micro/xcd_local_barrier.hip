// xcd_local_barrier.hip -- what does a per-step hand-off between the workgroups of ONE gene block cost when they all sit on
// one XCD and talk through ITS L2, against the agent-scope protocol the many-steps-per-launch kernel uses (write-through
// stores, counter in memory, loads past L1 and L2)?  G workgroups of 256 threads; per step every workgroup stores S rows of 256
// floats, all meet at a counter, every thread then sums its column over the G x S rows (what the finalize phase reads).
//   mode 0  agent scope: sc1 stores, s_waitcnt, agent-scope atomic add, agent-scope polling load, sc1 loads  (brie_step_body.inc)
//   mode 1  one XCD:     plain stores (write-through L1 -> L2), s_waitcnt, atomic add WITHOUT sc1 (executes in this XCD's L2),
//                        poll by an L2 atomic (fetch_or 0), `buffer_inv sc0` (drops this CU's L1), plain loads (L2 hits)
// grid = (8, G): workgroup (x, y) has linear id x + 8 y, the dispatcher deals linear ids round robin over the 8 XCDs, so column
// x = `xcd` should be XCD `xcd` -- every workgroup reads HW_REG_XCC_ID and the probe reports the set it saw.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/xcd_local_barrier.hip -o /tmp/xlb && /tmp/xlb <mode> <G> <steps> <spread>
//   spread 0: the G workgroups in ONE column (one XCD); 1: grid (G), linear ids = all XCDs (mode 0's normal placement)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int S = 4;

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float *rows, uint32_t *counter, int G, int steps, int column, int spread, uint32_t *out,
                                                double *sums) {
    if (!spread && static_cast<int>(blockIdx.x) != column) return;
    const int g = spread ? blockIdx.x : blockIdx.y, t = threadIdx.x;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (t == 0) atomicOr(out + 0, 1u << (xcc & 15));
    double total = 0.0;
    for (int it = 0; it < steps; ++it) {
        float *buf = rows + static_cast<int64_t>(it & 1) * G * S * 256;
        for (int s = 0; s < S; ++s) {
            const float v = static_cast<float>(it + 1) + 0.001f * g + s + 0.5f * (t & 7);
            if constexpr (MODE == 0) __hip_atomic_store(buf + (g * S + s) * 256 + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else buf[(g * S + s) * 256 + t] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const uint32_t want = static_cast<uint32_t>(it + 1) * G;
            uint32_t spins = 0;
            if constexpr (MODE == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(0);
            } else {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (;;) {      // (a fetch_or of 0 in C++ is folded into a workgroup-scope LOAD, which may hit this CU's L1: spelled out)
                    uint32_t seen;
                    const uint32_t zero = 0;
                    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(counter), "v"(zero) : "memory");
                    if (seen >= want || ++spins >= (1u << 22)) break;
                    __builtin_amdgcn_s_sleep(0);
                }
            }
            if (spins >= (1u << 22)) atomicAdd(out + 1, 1u);
        }
        __syncthreads();
        if constexpr (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
        double sum = 0.0;
        float xv[S][16];
        for (int c0 = 0; c0 < G; c0 += 16) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float *p = buf + (min(c0 + u, G - 1) * S + s) * 256 + t;
                    if constexpr (MODE == 0) xv[s][u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else xv[s][u] = *p;
                }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (c0 + u < G) sum += static_cast<double>(xv[s][u]);
        }
        double want_sum = 0.0;
        for (int c = 0; c < G; ++c)
            for (int s = 0; s < S; ++s) want_sum += static_cast<double>(static_cast<float>(it + 1) + 0.001f * c + s + 0.5f * (t & 7));
        if (sum != want_sum) atomicAdd(out + 2, 1u);
        total += sum;
        __syncthreads();
    }
    if (g == 0) sums[t] = total;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, G = argc > 2 ? atoi(argv[2]) : 16, steps = argc > 3 ? atoi(argv[3]) : 2000;
    const int spread = argc > 4 ? atoi(argv[4]) : 0, column = argc > 5 ? atoi(argv[5]) : 3;
    float *rows;
    uint32_t *counter, *out;
    double *sums;
    hipMalloc(reinterpret_cast<void **>(&rows), sizeof(float) * 2 * G * S * 256);
    hipMemset(rows, 0, sizeof(float) * 2 * G * S * 256);
    hipMalloc(reinterpret_cast<void **>(&counter), 4);
    hipMalloc(reinterpret_cast<void **>(&out), 16);
    hipMalloc(reinterpret_cast<void **>(&sums), 8 * 256);
    float best = 1e30f;
    uint32_t h[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(counter, 0, 4);
        hipMemset(out, 0, 16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const dim3 grid = spread ? dim3(G) : dim3(8, G);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, dim3(256), 0, 0, rows, counter, G, steps, column, spread, out, sums);
        else hipLaunchKernelGGL(probe<1>, grid, dim3(256), 0, 0, rows, counter, G, steps, column, spread, out, sums);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        uint32_t hh[4];
        hipMemcpy(hh, out, 16, hipMemcpyDeviceToHost);
        h[0] |= hh[0]; h[1] += hh[1]; h[2] += hh[2];
    }
    printf("mode %d (%s) G %d %s: %.3f us per step, XCC ids seen 0x%x, counter time-outs %u, wrong sums %u\n", mode,
           mode ? "one XCD, through its L2" : "agent scope", G, spread ? "spread over the XCDs" : "one column", 1e3 * best / steps, h[0], h[1], h[2]);
    return 0;
}

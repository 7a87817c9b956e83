// tagged_rows.hip -- does a (value, tag) pair published with a write-through store become visible to a POLLING reader on
// another XCD?  The many-steps-per-launch kernel (brie_step_body.inc, PERSIST) publishes every chunk row of a step; this
// probe isolates the protocol: G workgroups of 256 threads, each step every workgroup writes its row of 256 pairs and then
// reads all G rows until every pair carries the step's tag.  Variants (argv[1]):
//   0  dwordx4 sc1 stores (two pairs per store), no drain, 8-byte sc1 loads            (the candidate)
//   1  as 0 + s_waitcnt vmcnt(0) after the stores
//   2  8-byte agent-scope atomic stores, 8-byte sc1 loads
//   3  as 0, but the readers start polling only after a delay (s_sleep) -- do EARLY reads of a line keep it stale?
//   4  dwordx4 sc0 sc1 stores (system scope), 8-byte sc0 sc1 loads
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/tagged_rows.hip -o /tmp/tagged_rows && /tmp/tagged_rows 0
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *rows, int G, int steps, uint32_t bound, uint32_t *out,
                                                unsigned long long *cycles) {
    const int t = threadIdx.x, g = blockIdx.x;
    const int lane = t & 63, w = t >> 6;
    uint64_t spins_total = 0, worst = 0;
    const long long t0 = wall_clock64();
    for (int it = 0; it < steps; ++it) {
        const uint32_t tag = static_cast<uint32_t>(it) + 1u;
        unsigned long long *buf = rows + static_cast<int64_t>(it & 1) * G * 256;
        if (w == 0) {                               // lane l publishes genes 4l .. 4l+3 of this workgroup's row
            const float tg = __uint_as_float(tag);
            unsigned long long *dst = buf + g * 256 + lane * 4;
            float v[4];
            for (int k = 0; k < 4; ++k) v[k] = static_cast<float>(it * 1000 + g) + 0.25f * k + lane;
            if constexpr (V == 2) {
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(dst + k, (static_cast<unsigned long long>(tag) << 32) | __float_as_uint(v[k]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const floatx4 lo = {v[0], tg, v[1], tg}, hi = {v[2], tg, v[3], tg};
                if constexpr (V == 4) {
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(lo) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + 2), "v"(hi) : "memory");
                } else {
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(lo) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + 2), "v"(hi) : "memory");
                }
                if constexpr (V == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if constexpr (V == 3)
            for (int k = 0; k < 40; ++k) __builtin_amdgcn_s_sleep(127);
        // thread t reads gene t of every row
        uint32_t spins = 0;
        bool ok = false;
        double sum = 0.0;
        while (!ok) {
            uint32_t bad = 0;
            sum = 0.0;
            for (int c = 0; c < G; ++c) {
                unsigned long long x;
                if constexpr (V == 4) {
                    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(buf + c * 256 + t) : "memory");
                } else {
                    x = __hip_atomic_load(buf + c * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                bad |= static_cast<uint32_t>(x >> 32) ^ tag;
                sum += static_cast<double>(__uint_as_float(static_cast<uint32_t>(x)));
            }
            ok = bad == 0;
            if (!ok && ++spins > bound) break;
        }
        spins_total += spins;
        if (spins > worst) worst = spins;
        double want = 0.0;
        for (int c = 0; c < G; ++c) want += static_cast<double>(static_cast<float>(it * 1000 + c) + 0.25f * (t & 3) + (t >> 2));
        if (!ok) { atomicAdd(out + 0, 1u); atomicMax(out + 3, static_cast<uint32_t>(it)); }
        else if (sum != want) atomicAdd(out + 1, 1u);
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    atomicMax(out + 2, static_cast<uint32_t>(worst));
    if (t == 0 && g == 0) { cycles[0] = static_cast<unsigned long long>(t1 - t0); cycles[1] = spins_total; }
}

int main(int argc, char **argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 0, G = argc > 2 ? atoi(argv[2]) : 32, steps = argc > 3 ? atoi(argv[3]) : 2000;
    unsigned long long *rows, *cycles;
    uint32_t *out;
    hipMalloc(reinterpret_cast<void **>(&rows), sizeof(unsigned long long) * 2 * G * 256);
    hipMemset(rows, 0, sizeof(unsigned long long) * 2 * G * 256);
    hipMalloc(reinterpret_cast<void **>(&out), 16);
    hipMemset(out, 0, 16);
    hipMalloc(reinterpret_cast<void **>(&cycles), 16);
    const uint32_t bound = 1u << 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    switch (V) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, rows, G, steps, bound, out, cycles); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, rows, G, steps, bound, out, cycles); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(G), dim3(256), 0, 0, rows, G, steps, bound, out, cycles); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(G), dim3(256), 0, 0, rows, G, steps, bound, out, cycles); break;
        default: hipLaunchKernelGGL(probe<4>, dim3(G), dim3(256), 0, 0, rows, G, steps, bound, out, cycles); break;
    }
    hipEventRecord(e1);
    hipError_t rc = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[4];
    unsigned long long hc[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hc, cycles, 16, hipMemcpyDeviceToHost);
    printf("variant %d G %d steps %d: rc %d, %.3f us per step, timed-out thread-steps %u (last at step %u), wrong sums %u, worst polls %u, "
           "polls of thread 0 %llu\n", V, G, steps, static_cast<int>(rc), 1e3 * ms / steps, h[0], h[3], h[1], h[2], hc[1]);
    return 0;
}

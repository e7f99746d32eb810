// Microbenchmark: do VALU work, LDS traffic and barriers overlap with a saturating store stream?
// 2048 workgroups x 256 threads; each workgroup writes a 1 MiB stream in 128 steps of 8 KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VALU, int BARRIERS, int LDSOPS, bool STORES>
__global__ __launch_bounds__(256) void k(uint8_t* out, int steps, size_t stride, uint32_t seed) {
    __shared__ uint32_t lds[4096 + 16];
    uint8_t* o = out + (size_t)blockIdx.x * stride;
    uint32_t x = threadIdx.x * 2654435761u + seed, y = blockIdx.x + seed;
    for (int i = threadIdx.x; i < 4096 + 16; i += 256) lds[i] = i * seed;
    __syncthreads();
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < LDSOPS; i++) x += lds[(x >> 3) & 4095];
#pragma unroll
        for (int i = 0; i < VALU / 2; i++) { x = (x ^ (x >> 7)) + y; }   // 2 VALU each (v_lshrrev+xor fused? count from ISA)
#pragma unroll
        for (int b = 0; b < BARRIERS; b++) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        u32x4 v = {x, y, x ^ y, (uint32_t)s};
        u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
        if (STORES || x == 0x12345678u) { p[0] = v; p[1] = v; }
        y += x;
    }
}

template <int VALU, int BARRIERS, int LDSOPS, bool STORES>
void run(uint8_t* out, int g, int steps, size_t stride, hipEvent_t a, hipEvent_t b) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        k<VALU, BARRIERS, LDSOPS, STORES><<<g, 256>>>(out, steps, stride, 12345u + rep);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("valu %3d barriers %d ldsops %2d stores %d wgs %4d: %.1f us\n", VALU, BARRIERS, LDSOPS, (int)STORES, g, best * 1e3);
}

int main() {
    const int wgs = 2048, steps = 128;
    const size_t stride = (size_t)steps * 8192 + 256;
    uint8_t* out;
    if (hipMalloc(&out, wgs * stride + 4096) != hipSuccess) return 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    run<0, 0, 0, true>(out, wgs, steps, stride, a, b);
    run<100, 0, 0, true>(out, wgs, steps, stride, a, b);
    run<200, 0, 0, true>(out, wgs, steps, stride, a, b);
    run<200, 0, 0, false>(out, wgs, steps, stride, a, b);
    run<200, 3, 0, true>(out, wgs, steps, stride, a, b);
    run<200, 3, 0, false>(out, wgs, steps, stride, a, b);
    run<200, 3, 12, true>(out, wgs, steps, stride, a, b);
    run<200, 3, 12, false>(out, wgs, steps, stride, a, b);
    run<300, 3, 12, true>(out, wgs, steps, stride, a, b);
    run<300, 3, 12, false>(out, wgs, steps, stride, a, b);
    return 0;
}

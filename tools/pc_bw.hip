// Microbenchmark: persistent workgroups with a dedicated loader wave.  Waves 0..3 only store
// (8 KiB per step) and never wait on the vector-memory counter; wave 4 only loads (4 KiB every
// LOAD_EVERY steps, DEPTH loads in flight) and hands the bytes over through LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4* gcp;
typedef __attribute__((address_space(1))) u32x4* gp;

// One KiB of global memory straight into LDS (no registers): lane l's 16 bytes land at lds_dst + 16 l.
__device__ inline void glds16(uint32_t lds_dst, const uint8_t* gsrc) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int DEPTH, bool IL = false, bool BAR = true, int Q = 4>
__global__ __launch_bounds__(320) void pc(uint8_t* out, const uint8_t* in, int steps, int load_every) {
    __shared__ u32x4 ring[4][256];
    gp o = (gp)(out + (size_t)blockIdx.x * steps * 8192);
    const uint8_t* src = in + (size_t)blockIdx.x * (steps / load_every + DEPTH + 1) * 4096;
    const bool loader = threadIdx.x >= 256;
    const uint32_t l = threadIdx.x & 63;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&ring[0][0];
    u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
    int issued = 0;
    if (loader) {
        for (; issued < DEPTH; issued++)
            for (int q = 0; q < Q; q++)
                glds16(__builtin_amdgcn_readfirstlane(ring_lds + (issued & 3) * 4096 + q * 1024), src + (size_t)issued * 4096 + q * 1024 + l * 16);
    }
    for (int s = 0; s < steps; s++) {
        if (loader && s % load_every == 0) {
            // the oldest outstanding chunk has landed once at most (DEPTH-1)*4 loads are in flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * Q) : "memory");
            for (int q = 0; q < Q; q++)
                glds16(__builtin_amdgcn_readfirstlane(ring_lds + (issued & 3) * 4096 + q * 1024), (IL ? in + ((size_t)issued * gridDim.x + blockIdx.x) * 4096 : src + (size_t)issued * 4096) + q * 1024 + l * 16);
            issued++;
        }
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!loader) {
            if (BAR) { const u32x4 r = ring[(s / load_every) & 3][(threadIdx.x * 7 + s) & 255]; v.y += r.x; }
            gp p = IL ? (gp)(out + ((size_t)s * gridDim.x + blockIdx.x) * 8192) + threadIdx.x * 2 : o + (size_t)s * 512 + threadIdx.x * 2;
            p[0] = v; p[1] = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DEPTH, bool IL = false, bool BAR = true, int Q = 4>
void run(uint8_t* out, const uint8_t* in, int g, int steps, int load_every, hipEvent_t a, hipEvent_t b) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        pc<DEPTH, IL, BAR, Q><<<g, 320>>>(out, in, steps, load_every);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("%s %s loader wave, depth %d, %d KiB every %d steps, wgs %4d: %.1f us  written %.0f GB/s\n", IL ? "interleaved" : "streams    ", BAR ? "barrier/step" : "no barrier  ", DEPTH, Q, load_every, g, best * 1e3,
           (double)g * steps * 8192 / best / 1e6);
}

int main() {
    const int wgs = 2048, steps = 128;
    uint8_t *out, *in;
    if (hipMalloc(&out, (size_t)wgs * steps * 8192) != hipSuccess) return 1;
    if (hipMalloc(&in, (size_t)wgs * (steps + 8) * 4096) != hipSuccess) return 1;
    hipMemset(in, 1, (size_t)wgs * (steps + 8) * 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int g : {2048}) {
        run<1, false, true, 1>(out, in, g, steps, 1, a, b);
        run<1, false, true, 2>(out, in, g, steps, 1, a, b);
        run<2, false, true, 2>(out, in, g, steps, 1, a, b);
        run<4, false, true, 2>(out, in, g, steps, 1, a, b);
        run<1, false, true, 4>(out, in, g, steps, 2, a, b);
        run<2, false, true, 4>(out, in, g, steps, 5, a, b);
    }
    return 0;
}

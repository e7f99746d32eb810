// Microbenchmark: the flat one-tile-per-workgroup form WITH the inter-tile dependencies a
// sequential page imposes: workgroup (tile t, page p) waits for a 64-bit record published by
// (t-1, p) before it loads (start offset), publishes its own early, and waits for a second
// record (running value) before it stores.  Tile-major order: blockIdx = t * pages + p.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline uint64_t poll(const uint64_t* p, uint32_t epoch) {
    uint64_t v;
    do { v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(v >> 32) != epoch);
    return v;
}

// The same without the offset chain (tile starts known in advance) and with a decoupled
// look-back for the running value: a tile publishes its own sum at once (state 1), then walks
// back over its predecessors adding sums until it meets an inclusive prefix (state 2).
template <int VALU>
__global__ __launch_bounds__(256) void lookback(uint8_t* out, const uint8_t* in, uint64_t* carry, int pages, uint32_t epoch) {
    __shared__ uint64_t sh[2];
    const uint32_t b = blockIdx.x, t = b / pages, p = b % pages;
    const size_t tile = (size_t)p * (gridDim.x / pages) + t;
    u32x4 v = {threadIdx.x, b, 3, 4};
    v += *(const u32x4*)(in + tile * 1664 + (threadIdx.x * 16) % 1664);
    uint32_t x = v.x, y = v.y;
#pragma unroll
    for (int i = 0; i < VALU / 2; i++) { x = (x ^ (x >> 7)) + y; }
#pragma unroll
    for (int i = 0; i < VALU / 2; i++) { y = (y ^ (y >> 5)) + x; }
    if (threadIdx.x == 0) {
        const uint32_t mine = y & 0xffffff;
        // record: epoch in the top 32 bits, state in bits 30..31 of the low word, value below
        __hip_atomic_store(&carry[b], ((uint64_t)epoch << 32) | (t == 0 ? 2u << 30 : 1u << 30) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t acc = 0;
        if (t > 0) {
            uint32_t k = b - pages;
            for (;;) {
                const uint64_t r = poll(&carry[k], epoch);
                acc += (uint32_t)r & 0x3fffffff;
                if (((uint32_t)r >> 30) == 2 || k < (uint32_t)pages) break;
                k -= pages;
            }
            __hip_atomic_store(&carry[b], ((uint64_t)epoch << 32) | (2u << 30) | ((acc + mine) & 0x3fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sh[1] = acc;
    }
    __syncthreads();
    v.x = x; v.y = y + (uint32_t)sh[1];
    u32x4* q = (u32x4*)(out + tile * 8192 + threadIdx.x * 32);
    q[0] = v; q[1] = v;
}

template <int VALU>
__global__ __launch_bounds__(256) void chain(uint8_t* out, const uint8_t* in, uint64_t* link, uint64_t* carry, int pages, uint32_t epoch) {
    __shared__ uint64_t sh[2];
    const uint32_t b = blockIdx.x, t = b / pages, p = b % pages;
    const size_t tile = (size_t)p * (gridDim.x / pages) + t; // where this tile's bytes live: page-major memory
    uint64_t start = 0;
    if (threadIdx.x == 0) {
        if (t > 0) start = poll(&link[b - pages], epoch);
        sh[0] = start;
    }
    __syncthreads();
    start = sh[0];
    u32x4 v = {threadIdx.x, (uint32_t)start, 3, 4};
    v += *(const u32x4*)(in + tile * 1664 + (threadIdx.x * 16) % 1664);
    uint32_t x = v.x, y = v.y;
#pragma unroll
    for (int i = 0; i < VALU / 2; i++) { x = (x ^ (x >> 7)) + y; }
    if (threadIdx.x == 0) __hip_atomic_store(&link[b], ((uint64_t)epoch << 32) | (x & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < VALU / 2; i++) { y = (y ^ (y >> 5)) + x; }
    if (threadIdx.x == 0) {
        uint64_t c = 0;
        if (t > 0) c = poll(&carry[b - pages], epoch);
        __hip_atomic_store(&carry[b], ((uint64_t)epoch << 32) | (uint32_t)(c + y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[1] = c;
    }
    __syncthreads();
    v.x = x; v.y = y + (uint32_t)sh[1];
    u32x4* q = (u32x4*)(out + tile * 8192 + threadIdx.x * 32);
    q[0] = v; q[1] = v;
}

int main() {
    const int pages = 2048, tiles = 128;
    const size_t n = (size_t)pages * tiles;
    uint8_t *out, *in; uint64_t *link, *carry;
    if (hipMalloc(&out, n * 8192) != hipSuccess || hipMalloc(&in, n * 1664 + 4096) != hipSuccess) return 1;
    if (hipMalloc(&link, n * 8) != hipSuccess || hipMalloc(&carry, n * 8) != hipSuccess) return 1;
    hipMemset(in, 1, n * 1664 + 4096); hipMemset(link, 0, n * 8); hipMemset(carry, 0, n * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    uint32_t epoch = 0;
    for (int valu = 0; valu <= 0; valu += 100) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            epoch++;
            hipEventRecord(a);
            if (valu == 0) chain<0><<<n, 256>>>(out, in, link, carry, pages, epoch);
            if (valu == 100) chain<100><<<n, 256>>>(out, in, link, carry, pages, epoch);
            if (valu == 200) chain<200><<<n, 256>>>(out, in, link, carry, pages, epoch);
            if (valu == 300) chain<300><<<n, 256>>>(out, in, link, carry, pages, epoch);
            hipEventRecord(b);
            if (hipEventSynchronize(b) != hipSuccess) { printf("failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        printf("chained flat tiles, %3d dependent VALU ops: %.1f us  written %.0f GB/s\n", valu, best * 1e3, n * 8192.0 / best / 1e6);
    }
    for (int valu = 0; valu <= 300; valu += 100) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            epoch++;
            hipEventRecord(a);
            if (valu == 0) lookback<0><<<n, 256>>>(out, in, carry, pages, epoch);
            if (valu == 100) lookback<100><<<n, 256>>>(out, in, carry, pages, epoch);
            if (valu == 200) lookback<200><<<n, 256>>>(out, in, carry, pages, epoch);
            if (valu == 300) lookback<300><<<n, 256>>>(out, in, carry, pages, epoch);
            hipEventRecord(b);
            if (hipEventSynchronize(b) != hipSuccess) { printf("failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        printf("flat tiles + look-back, %3d dependent VALU ops: %.1f us  written %.0f GB/s\n", valu, best * 1e3, n * 8192.0 / best / 1e6);
    }
    return 0;
}

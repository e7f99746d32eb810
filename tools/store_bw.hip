// Microbenchmark: what the store pattern of the DELTA kernel can reach on its own.
// 2048 workgroups x 256 threads; each workgroup writes a 1 MiB stream in 128 steps of 8 KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t* out, const uint8_t* in, int steps, size_t stride, int in_bytes_per_step) {
    uint8_t* o = out + (size_t)blockIdx.x * stride;
    const uint8_t* src = in + (size_t)blockIdx.x * (stride / 5);
    u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
    for (int s = 0; s < steps; s++) {
        if (MODE == 0) { // thread t: 32 contiguous bytes, as two 16-byte stores
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            p[0] = v; p[1] = v;
        } else if (MODE == 1) { // each store instruction covers 1 KiB contiguous per wave
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63) * 16);
            p[0] = v; p[64] = v;
        } else if (MODE == 2) { // MODE 0 + 8 bytes of misalignment
            u32x4* p = (u32x4*)(o + 8 + (size_t)s * 8192 + threadIdx.x * 32);
            typedef u32x4 __attribute__((aligned(8))) ua;
            ((ua*)p)[0] = v; ((ua*)p)[1] = v;
        } else if (MODE == 3) { // MODE 0 + the read side: 1.6 KiB in per step
            if (threadIdx.x * 16 < in_bytes_per_step) {
                u32x4 w = *(const u32x4*)(src + (size_t)s * in_bytes_per_step + threadIdx.x * 16);
                v += w;
            }
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            p[0] = v; p[1] = v;
        }
        else if (MODE == 4) { // the load of step s is consumed at step s+1 (all waves load)
            u32x4 w = *(const u32x4*)(src + (size_t)s * 4096 + threadIdx.x * 16);
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            p[0] = v; p[1] = v;
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            v += w;
        } else if (MODE == 5) { // only wave 0 loads (and waits); all waves store; LDS hand-off + barrier
            __shared__ uint32_t lds[1024];
            u32x4 w = {0, 0, 0, 0};
            if (threadIdx.x < 64) w = *(const u32x4*)(src + (size_t)s * 1024 + threadIdx.x * 16);
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            p[0] = v; p[1] = v;
            if (threadIdx.x < 64) { *(u32x4*)&lds[threadIdx.x * 4] = w; }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v.y += lds[(threadIdx.x * 7) & 255];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (MODE == 6 || MODE == 7) { // stores + ~200 dependent VALU ops (+ 3 barriers in mode 7)
            uint32_t x = v.x;
#pragma unroll
            for (int i = 0; i < 100; i++) { x = x * 1664525u + 1013904223u; }
            v.z = x;
            if (MODE == 7) { asm volatile("s_barrier" ::: "memory"); asm volatile("s_barrier" ::: "memory"); asm volatile("s_barrier" ::: "memory"); }
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            p[0] = v; p[1] = v;
        }
        v.x += s;
    }
}

// Producer/consumer split: the last wave only loads (never stores), waves 0..3 only store.
template <int DEPTH>
__global__ __launch_bounds__(320) void kp(uint8_t* out, const uint8_t* in, int steps, size_t stride) {
    __shared__ uint32_t lds[4][1024];
    uint8_t* o = out + (size_t)blockIdx.x * stride;
    const uint8_t* src = in + (size_t)blockIdx.x * (stride / 5);
    const bool producer = threadIdx.x >= 256;
    const uint32_t l = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
    u32x4 w[DEPTH];
    if (producer)
        for (int d = 0; d < DEPTH; d++) w[d] = *(const u32x4*)(src + (size_t)d * 4096 + l * 64);
    for (int s = 0; s < steps; s++) {
        if (producer) {
            // oldest outstanding load -> LDS, then request step s + DEPTH
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
            *(u32x4*)&lds[s & 3][l * 4] = w[0];
            for (int d = 0; d + 1 < DEPTH; d++) w[d] = w[d + 1];
            w[DEPTH - 1] = *(const u32x4*)(src + (size_t)(s + DEPTH) * 4096 + l * 64);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!producer) {
            v.y += lds[s & 3][(threadIdx.x * 7) & 255];
            u32x4* p = (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            p[0] = v; p[1] = v;
        }
    }
}

// Closer model of the decoder: 8 KiB of input serve 5 steps (1.6 KiB per step).  One
// loader wave requests 8 KiB every 5 steps and waits for it 5 steps later.  PURE: the loader
// is a fifth wave that never stores; otherwise it is wave 0, which stores like the others.
template <bool PURE, int PERIOD, bool STORES, int NT = 0, bool IL = false>
__global__ __launch_bounds__(PURE ? 320 : 256) void kq(uint8_t* out, const uint8_t* in, int steps, size_t stride) {
    __shared__ uint32_t lds[2][2048];
    uint8_t* o = out + (size_t)blockIdx.x * stride;
    const uint8_t* src = in + (size_t)blockIdx.x * (stride / 5);
    const bool loader = PURE ? threadIdx.x >= 256 : threadIdx.x < 64;
    const bool storer = threadIdx.x < 256;
    const uint32_t l = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
    u32x4 w[8];
    if (loader)
        for (int d = 0; d < 8; d++) w[d] = *(const u32x4*)(src + d * 1024 + l * 16);
    for (int s = 0; s < steps; s++) {
        if (s % PERIOD == 0 && loader) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int d = 0; d < 8; d++) *(u32x4*)&lds[(s / PERIOD) & 1][d * 256 + l * 4] = w[d];
            for (int d = 0; d < 8; d++) { const u32x4* q = IL ? (const u32x4*)(in + (((size_t)(s / PERIOD + 1) * gridDim.x + blockIdx.x) % 40000) * 8192 + d * 1024 + l * 16) : (const u32x4*)(src + (size_t)(s / PERIOD + 1) * 8192 + d * 1024 + l * 16); w[d] = (NT & 2) ? __builtin_nontemporal_load(q) : *q; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (storer) {
            v.y += lds[(s / PERIOD) & 1][(threadIdx.x * 7 + s) & 2047];
            u32x4* p = IL ? (u32x4*)(out + ((size_t)s * gridDim.x + blockIdx.x) * 8192 + threadIdx.x * 32) : (u32x4*)(o + (size_t)s * 8192 + threadIdx.x * 32);
            if (STORES || v.y == 0x12345) { if (NT & 1) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, p + 1); } else { p[0] = v; p[1] = v; } }
        }
    }
}

template <bool PURE, int PERIOD, bool STORES, int NT = 0, bool IL = false>
float runq(int g, uint8_t* out, const uint8_t* in, int steps, size_t stride, hipEvent_t a, hipEvent_t b) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        kq<PURE, PERIOD, STORES, NT, IL><<<g, PURE ? 320 : 256>>>(out, in, steps, stride);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("il=%d nt=%d 8KiB/%2d steps %s %s wgs %4d: %.1f us  %.0f GB/s written-equivalent\n", (int)IL, NT, PERIOD, PURE ? "pure" : "wave0", STORES ? "stores" : "no-stores", g, best * 1e3, (double)g * steps * 8192 / best / 1e6);
    return best;
}

int main() {
    const int wgs = 2048, steps = 128;
    const size_t stride = (size_t)steps * 8192 + 256;
    uint8_t *out, *in;
    CK(hipMalloc(&out, wgs * stride + 4096));
    CK(hipMalloc(&in, wgs * stride / 5 + (1 << 20)));
    CK(hipMemset(in, 1, wgs * stride / 5 + (1 << 20)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 0; mode++) {
        for (int g : {256, 2048, 4096}) {
            if ((size_t)g > wgs && mode) continue;
            int gg = g > wgs ? wgs : g;
            float best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                CK(hipEventRecord(a));
                switch (mode) {
                case 0: k<0><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 1: k<1><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 2: k<2><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 3: k<3><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 4: k<4><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 5: k<5><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 6: k<6><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                case 7: k<7><<<gg, 256>>>(out, in, steps, stride, 1664); break;
                }
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf("mode %d wgs %4d: %.1f us  %.0f GB/s written\n", mode, gg, best * 1e3, (double)gg * steps * 8192 / best / 1e6);
        }
    }
    for (size_t st : {stride, (size_t)steps * 8192}) {
        printf("stride %zu\n", st);
        runq<false, 5, true, 0>(2048, out, in, steps, st, a, b);
        runq<false, 1000, true, 0>(2048, out, in, steps, st, a, b);
    }
    runq<false, 5, true, 0, true>(2048, out, in, steps, stride, a, b);
    runq<false, 1000, true, 0, true>(2048, out, in, steps, stride, a, b);
    for (int g : {2048}) {
        runq<false, 5, true, 0>(g, out, in, steps, stride, a, b);
        runq<false, 5, true, 1>(g, out, in, steps, stride, a, b);
        runq<false, 5, true, 2>(g, out, in, steps, stride, a, b);
        runq<false, 5, true, 3>(g, out, in, steps, stride, a, b);
    }
    return 0;
}

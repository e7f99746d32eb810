// Microbenchmark: the decoder's traffic mix (1.6 KiB in, 8 KiB out per tile) as a FLAT grid
// of short-lived workgroups, one tile each, against the persistent one-workgroup-per-page form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int IN_PER_TILE, bool ALL_WAIT = false>
__global__ __launch_bounds__(256) void flat(uint8_t* out, const uint8_t* in, int tiles_per_wg) {
    for (int k = 0; k < tiles_per_wg; k++) {
        const size_t t = (size_t)blockIdx.x * tiles_per_wg + k;
        u32x4 v = {threadIdx.x, (uint32_t)t, 3, 4};
        if (ALL_WAIT) v += *(const u32x4*)(in + t * IN_PER_TILE + (threadIdx.x * 16) % IN_PER_TILE);  // every wave waits for a load of the tile's bytes
        else if (threadIdx.x * 16 < IN_PER_TILE) v += *(const u32x4*)(in + t * IN_PER_TILE + threadIdx.x * 16);
        u32x4* p = (u32x4*)(out + t * 8192 + threadIdx.x * 32);
        p[0] = v; p[1] = v;
    }
}

int main() {
    const size_t tiles = 262144;
    uint8_t *out, *in;
    if (hipMalloc(&out, tiles * 8192) != hipSuccess || hipMalloc(&in, tiles * 4096 + 4096) != hipSuccess) return 1;
    hipMemset(in, 1, tiles * 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int tpw : {1, 2, 3, 4}) {
        for (int mode = 0; mode < 4; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(a);
                if (mode == 0) flat<0><<<tiles / tpw, 256>>>(out, in, tpw);
                if (mode == 1) flat<1664><<<tiles / tpw, 256>>>(out, in, tpw);
                if (mode == 2) flat<4096><<<tiles / tpw, 256>>>(out, in, tpw);
                if (mode == 3) flat<1664, true><<<tiles / tpw, 256>>>(out, in, tpw);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep && ms < best) best = ms;
            }
            const int inb = mode == 0 ? 0 : mode == 2 ? 4096 : 1664;
            printf("tiles/wg %3d %s in/tile %4d: %.1f us  written %.0f GB/s  total %.0f GB/s\n", tpw, mode == 3 ? "all waves wait" : "             ", inb, best * 1e3,
                   tiles * 8192.0 / best / 1e6, tiles * (8192.0 + inb) / best / 1e6);
        }
    }
    return 0;
}

// Does LDS take 4-byte accesses at addresses that are not multiples of four on this chip?
// hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) buf[i] = 0;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)buf + lane * 5 + 1; // odd and even offsets
    const uint32_t val = 0x04030201u + lane * 0x01010101u;
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(val) : "memory");
    __syncthreads();
    uint32_t got = 0;
    for (uint32_t e = 0; e < 4; e++) got |= (uint32_t)buf[lane * 5 + 1 + e] << (8 * e);
    uint32_t rd;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rd) : "v"(addr) : "memory");
    out[lane] = got;
    out[64 + lane] = rd;
    out[128 + lane] = buf[lane * 5]; // the byte in front stays 0
}
int main() {
    uint32_t* d; hipMalloc(&d, 192 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok_w = 0, ok_r = 0, clean = 0;
    for (int l = 0; l < 64; l++) { const uint32_t v = 0x04030201u + l * 0x01010101u; ok_w += h[l] == v; ok_r += h[64 + l] == v; clean += h[128 + l] == 0; }
    printf("unaligned ds_write_b32 right in %d of 64 lanes, ds_read_b32 in %d of 64, byte in front untouched in %d (lane 1: wrote %08x read %08x)\n", ok_w, ok_r, clean, h[1], h[65]);
    return 0;
}

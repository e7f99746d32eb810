// Microbenchmark: what the memory system allows for the shape of an RLE_DICTIONARY int64 page
// batch with a dictionary that LDS cannot hold -- 16 Mi uniform 16-bit indexes (already whole
// uint16 words: nothing to unpack), an 8-byte look-up in a 64 Ki-entry (512 KB) table that stays
// in L2, and 128 MB of 16-byte stores.  No page plumbing, no run tables, no bounds checks.
//
//   store      the 128 MB of stores alone
//   flat       a workgroup per 2048 values: 16 bytes of indexes per thread, eight gathers, four stores
//   persist    1024-thread workgroups, one per CU, each wavefront walking tiles with its next indexes in flight
//   part       the same with the first LDS_KB of the table in LDS and only the rest gathered from memory
//   lds        every index folded into the LDS part (the time if the whole table fitted)
//   sorted-T   `persist` over indexes sorted inside stretches of T values: what any scheme that
//              reorders look-ups inside a stretch of T could at best save in cache-line requests
//   after-stores / cold   `part` and `persist` again (a) right behind a kernel that writes 2 GB elsewhere, as the dictionary
//              batch runs behind the DELTA batch in bench.py, and (b) behind that kernel and a 1 ms idle
//              spin: caches cold, nothing draining.  Launched over and over by itself the whole 168 MB
//              working set sits in the 256 MB last-level cache; (a) and (b) tell a cold cache from a busy memory
//
// hipcc --offload-arch=gfx950 -O3 -o gather_floor gather_floor.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t TILE = 2048;

__global__ __launch_bounds__(256) void store_kernel(uint64_t* out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 2048 + threadIdx.x * 2;
    if (i + 2048 > n + threadIdx.x * 2) return;
    u64x2 v = {i, i + 1};
#pragma unroll
    for (int k = 0; k < 4; k++) *(u64x2*)(out + i + k * 512) = v;
}

__global__ __launch_bounds__(256) void flat_kernel(uint64_t* out, const uint16_t* idx, const uint64_t* table, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 2048 + threadIdx.x * 2;
    if (i + 2048 > n + threadIdx.x * 2) return;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = *(const uint32_t*)(idx + i + k * 512);
    uint64_t x[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        x[2 * k] = table[w[k] & 0xffff];
        x[2 * k + 1] = table[w[k] >> 16];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { u64x2 v = {x[2 * k], x[2 * k + 1]}; *(u64x2*)(out + i + k * 512) = v; }
}

// Persistent: a tile is 16 steps of 128 values; lane l owns values 2l, 2l + 1 of every step, so a store
// instruction of the wavefront writes 1 KB of whole cache lines.  The indexes of the next tile are in flight
// while this one is looked up.  PIPE: the gathers of the NEXT half tile are issued before the stores of this
// one, so that (memory operations of a wavefront retire in order) no gather is waited for behind a store
// issued just before it.
template <bool PART, bool FOLD, bool PIPE>
__global__ __launch_bounds__(1024) void persist_kernel(uint64_t* out, const uint16_t* idx, const uint64_t* table, uint64_t n_tiles,
                                                       uint32_t lds_n, uint32_t table_n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t shm[];
    uint64_t* lds = reinterpret_cast<uint64_t*>(shm);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (PART || FOLD) {
        for (uint32_t k = tid * 2; k < lds_n; k += 2048) *(u64x2*)(lds + k) = *(const u64x2*)(table + k);
        __syncthreads();
    }
    const uint64_t T0 = n_tiles * blockIdx.x / gridDim.x, T1 = n_tiles * (blockIdx.x + 1) / gridDim.x;
    uint64_t t = T0 + wave;
    if (t >= T1) return;
    uint32_t w[16], wn[16];
    auto fetch = [&](uint64_t tile, uint32_t (&d)[16]) {
#pragma unroll
        for (int s = 0; s < 16; s++) d[s] = *(const uint32_t*)(idx + tile * TILE + s * 128 + lane * 2);
    };
    auto gather = [&](const uint32_t (&d)[16], int half, uint64_t (&g)[16]) { // the memory side of half a tile's look-ups
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t word = d[half * 8 + (k >> 1)];
            const uint32_t i = (k & 1) ? word >> 16 : word & 0xffff;
            if (FOLD) g[k] = 0;
            else if (PART) g[k] = table[i < lds_n ? 0 : i];
            else g[k] = table[i];
        }
    };
    auto finish = [&](uint64_t tile, const uint32_t (&d)[16], int half, const uint64_t (&g)[16]) { // the LDS side, and the stores
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            uint64_t x[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t word = d[half * 8 + (k >> 1)];
                uint32_t i = e ? word >> 16 : word & 0xffff;
                if (FOLD) { i = i % lds_n; x[e] = lds[i]; }
                else if (PART) { const bool in = i < lds_n; const uint64_t a = lds[in ? i : 0]; x[e] = in ? a : g[k + e]; }
                else x[e] = g[k + e];
            }
            u64x2 v = {x[0], x[1]};
            *(u64x2*)(out + tile * TILE + (half * 8 + (k >> 1)) * 128 + lane * 2) = v;
        }
    };
    fetch(t, w);
    uint64_t g0[16], g1[16];
    if (PIPE) gather(w, 0, g0);
    for (;;) {
        const uint64_t tn = t + 16;
        const bool more = tn < T1;
        if (more) fetch(tn, wn);
        if (PIPE) {
            gather(w, 1, g1);
            finish(t, w, 0, g0);
            if (more) gather(wn, 0, g0);
            finish(t, w, 1, g1);
        } else {
            gather(w, 0, g0);
            finish(t, w, 0, g0);
            gather(w, 1, g1);
            finish(t, w, 1, g1);
        }
        if (!more) break;
#pragma unroll
        for (int s = 0; s < 16; s++) w[s] = wn[s];
        t = tn;
    }
}

__global__ __launch_bounds__(256) void flush_kernel(u64x2* big, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { u64x2 v = {i, i}; big[i] = v; }
}
__global__ void spin_kernel(uint64_t ticks) { // s_memrealtime counts at 100 MHz
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

static float time_it(int reps, void (*launch)(void*), void* arg, float* median) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    std::vector<float> ms(reps);
    for (int r = 0; r < reps + 2; r++) {
        CHECK(hipEventRecord(a));
        launch(arg);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float m; CHECK(hipEventElapsedTime(&m, a, b));
        if (r >= 2) ms[r - 2] = m;
    }
    std::sort(ms.begin(), ms.end());
    *median = ms[reps / 2];
    return ms[0];
}

struct Args { uint64_t* out; const uint16_t* idx; const uint64_t* table; uint64_t n; uint32_t lds_n; int cus; };

int main(int argc, char** argv) {
    const uint64_t n = 16ull << 20;
    const uint32_t table_n = 65536;
    const uint32_t lds_kb = argc > 1 ? atoi(argv[1]) : 152;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<uint16_t> h(n);
    uint64_t s = 0xC1;
    for (uint64_t i = 0; i < n; i++) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        h[i] = (uint16_t)z;
    }
    std::vector<uint64_t> ht(table_n);
    for (uint32_t i = 0; i < table_n; i++) ht[i] = 0x0123456789abcdefull * (i + 1);
    uint16_t* idx; uint64_t *table, *out;
    CHECK(hipMalloc(&idx, n * 2 + 64)); CHECK(hipMalloc(&table, table_n * 8)); CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMemcpy(table, ht.data(), table_n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(idx, h.data(), n * 2, hipMemcpyHostToDevice));
    Args A{out, idx, table, n, lds_kb * 1024 / 8, cus};
    const double alg = n * 10.0 + table_n * 8.0;
    auto report = [&](const char* name, float best, float med) {
        printf("%-72s best %7.1f us  median %7.1f us   %6.0f GB/s algorithmic, frac of 8 TB/s %.3f\n", name, best * 1e3, med * 1e3, alg / med / 1e6, alg / med / 1e6 / 8000.0);
    };
    float med, best;
    best = time_it(20, [](void* p) { Args& a = *(Args*)p; store_kernel<<<a.n / 2048, 256>>>(a.out, a.n); }, &A, &med);
    printf("%-72s best %7.1f us  median %7.1f us   (%.0f GB/s written)\n", "store: 128 MB of 16-byte stores", best * 1e3, med * 1e3, n * 8.0 / med / 1e6);
    auto check = [&](const char* what, const std::vector<uint16_t>& hi) {
        std::vector<uint64_t> ho(4096);
        CHECK(hipMemcpy(ho.data(), out + (n - 4096), 4096 * 8, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < 4096; k++)
            if (ho[k] != ht[hi[n - 4096 + k]]) { fprintf(stderr, "%s: wrong value at %u\n", what, k); exit(2); }
    };
    best = time_it(20, [](void* p) { Args& a = *(Args*)p; flat_kernel<<<a.n / 2048, 256>>>(a.out, a.idx, a.table, a.n); }, &A, &med);
    check("flat", h);
    report("flat: workgroup per 2048 values", best, med);
    typedef void (*Kern)(uint64_t*, const uint16_t*, const uint64_t*, uint64_t, uint32_t, uint32_t);
    struct Variant { const char* name; Kern k; bool lds; };
    const Variant all_mem[2] = {{"persist", persist_kernel<false, false, false>, false}, {"persist, gathers ahead", persist_kernel<false, false, true>, false}};
    const Variant part[2] = {{"part", persist_kernel<true, false, false>, true}, {"part, gathers ahead", persist_kernel<true, false, true>, true}};
    const Variant fold = {"lds: every look-up in LDS", persist_kernel<false, true, false>, true};
    for (Kern k : {part[0].k, part[1].k, fold.k}) CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t ea, eb;
    CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
    u64x2* big;
    const uint64_t big_n16 = (2048ull << 20) / 16;
    CHECK(hipMalloc(&big, big_n16 * 16));
    // mode 0: launched over and over; 1: right behind 2 GB of stores to another buffer; 2: behind those and an idle millisecond
    auto run = [&](const Variant& v, uint32_t kb, int wgs_per_cu, int mode, const std::vector<uint16_t>& hi, const char* note) {
        const uint32_t lds_n = v.lds ? kb * 1024 / 8 : 0;
        std::vector<float> ms;
        for (int r = 0; r < 14; r++) {
            if (mode > 0) flush_kernel<<<2048, 256>>>(big, big_n16);
            if (mode == 2) spin_kernel<<<1, 64>>>(100000);
            CHECK(hipEventRecord(ea));
            v.k<<<cus * wgs_per_cu, 1024, lds_n * 8>>>(out, idx, table, n / TILE, lds_n, table_n);
            CHECK(hipEventRecord(eb));
            CHECK(hipEventSynchronize(eb));
            float m; CHECK(hipEventElapsedTime(&m, ea, eb));
            if (r >= 2) ms.push_back(m);
        }
        CHECK(hipGetLastError());
        std::sort(ms.begin(), ms.end());
        if (v.k != fold.k) check(v.name, hi);
        char name[128];
        if (v.lds && v.k != fold.k) snprintf(name, sizeof name, "%s (%u KB in LDS)%s", v.name, kb, note);
        else snprintf(name, sizeof name, "%s%s", v.name, note);
        report(name, ms[0], ms[ms.size() / 2]);
    };
    for (int p2 = 0; p2 < 2; p2++) run(all_mem[p2], 0, 1, 0, h, "");
    run(all_mem[1], 0, 2, 0, h, ", two workgroups per CU");
    for (uint32_t kb : {64u, 128u, lds_kb}) for (int p2 = 0; p2 < 2; p2++) run(part[p2], kb, 1, 0, h, "");
    run(fold, lds_kb, 1, 0, h, "");
    for (int mode = 1; mode < 3; mode++)
        for (const Variant& v : {all_mem[0], all_mem[1], part[0], part[1], fold})
            run(v, lds_kb, 1, mode, h, mode == 1 ? ", right behind 2 GB of stores" : ", behind 2 GB of stores + 1 ms idle");
    // indexes sorted inside stretches of T values
    for (uint32_t T : {2048u, 16384u, 65536u, 262144u}) {
        std::vector<uint16_t> hs(h);
        for (uint64_t i = 0; i < n; i += T) std::sort(hs.begin() + i, hs.begin() + i + T);
        CHECK(hipMemcpy(idx, hs.data(), n * 2, hipMemcpyHostToDevice));
        char note[64];
        snprintf(note, sizeof note, ", indexes sorted per %u", T);
        run(all_mem[1], 0, 1, 0, hs, note);
        run(part[1], lds_kb, 1, 0, hs, note);
    }
    return 0;
}

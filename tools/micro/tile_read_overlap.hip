// tools/micro/tile_read_overlap.hip -- would k_inv_cols_r gain from loads that travel while the block computes?  A MODEL of
// that kernel: a block reads one tile of Q (601 rows x 128 bytes, every thread's 12 - 13 loads in flight together), puts it
// in LDS, and then "computes" for about as long as the kernel's three stages and its scan take (a loop of fused
// multiply-adds on registers with LDS reads and writes in between; `work` = iterations, calibrated below so that the
// one-shot form takes about the kernel's 0.30 ms).  Same launch order (tile-major), same occupancy (76.8 KB of LDS, 512
// threads: two blocks per CU).  Forms:
//   O   one-shot blocks, as the kernel is; with and without its stagger (odd blocks of the first generation start late)
//   P   persistent blocks (one per slot) with the NEXT tile's loads in registers during the compute phase (48 more VGPRs)
//   D   persistent blocks, the next tile's rows sent straight into LDS (global_load_lds_dwordx4, no registers) as soon as the
//       compute phase has read the tile into registers -- the part of the real kernel that could overlap is its last stage
//       and scan, ~25 % of a block's life: `tail` = the share of the compute loop that runs behind the DMA issue
//   E   ONE persistent block per CU with TWO tiles of LDS (153.6 KB): the next tile's DMA runs behind ALL of this tile's compute, at half the waves
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/tile_read_overlap tools/micro/tile_read_overlap.hip && /tmp/tile_read_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ float4 lds[];
constexpr int M1 = 600, M2 = 2400, NT = 512, RL = 6, MB = M1 / RL, ITEMS = (MB / 2) * 8, NTILES = M2 / 16;
constexpr size_t PITCH4 = (size_t)M2 / 2;
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float4 *p)
{
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
// the kernel's work item e -> the 2 RL rows it loads (butterflies u_b and MB - u_b)
__device__ __forceinline__ void ask(const float4 *q, size_t base, int e, float4 (&a)[RL], float4 (&b)[RL])
{
    const int g = e & 7, v = e >> 3, ub = v, ubp = v == 0 ? MB / 2 : MB - v;
#pragma unroll
    for (int t = 0; t < RL; t++) {
        a[t] = ldnt(q + base + (size_t)(ub + t * MB) * PITCH4 + g);
        b[t] = ldnt(q + base + (size_t)(ubp + t * MB) * PITCH4 + g);
    }
}
__device__ __forceinline__ void put(int e, const float4 (&a)[RL], const float4 (&b)[RL])
{
    const int g = e & 7, v = e >> 3, ub = v, ubp = v == 0 ? MB / 2 : MB - v;
#pragma unroll
    for (int t = 0; t < RL; t++) {
        lds[(ub + t * MB) * 8 + g] = a[t];
        lds[(ubp + t * MB) * 8 + g] = b[t];
    }
}
// `work` rounds of: read 10 slots, 40 x 4 dependent-free fused multiply-adds, write 10 slots (rounds 0 .. work-1; a barrier every `per` rounds)
__device__ __forceinline__ float compute(int tid, int first, int last, float acc, float4 *L = lds)
{
    float4 r[10];
    for (int it = first; it < last; it++) {
#pragma unroll
        for (int t = 0; t < 10; t++) r[t] = L[(tid + t * 480) % 4800];
#pragma unroll
        for (int k = 0; k < 12; k++)
#pragma unroll
            for (int t = 0; t < 10; t++) {
                r[t].x = fmaf(r[t].x, 1.0001f, r[t].y); r[t].y = fmaf(r[t].y, 0.9999f, r[t].z);
                r[t].z = fmaf(r[t].z, 1.0002f, r[t].w); r[t].w = fmaf(r[t].w, 0.9998f, r[t].x);
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 10; t++) { L[(tid + t * 480) % 4800] = r[t]; acc += r[t].x; }
        __syncthreads();
    }
    return acc;
}

template <int FORM>
__global__ __launch_bounds__(NT, 4) void k_model(const float4 *__restrict__ q, float *out, int npairs, int work, int tail, int stagger, unsigned first_gen)
{
    const int tid = threadIdx.x;
    const unsigned total = (unsigned)npairs * NTILES;
    float acc = 0.f;
    if (FORM == 0) {
        const unsigned lin = blockIdx.x;             // tile-major: pair fastest
        if (stagger && total > first_gen && lin < first_gen && (lin & 1u))
            for (int i = 0; i < 600 * 27 / 64 / 2; i += 16) __builtin_amdgcn_s_sleep(16);
        const unsigned pair = lin % npairs, tile = lin / npairs;
        const size_t base = (size_t)pair * (M1 + 1) * PITCH4 + (size_t)tile * 8;
        float4 a[RL], b[RL];
        if (tid < ITEMS) { ask(q, base, tid, a, b); put(tid, a, b); }
        __syncthreads();
        acc = compute(tid, 0, work, acc);
    } else if (FORM == 1) {
        float4 a[RL], b[RL];
        unsigned lin = blockIdx.x;
        if (lin >= total) return;
        { const unsigned pair = lin % npairs, tile = lin / npairs; if (tid < ITEMS) ask(q, (size_t)pair * (M1 + 1) * PITCH4 + (size_t)tile * 8, tid, a, b); }
        for (;;) {
            if (tid < ITEMS) put(tid, a, b);
            const unsigned nxt = lin + gridDim.x;
            if (nxt < total) { const unsigned pair = nxt % npairs, tile = nxt / npairs; if (tid < ITEMS) ask(q, (size_t)pair * (M1 + 1) * PITCH4 + (size_t)tile * 8, tid, a, b); }
            __syncthreads();
            acc = compute(tid, 0, work, acc);
            if (nxt >= total) break;
            lin = nxt;
        }
    } else if (FORM == 3) {
        // E: ONE block per CU with TWO tiles of LDS: the next tile travels into the other buffer during ALL of this tile's compute
        unsigned lin = blockIdx.x;
        if (lin >= total) return;
        const int lane = tid & 63, wave = tid >> 6;
        auto dma = [&](unsigned l, float4 *dst) {
            const unsigned pair = l % npairs, tile = l / npairs;
            const float4 *src = q + (size_t)pair * (M1 + 1) * PITCH4 + (size_t)tile * 8;
            for (int r0 = 8 * wave; r0 < M1; r0 += 8 * (NT / 64)) {
                const float4 *g = src + (size_t)(r0 + (lane >> 3)) * PITCH4 + (lane & 7);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(dst + r0 * 8), 16, 0, 0);
            }
        };
        int cur = 0;
        dma(lin, lds);
        for (;;) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned nxt = lin + gridDim.x;
            if (nxt < total) dma(nxt, lds + (cur ^ 1) * 4800);
            acc = compute(tid, 0, work, acc, lds + cur * 4800);
            if (nxt >= total) break;
            lin = nxt; cur ^= 1;
        }
    } else {
        // D: rows by LDS-DMA.  Wave w sends rows 8 (w + 8 i) .. + 7 (eight lanes per 128-byte row piece): 1 KiB per instruction, contiguous in LDS
        unsigned lin = blockIdx.x;
        if (lin >= total) return;
        const int lane = tid & 63, wave = tid >> 6;
        auto dma = [&](unsigned l) {
            const unsigned pair = l % npairs, tile = l / npairs;
            const float4 *src = q + (size_t)pair * (M1 + 1) * PITCH4 + (size_t)tile * 8;
            for (int r0 = 8 * wave; r0 < M1; r0 += 64) { // rows r0 .. r0 + 7 (M1 = 600 = 75 x 8: whole instructions)
                const float4 *g = src + (size_t)(r0 + (lane >> 3)) * PITCH4 + (lane & 7);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(lds + r0 * 8), 16, 0, 0);
            }
        };
        dma(lin);
        for (;;) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc = compute(tid, 0, work - tail, acc);   // ends with a barrier: every thread holds what it needs in registers
            const unsigned nxt = lin + gridDim.x;
            float4 keep[10];
#pragma unroll
            for (int t = 0; t < 10; t++) keep[t] = lds[(tid + t * 480) % 4800];
            __syncthreads();
            if (nxt < total) dma(nxt);                 // the tile is free: the next one travels during the tail
            for (int it = 0; it < tail; it++)
#pragma unroll
                for (int k = 0; k < 12; k++)
#pragma unroll
                    for (int t = 0; t < 10; t++) {
                        keep[t].x = fmaf(keep[t].x, 1.0001f, keep[t].y); keep[t].y = fmaf(keep[t].y, 0.9999f, keep[t].z);
                        keep[t].z = fmaf(keep[t].z, 1.0002f, keep[t].w); keep[t].w = fmaf(keep[t].w, 0.9998f, keep[t].x);
                    }
#pragma unroll
            for (int t = 0; t < 10; t++) acc += keep[t].x;
            if (nxt >= total) break;
            lin = nxt;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int pairs = 124;
    const size_t n4 = (size_t)pairs * (M1 + 1) * PITCH4;
    float4 *q; float *o;
    (void)hipMalloc(&q, n4 * 16); (void)hipMalloc(&o, 64); (void)hipMemset(q, 0, n4 * 16);
    const double bytes = (double)n4 * 16;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const void *fns[4] = { (const void *)k_model<0>, (const void *)k_model<1>, (const void *)k_model<2>, (const void *)k_model<3> };
    for (const void *f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto run = [&](int form, int work, int tail, int stagger) {
        int per_cu = 0;
        const size_t ldsb = form == 3 ? 2 * 76800 : 76800;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fns[form], NT, ldsb);
        const unsigned slots = (unsigned)per_cu * 256u, total = (unsigned)pairs * NTILES;
        float best = 1e9f;
        for (int r = 0; r < 6; r++) {
            float ms = 0;
            (void)hipEventRecord(a);
            if (form == 0) hipLaunchKernelGGL(k_model<0>, dim3(total), dim3(NT), ldsb, 0, q, o, pairs, work, tail, stagger, slots);
            if (form == 1) hipLaunchKernelGGL(k_model<1>, dim3(slots), dim3(NT), ldsb, 0, q, o, pairs, work, tail, stagger, slots);
            if (form == 2) hipLaunchKernelGGL(k_model<2>, dim3(slots), dim3(NT), ldsb, 0, q, o, pairs, work, tail, stagger, slots);
            if (form == 3) hipLaunchKernelGGL(k_model<3>, dim3(slots), dim3(NT), ldsb, 0, q, o, pairs, work, tail, stagger, slots);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
            if (r >= 1 && ms < best) best = ms;
        }
        return std::pair<float, int>(best, per_cu);
    };
    printf("model of k_inv_cols_r: %.3f GB of tiles per launch; ms (TB/s) by compute rounds per tile\n", bytes / 1e9);
    printf("%-58s", "rounds:");
    const int works[] = { 0, 1, 2, 3, 4, 6 };
    for (int w : works) printf("  %6d      ", w);
    printf("\n");
    struct Row { const char *name; int form, stagger, tail_of_4; } rows[] = {
        { "O one-shot", 0, 0, 0 }, { "O one-shot, staggered first generation", 0, 1, 0 },
        { "P persistent, next tile in registers", 1, 0, 0 },
        { "D persistent, LDS-DMA behind the last round (tail 1)", 2, 0, 1 }, { "D persistent, LDS-DMA, tail 2 rounds", 2, 0, 2 },
        { "E one block per CU, two tiles of LDS, DMA behind everything", 3, 0, 0 },
    };
    for (int rep = 0; rep < 2; rep++)
        for (const Row &R : rows) {
            printf("%-58s", R.name);
            for (int w : works) {
                const int tail = R.form == 2 ? (w < R.tail_of_4 ? w : R.tail_of_4) : 0;
                const auto res = run(R.form, w, tail, R.stagger);
                printf("  %.3f (%4.2f)", res.first, bytes / res.first / 1e9);
            }
            printf("\n");
        }
    return 0;
}

// tools/micro/pipe_handoff.hip -- can two kernels that run SIDE BY SIDE on disjoint sets of CUs (CU-masked
// streams) hand data from one to the other inside their lifetimes, through buffers small enough to stay in
// the Infinity Cache?  (The question behind a spatial pipeline fwd_cols | rows | inv_cols: DESIGN.md 5.)
//   1. which CUs does a CU-masked stream use (HW_ID / XCC_ID per block)?
//   2. producer: writes slices of ring slot p % RING with write-through (sc1) 16-byte stores, drains them
//      (s_waitcnt vmcnt(0) per wave, block barrier), then one lane adds to done[p] (agent scope);
//      before reusing a ring slot it polls consumed[p - RING].
//      consumer: polls done[p] with an sc1 load, block barrier, reads the slot TRANSPOSED (every consumer block
//      reads a piece of every producer block's slice) with sc1 16-byte loads, checks every word, adds to consumed[p].
//   3. time against the same two kernels run one after the other over the whole batch (kernel boundary).
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pipe_handoff pipe_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef NB
#define NB 256                       // producer blocks per item
#endif
constexpr int SLICE = 64 * 1024;     // bytes per producer block -> NB * 64 KiB per item (16 MiB: about one pair's Z)
#ifndef NBC
#define NBC 512                      // consumer blocks per item
#endif
constexpr int CPAD = 64;             // counters 256 bytes apart: adders and pollers of different items on different lines
constexpr int SPIN_MAX = 1 << 15;  // x ~0.6 us of s_sleep: a stuck wait gives up after ~80 ms and is counted

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float pattern(int item, unsigned idx) { return (float)((item * 7919u + idx * 2654435761u) >> 8 & 0xFFFFF); }

__global__ __launch_bounds__(256) void k_whoami(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    // stay a while so that blocks spread over the allowed CUs
    for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(10);
}

// mode bits: 1 polling / signalling (hand-offs inside the kernels' lifetimes), 2 sc1 stores, 4 sc1 loads; 0 = plain, kernel boundary orders them
__global__ __launch_bounds__(256) void k_producer(float *ring, int nring, int nitems, unsigned *done, const unsigned *consumed, unsigned *err, int mode)
{
    const int pipelined = mode & 1;
    const int item = blockIdx.x / NB, b = blockIdx.x % NB;
    if (pipelined && item >= nring) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (ld_sc1(&consumed[(item - nring) * CPAD]) < (unsigned)NBC) {
                __builtin_amdgcn_s_sleep(127);
                if (++spins > SPIN_MAX) { atomicAdd(&err[1], 1u); break; }
            }
        }
        __syncthreads();
    }
    float *slot = ring + (size_t)(item % nring) * (NB * SLICE / 4) + (size_t)b * (SLICE / 4);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slot, 0, SLICE, 0x00020000);
    for (unsigned i = threadIdx.x; i < SLICE / 16; i += 256) {
        const unsigned e = (unsigned)b * (SLICE / 4) + i * 4;
        f4 v = { pattern(item, e), pattern(item, e + 1), pattern(item, e + 2), pattern(item, e + 3) };
        if (mode & 2) __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 16, 0, 16 /* sc1 */);
        else __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 16, 0, 0);
    }
    if (pipelined) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&done[item * CPAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_consumer(const float *ring, int nring, int nitems, const unsigned *done, unsigned *consumed, unsigned *err, int mode)
{
    const int pipelined = mode & 1;
    const int item = blockIdx.x / NBC, c = blockIdx.x % NBC;
    if (pipelined) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (ld_sc1(&done[item * CPAD]) < (unsigned)NB) {
                __builtin_amdgcn_s_sleep(127);
                if (++spins > SPIN_MAX) { atomicAdd(&err[1], 1u); break; }
            }
        }
        __syncthreads();
    }
    const float *slot = ring + (size_t)(item % nring) * (NB * SLICE / 4);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)slot, 0, NB * SLICE, 0x00020000);
    // transposed read: from every producer slice the c-th piece of SLICE / NBC bytes
    unsigned bad = 0;
    constexpr unsigned PIECE = SLICE / NBC; // bytes
    for (unsigned i = threadIdx.x; i < NB * PIECE / 16; i += 256) {
        const unsigned pb = i / (PIECE / 16), k = i % (PIECE / 16);
        const unsigned off = pb * SLICE + (unsigned)c * PIECE + k * 16;
        f4 v = (mode & 4) ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16 /* sc1 */) : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        const unsigned e = off / 4;
        bad += (v.x != pattern(item, e)) + (v.y != pattern(item, e + 1)) + (v.z != pattern(item, e + 2)) + (v.w != pattern(item, e + 3));
    }
    if (bad) atomicAdd(&err[0], bad);
    if (pipelined) {
        __syncthreads(); // every wave's loads have returned (they were consumed above)
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&consumed[item * CPAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv)
{
    const int nitems = argc > 1 ? atoi(argv[1]) : 256, nring = argc > 2 ? atoi(argv[2]) : 4;
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", ncu);
    const int words = (ncu + 31) / 32;
    // masks: A = even CU bits, B = odd CU bits; C = first half, D = second half
    std::vector<uint32_t> mA(words, 0x55555555u), mB(words, 0xAAAAAAAAu), mC(words, 0), mD(words, 0);
    for (int i = 0; i < ncu; i++) (i < ncu / 2 ? mC : mD)[i / 32] |= 1u << (i % 32);
    hipStream_t sA, sB, sC, sD;
    CK(hipExtStreamCreateWithCUMask(&sA, words, mA.data()));
    CK(hipExtStreamCreateWithCUMask(&sB, words, mB.data()));
    CK(hipExtStreamCreateWithCUMask(&sC, words, mC.data()));
    CK(hipExtStreamCreateWithCUMask(&sD, words, mD.data()));
    unsigned *who;
    CK(hipMalloc(&who, 4096 * 2 * 4));
    auto census = [&](hipStream_t s, const char *name) {
        CK(hipMemset(who, 0, 4096 * 2 * 4));
        hipLaunchKernelGGL(k_whoami, dim3(2048), dim3(256), 0, s, who);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> h(4096);
        CK(hipMemcpy(h.data(), who, 4096 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus;
        int per_xcc[8] = { 0 };
        for (int i = 0; i < 2048; i++) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
            const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7; // gfx9 HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
            const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
            if (cus.insert(key).second) per_xcc[xcc & 7]++;
        }
        printf("%-10s distinct CUs %zu, per XCC:", name, cus.size());
        for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
        printf("\n");
    };
    census(nullptr, "unmasked");
    census(sA, "even bits");
    census(sB, "odd bits");
    census(sC, "low half");
    census(sD, "high half");

    float *ring;
    unsigned *done, *consumed, *err;
    CK(hipMalloc(&ring, (size_t)nitems * NB * SLICE)); // big enough for the unpipelined run (no reuse)
    CK(hipMalloc(&done, nitems * 4 * CPAD));
    CK(hipMalloc(&consumed, nitems * 4 * CPAD));
    CK(hipMalloc(&err, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char *name, float ms) {
        unsigned h[2];
        CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
        const double gb = (double)nitems * NB * SLICE / 1e9;
        printf("%-46s %8.3f ms  %6.2f TB/s written+read  bad words %u  spin timeouts %u\n", name, ms, 2 * gb / ms, h[0], h[1]);
    };
    auto sequential = [&](int mode, const char *name) {
        CK(hipMemset(err, 0, 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(k_producer, dim3(nitems * NB), dim3(256), 0, nullptr, ring, nitems, nitems, done, consumed, err, mode);
        hipLaunchKernelGGL(k_consumer, dim3(nitems * NBC), dim3(256), 0, nullptr, ring, nitems, nitems, done, consumed, err, mode);
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        report(name, ms);
    };
    auto side_by_side = [&](int mode, int ring_items, const char *name) {
        CK(hipMemset(err, 0, 8)); CK(hipMemset(done, 0, nitems * 4 * CPAD)); CK(hipMemset(consumed, 0, nitems * 4 * CPAD));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sC));
        CK(hipStreamWaitEvent(sD, e0, 0));
        hipLaunchKernelGGL(k_producer, dim3(nitems * NB), dim3(256), 0, sC, ring, ring_items, nitems, done, consumed, err, mode);
        hipLaunchKernelGGL(k_consumer, dim3(nitems * NBC), dim3(256), 0, sD, ring, ring_items, nitems, done, consumed, err, mode);
        CK(hipEventRecord(e1, sD));
        CK(hipStreamWaitEvent(sC, e1, 0));
        CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(sC));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        report(name, ms);
    };
    for (int rep = 0; rep < 2; rep++) {
        sequential(0, "sequential, plain");
        sequential(2, "sequential, sc1 stores");
        sequential(4, "sequential, sc1 loads");
        sequential(6, "sequential, sc1 stores + loads");
        side_by_side(0, nitems, "side by side halves, no sync (WRONG), plain");
        side_by_side(6, nitems, "side by side halves, no sync (WRONG), sc1");
        side_by_side(7, nitems, "pipelined halves, sc1, no slot reuse");
        side_by_side(7, nring, "pipelined halves, sc1, ring");
        side_by_side(1, nring, "pipelined halves, PLAIN (stale?), ring");
    }
    return 0;
}

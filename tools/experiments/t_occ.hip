// tools/experiments/t_occ.hip -- occupancy experiment: LDS-resident forward + inverse
// 1200-point transforms in a loop, no HBM traffic inside the timed region; two-member work items
// (256 threads, 16 waves per CU) against single-member work items (512 threads, up to 32 waves per CU).
// Needs the templated stage code of kargs_cx1.patch:
//   git worktree add /tmp/wt 77fbf5d && cd /tmp/wt && git apply <repo>/tools/experiments/kargs_cx1.patch
//   cp <repo>/tools/experiments/t_occ.hip old-audiosync_amd/csrc/ && cd old-audiosync_amd
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-finite-math-only -fno-slp-vectorize -w -I../include -Icsrc -o t_occ csrc/t_occ.hip
// Measured (MI355X): 1.79 / 2.08 / 2.02 / 1.87 / 1.80 ms for the five configurations below: no gain from occupancy.
#include "asx_internal.h"
#include "lds_fft.h"
#include <cmath>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) float2 smem[];
template <int MAXR, int THREADS, int WAVES, typename T>
__global__ __launch_bounds__(THREADS, WAVES) void kt(AsxStagePlan sp, const float2 *tw, LdsLayout L, float2 *out, int loops)
{
    using S = typename LdsSlot<T>::type;
    S *lds = reinterpret_cast<S *>(smem);
    const int total = 4 * sp.n;
    float2 *o = out + (size_t)blockIdx.x * total;
    for (int e = threadIdx.x; e < total; e += blockDim.x) smem[e] = o[e];
    __syncthreads();
    for (int it = 0; it < loops; it++) {
        const TwPre pre = tw_prefetch<false>(sp, 0, L, tw);
        lds_fft<MAXR, false, false, T>(lds, sp, L, tw, pre);
        const TwPre pre2 = tw_prefetch<false>(sp, sp.nstages - 1, L, tw);
        lds_fft<MAXR, true, false, T>(lds, sp, L, tw, pre2);
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const float2 v = smem[e];
            smem[e] = make_float2(v.x * (1.f / 1200.f), v.y * (1.f / 1200.f));
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < total; e += blockDim.x) o[e] = smem[e];
}
template <typename K> static float timeit(K launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main()
{
    const int n = 1200, blocks = 256 * 4 * 6, loops = 16;
    AsxStagePlan sp; sp.n = n; sp.nstages = 3; sp.radices = 10ull | (10ull << 8) | (12ull << 16);
    std::vector<float2> tw(n);
    for (int i = 0; i < n; i++) tw[i] = make_float2((float)cos(2 * M_PI * i / n), (float)-sin(2 * M_PI * i / n));
    float2 *dtw, *d;
    (void)hipMalloc(&dtw, n * sizeof(float2));
    (void)hipMemcpy(dtw, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    (void)hipMalloc(&d, (size_t)blocks * 4 * n * sizeof(float2));
    std::vector<float2> h((size_t)blocks * 4 * n);
    for (size_t i = 0; i < h.size(); i++)
        h[i] = make_float2((float)((i * 2654435761u) % 1000) / 1000.f - 0.5f, (float)((i * 40503u) % 1000) / 1000.f - 0.5f);
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    const size_t lds = 4 * n * sizeof(float2);
    LdsLayout L2; L2.ngroups = 2; L2.log_ngroups = 1; L2.elem_stride = 1; L2.group_stride = n; // float4 slots: two pairs
    LdsLayout L1; L1.ngroups = 4; L1.log_ngroups = 2; L1.elem_stride = 1; L1.group_stride = n; // float2 slots: four transforms
    auto report = [&](const char *name, float ms) { printf("%-44s %.3f ms\n", name, ms); };
    report("two-member, 256 thr, 16 waves/CU", timeit([&] { hipLaunchKernelGGL((kt<12, 256, 4, v2f>), dim3(blocks), dim3(256), lds, 0, sp, dtw, L2, d, loops); }));
    report("single-member, 512 thr, 16 waves/CU cap", timeit([&] { hipLaunchKernelGGL((kt<12, 512, 4, float>), dim3(blocks), dim3(512), lds, 0, sp, dtw, L1, d, loops); }));
    report("single-member, 512 thr, 24 waves/CU", timeit([&] { hipLaunchKernelGGL((kt<12, 512, 6, float>), dim3(blocks), dim3(512), lds, 0, sp, dtw, L1, d, loops); }));
    report("single-member, 512 thr, 32 waves/CU", timeit([&] { hipLaunchKernelGGL((kt<12, 512, 8, float>), dim3(blocks), dim3(512), lds, 0, sp, dtw, L1, d, loops); }));
    report("single-member, 256 thr x 2 items", timeit([&] { hipLaunchKernelGGL((kt<12, 256, 8, float>), dim3(blocks), dim3(256), lds, 0, sp, dtw, L1, d, loops); }));
    std::vector<float2> back(h.size());
    (void)hipMemcpy(back.data(), d, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
    printf("round trip check: %.4f %.4f (input %.4f %.4f)\n", back[5].x, back[5].y, h[5].x, h[5].y);
    return 0;
}

// tools/micro/rowload_latency.hip -- how long a k_rows-like block waits for its rows: 4 blocks of 256
// threads per CU, each loads four 9.6 KB rows (8 or 16 bytes per lane and instruction), idles for a
// given number of cycles ("compute"), stores two rows.  Reports the median cycles from the first load
// instruction to the arrival of the last byte, and the bandwidth the grid reached.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/rowload rowload_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int M2 = 1200;
template <int WIDE> __global__ __launch_bounds__(256, 4) void k(const float2 *x, const float2 *y, float2 *g, int nrows, int compute_cycles,
                                                                 unsigned *lat, int scattered)
{
    __shared__ float4 lds[2 * M2 + 64]; // 38.4 KB: four blocks per CU like k_rows
    const int task = blockIdx.x;
    // scattered: rows anywhere in the 1.4 GB buffers (TLB misses); otherwise like k_rows: the tasks of one pair
    // (601 of them) read rows of that pair's 1200-row matrix
    const unsigned long long pr = (unsigned long long)task / 601ull, kk = (unsigned long long)task % 601ull;
    const size_t pa = scattered ? (size_t)(((unsigned long long)task * 7919ull) % (unsigned long long)nrows) : (size_t)(pr * 1200ull + (kk * 463ull) % 1200ull);
    const size_t pb = scattered ? (size_t)(((unsigned long long)task * 104729ull + 13ull) % (unsigned long long)nrows) : (size_t)(pr * 1200ull + (1200ull - (kk * 463ull) % 1200ull) % 1200ull);
    const long long t0 = clock64();
    float acc = 0;
    if (WIDE) {
        const float4 *xa = (const float4 *)(x + pa * M2), *ya = (const float4 *)(y + pa * M2);
        const float4 *xb = (const float4 *)(x + pb * M2), *yb = (const float4 *)(y + pb * M2);
        float4 v[12];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int j = threadIdx.x + 256 * i;
            const bool ok = j < M2 / 2;
            v[4 * i + 0] = ok ? xa[j] : make_float4(0, 0, 0, 0); v[4 * i + 1] = ok ? ya[j] : make_float4(0, 0, 0, 0);
            v[4 * i + 2] = ok ? xb[j] : make_float4(0, 0, 0, 0); v[4 * i + 3] = ok ? yb[j] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    } else {
        const float2 *xa = x + pa * M2, *ya = y + pa * M2, *xb = x + pb * M2, *yb = y + pb * M2;
        float2 v[20];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = threadIdx.x + 256 * i;
            const bool ok = j < M2;
            v[4 * i + 0] = ok ? xa[j] : make_float2(0, 0); v[4 * i + 1] = ok ? ya[j] : make_float2(0, 0);
            v[4 * i + 2] = ok ? xb[j] : make_float2(0, 0); v[4 * i + 3] = ok ? yb[j] : make_float2(0, 0);
        }
#pragma unroll
        for (int i = 0; i < 20; i++) acc += v[i].x + v[i].y;
    }
    lds[threadIdx.x] = make_float4(acc, acc, acc, acc);
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) lat[task] = (unsigned)(t1 - t0);
    while (clock64() - t1 < compute_cycles) __builtin_amdgcn_s_sleep(8);
    const float4 r = lds[(threadIdx.x * 7) & 255];
    for (int j = threadIdx.x; j < M2; j += 256) {
        g[pa * M2 + j] = make_float2(r.x + j, r.y);
        g[pb * M2 + j] = make_float2(r.z, r.w + j);
    }
}
int main()
{
    const int nrows = 124 * 1200; // one launch group of the headline workload
    const size_t bytes = (size_t)nrows * M2 * sizeof(float2);
    float2 *x, *y, *g; unsigned *lat;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&y, bytes); (void)hipMalloc(&g, bytes);
    (void)hipMemset(x, 0, bytes); (void)hipMemset(y, 0, bytes);
    const int tasks = 124 * 601;
    (void)hipMalloc(&lat, tasks * sizeof(unsigned));
    std::vector<unsigned> h(tasks);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int scattered = 1; scattered >= 0; scattered--)
    for (int wide = 0; wide < 2; wide++)
        for (int cc : { 0, 15000, 30000 }) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(a);
                if (wide) hipLaunchKernelGGL(k<1>, dim3(tasks), dim3(256), 0, 0, x, y, g, nrows, cc, lat, scattered);
                else hipLaunchKernelGGL(k<0>, dim3(tasks), dim3(256), 0, 0, x, y, g, nrows, cc, lat, scattered);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
            }
            (void)hipMemcpy(h.data(), lat, tasks * sizeof(unsigned), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("%s %2d B/lane  compute %5d cycles: load wait median %6u p90 %6u cycles; kernel %.3f ms = %.2f TB/s\n", scattered ? "scattered" : "per-pair ", wide ? 16 : 8, cc,
                   h[tasks / 2], h[tasks * 9 / 10], ms, (double)tasks * 6 * M2 * 8 / ms / 1e9);
        }
    return 0;
}

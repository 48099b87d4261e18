// tools/micro/bfly_rate.hip -- how fast does gfx950 run the REAL butterfly code when nothing but
// the VALU is in the way?  Register-only loop over Bfly<R> + stage twiddles (lds_fft.h), no LDS, no
// HBM; W waves per SIMD.  Reports time per loop iteration; divide by the VALU instruction count
// of the loop body (hipcc -S) to get cycles per wave-instruction per SIMD.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -I../../old-audiosync_amd/csrc -o /tmp/bfly_rate bfly_rate.hip
#include "lds_fft.h"
#include <cstdio>

template <int R> __global__ __launch_bounds__(256, 4) void k(float4 *out, const float2 *tw, int iters)
{
    Cx2 v[R];
    for (int t = 0; t < R; t++) v[t] = Cx2{ v2f{ threadIdx.x * 0.001f + t, 1.f - t }, v2f{ 0.5f * t, threadIdx.x * 0.002f } };
    float2 w1 = tw[threadIdx.x & 63], w4 = tw[64 + (threadIdx.x & 63)];
    for (int it = 0; it < iters; it++) {
        float2 tww[R];
        stage_twiddles_from<R>(w1, w4, tww);
        Bfly<R, false>::run(v);
        static_for<1, R>([&](auto U) __attribute__((always_inline)) { v[U] = mulw(v[U], tww[U]); });
        // keep the twiddles loop-variant
        w1 = make_float2(w1.y, w1.x);
        w4 = make_float2(w4.y, w4.x);
    }
    Cx2 s = v[0];
    for (int t = 1; t < R; t++) s = s + v[t];
    out[blockIdx.x * 256 + threadIdx.x] = make_float4(s.re.x, s.re.y, s.im.x, s.im.y);
}

template <int R> static void run(float4 *d, float2 *tw, int waves_per_simd)
{
    const int iters = 4096, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(256), 0, 0, d, tw, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("radix %2d  waves/SIMD %d  %.3f ms  %.1f ns per iteration per wave -> x2.4 = %.0f cycles per wave-iteration per SIMD\n", R,
           waves_per_simd, ms, ms * 1e6 / iters / waves_per_simd, ms * 1e6 / iters / waves_per_simd * 2.4);
}

int main()
{
    float4 *d; float2 *tw;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float4));
    (void)hipMalloc(&tw, 128 * sizeof(float2));
    (void)hipMemset(tw, 0, 128 * sizeof(float2));
    for (int w : { 1, 2, 4 }) { run<10>(d, tw, w); run<12>(d, tw, w); run<4>(d, tw, w); }
    return 0;
}

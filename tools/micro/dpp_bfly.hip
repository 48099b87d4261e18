// tools/micro/dpp_bfly.hip -- what a CROSS-LANE radix-16 stage would cost against the in-register butterflies of lds_fft.h
// (VERDICT r3 #1b: "cross-lane (DPP row_*) butterflies for the innermost radix stage so that a 1200-point row needs two LDS
// exchanges at the three-pass kernel's register budget").  Register-only loops, W waves per SIMD, nothing but the VALU in
// the way:
//   in-register  one thread holds the 16 points of a butterfly (pair-planar Cx2, as the row kernels do): Bfly<16>
//   cross-lane   every lane holds ONE point of each of K independent 16-point transforms; a radix-16 step = two radix-4
//                steps inside quads / across quads of a 16-lane row with DPP operands (quad_perm broadcasts, row_shr/ror),
//                per-lane signs and the inner twiddle.  The data never leaves the registers -- and every lane computes one
//                output of a 4-point DFT from four inputs instead of four outputs from four inputs.
// Reports nanoseconds per complex POINT per radix-16 transform: the quantity that decides whether replacing an LDS exchange
// by lane exchanges pays.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -I../../old-audiosync_amd/csrc -o /tmp/dpp_bfly dpp_bfly.hip
#include "lds_fft.h"
#include <cstdio>

template <int CTRL> __device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// quad_perm broadcasts of lane q of every quad: 0x00, 0x55, 0xAA, 0xFF
// 4-point DFT across the lanes of a quad: lane j gets sum_k x_k (-i)^(jk)
__device__ __forceinline__ void quad_dft4(float &re, float &im, float sg, float cr, float ci)
{
    // a = x0 + sg x2, b = x1 + sg x3 (sg = +1 on even lanes, -1 on odd), out = a + (cr + i ci) b, (cr, ci) = (-i)^j
    const float are = fmaf(sg, dpp<0xAA>(re), dpp<0x00>(re)), aim = fmaf(sg, dpp<0xAA>(im), dpp<0x00>(im));
    const float bre = fmaf(sg, dpp<0xFF>(re), dpp<0x55>(re)), bim = fmaf(sg, dpp<0xFF>(im), dpp<0x55>(im));
    re = are + cr * bre - ci * bim;
    im = aim + cr * bim + ci * bre;
}
// the same across quads of a 16-lane row (lane j = quad index): row_shr / row_ror gathers
__device__ __forceinline__ void row_dft4(float &re, float &im, float sg, float cr, float ci)
{
    // partners at lane distance 4, 8, 12 inside the row: row_ror:4 = 0x124, row_ror:8 = 0x128, row_ror:12 = 0x12C
    const float x1r = dpp<0x124>(re), x2r = dpp<0x128>(re), x3r = dpp<0x12C>(re);
    const float x1i = dpp<0x124>(im), x2i = dpp<0x128>(im), x3i = dpp<0x12C>(im);
    const float are = fmaf(sg, x2r, re), aim = fmaf(sg, x2i, im);
    const float bre = fmaf(sg, x3r, x1r), bim = fmaf(sg, x3i, x1i);
    re = are + cr * bre - ci * bim;
    im = aim + cr * bim + ci * bre;
}

template <int K> __global__ __launch_bounds__(256, 4) void k_dpp(float4 *out, const float2 *tw, int iters)
{
    float re[2 * K], im[2 * K]; // K transforms x two members (pair-planar, like the row kernels)
    for (int t = 0; t < 2 * K; t++) { re[t] = threadIdx.x * 0.001f + t; im[t] = 1.f - t * 0.5f; }
    const int lane = threadIdx.x & 15;
    const float sg = (lane & 1) ? -1.f : 1.f, sg2 = (lane & 4) ? -1.f : 1.f;
    const float crs[4] = { 1.f, 0.f, -1.f, 0.f }, cis[4] = { 0.f, -1.f, 0.f, 1.f };
    const float cr = crs[lane & 3], ci = cis[lane & 3], cr2 = crs[(lane >> 2) & 3], ci2 = cis[(lane >> 2) & 3];
    float2 w = tw[threadIdx.x & 63]; // the inner twiddle w_16^(j1 j2): per lane
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 2 * K; t++) {
            quad_dft4(re[t], im[t], sg, cr, ci);
            const float r = re[t] * w.x - im[t] * w.y, i = re[t] * w.y + im[t] * w.x;
            re[t] = r; im[t] = i;
            row_dft4(re[t], im[t], sg2, cr2, ci2);
        }
        w = make_float2(w.y, w.x);
    }
    float s = 0;
    for (int t = 0; t < 2 * K; t++) s += re[t] + im[t];
    out[blockIdx.x * 256 + threadIdx.x] = make_float4(s, s, s, s);
}

__global__ __launch_bounds__(256, 4) void k_reg16(float4 *out, const float2 *tw, int iters)
{
    Cx2 v[16];
    for (int t = 0; t < 16; t++) v[t] = Cx2{ v2f{ threadIdx.x * 0.001f + t, 1.f - t }, v2f{ 0.5f * t, threadIdx.x * 0.002f } };
    for (int it = 0; it < iters; it++) Bfly<16, false>::run(v);
    Cx2 s = v[0];
    for (int t = 1; t < 16; t++) s = s + v[t];
    out[blockIdx.x * 256 + threadIdx.x] = make_float4(s.re.x, s.re.y, s.im.x, s.im.y);
}
__global__ __launch_bounds__(256, 4) void k_reg10(float4 *out, const float2 *tw, int iters)
{
    Cx2 v[10];
    for (int t = 0; t < 10; t++) v[t] = Cx2{ v2f{ threadIdx.x * 0.001f + t, 1.f - t }, v2f{ 0.5f * t, threadIdx.x * 0.002f } };
    for (int it = 0; it < iters; it++) Bfly<10, false>::run(v);
    Cx2 s = v[0];
    for (int t = 1; t < 10; t++) s = s + v[t];
    out[blockIdx.x * 256 + threadIdx.x] = make_float4(s.re.x, s.re.y, s.im.x, s.im.y);
}

template <typename F> static float run(F launch)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); }
    return ms;
}
int main()
{
    float4 *d; float2 *tw;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float4)); (void)hipMalloc(&tw, 128 * sizeof(float2)); (void)hipMemset(tw, 0, 128 * sizeof(float2));
    const int iters = 4096;
    for (int w : { 1, 2, 4 }) {
        const int blocks = 256 * w; // 256 CUs x w blocks of 4 waves = w waves per SIMD
        // points per thread and iteration: in-register 16 (x2 members), cross-lane K (x2 members)
        float ms = run([&] { hipLaunchKernelGGL(k_reg16, dim3(blocks), dim3(256), 0, 0, d, tw, iters); });
        printf("waves/SIMD %d  in-register radix 16 (Bfly<16>, pair-planar): %.3f ms  %.3f ns per point and wave\n", w, ms, ms * 1e6 / iters / 32.0);
        ms = run([&] { hipLaunchKernelGGL(k_reg10, dim3(blocks), dim3(256), 0, 0, d, tw, iters); });
        printf("waves/SIMD %d  in-register radix 10 (Bfly<10>, pair-planar): %.3f ms  %.3f ns per point and wave\n", w, ms, ms * 1e6 / iters / 20.0);
        ms = run([&] { hipLaunchKernelGGL(k_dpp<5>, dim3(blocks), dim3(256), 0, 0, d, tw, iters); });
        printf("waves/SIMD %d  cross-lane radix 16 (two DPP radix-4 steps + inner twiddle), 5 points per lane x 2 members: %.3f ms  %.3f ns per point and wave\n", w, ms, ms * 1e6 / iters / 10.0);
    }
    return 0;
}

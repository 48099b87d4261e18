// tools/micro/valu_rate.hip -- issue rate of scalar vs packed fp32 FMA on gfx950 (diagnostic).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    if (MODE == 0) {
        float x[16];
        for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
        }
        float s = 0; for (int i = 0; i < 16; i++) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        v2f x[8];
        for (int i = 0; i < 8; i++) x[i] = v2f{ threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i };
        const v2f va = { a, a * 1.0001f }, vb = { b, b * 0.9999f };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = __builtin_elementwise_fma(x[i], va, vb);
        }
        v2f s = { 0, 0 }; for (int i = 0; i < 8; i++) s += x[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    }
}

int main()
{
    float *d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    const int iters = 4096;
    for (int mode = 0; mode < 2; mode++) {
        for (int blocks : { 1024, 2048, 4096, 8192 }) {   // 4, 8, 16, 32 waves per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)blocks * 256 * iters * 16;          // scalar-equivalent FMAs
            const double instr = fma / (mode == 0 ? 64.0 : 128.0);          // wave instructions
            printf("%s blocks=%5d  %.3f ms  %.1f TFLOP/s  %.2f cycles/wave-instr/SIMD @2.4GHz\n",
                   mode == 0 ? "v_fma_f32   " : "v_pk_fma_f32", blocks, ms, 2 * fma / ms / 1e9,
                   ms * 1e-3 * 2.4e9 * 1024 / instr);
        }
    }
    return 0;
}

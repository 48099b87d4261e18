// tools/micro/pk_forms.hip -- issue cost and dependent-issue latency of the packed fp32 forms the
// transforms use (operand-select / negate modifiers included) vs scalar fp32, on gfx950 (diagnostic).
// build: hipcc --offload-arch=gfx950 -O3 -o pk_forms pk_forms.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));


template <int FORM, int CHAINS> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    cf x[CHAINS];
    for (int i = 0; i < CHAINS; i++) x[i] = cf{ threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i };
    const cf va = { a, a * 1.0001f }, vb = { b, b * 0.9999f };
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
      for (int rep = 0; rep < 16; rep++)
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (FORM == 0) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(va.x), "v"(vb.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].y) : "v"(va.y), "v"(vb.y));
            } else if (FORM == 6) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(vb.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i].y) : "v"(vb.y));
            } else if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
            else if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(x[i]) : "v"(va), "v"(vb));
            else if (FORM == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(vb));
            else if (FORM == 4) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(x[i]) : "v"(vb));
            else if (FORM == 5) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(x[i]) : "v"(va));
            else if (FORM == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(va));
        }
    }
    cf s = { 0, 0 };
    for (int i = 0; i < CHAINS; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

static const char *names[] = { "v_fma_f32 x2 (same flops)", "v_pk_fma_f32", "v_pk_fma_f32 op_sel+neg", "v_pk_add_f32",
                               "v_pk_add_f32 op_sel+neg", "v_pk_mul_f32 op_sel_hi", "v_add_f32 x2", "v_pk_mul_f32" };

template <int FORM, int CHAINS> static void run(float *d, int waves_per_simd)
{
    const int iters = 1 << 16;
    const int blocks = 256 * waves_per_simd; // 256 CUs, a 256-thread block = one wave on each SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FORM, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // "pair-op": one packed instruction or two scalar ones
    const double pair_ops_per_simd = (double)waves_per_simd * iters * CHAINS;
    printf("%-28s chains=%d waves/SIMD=%d  %8.3f ms  %.2f ns per pair-op per SIMD (%.2f cycles @2.4GHz)\n", names[FORM], CHAINS,
           waves_per_simd, ms, ms * 1e6 / pair_ops_per_simd, ms * 1e-3 * 2.4e9 / pair_ops_per_simd);
}

template <int FORM> static void all(float *d)
{
    run<FORM, 8>(d, 8);
    run<FORM, 8>(d, 4);
    run<FORM, 8>(d, 2);
    run<FORM, 8>(d, 1);
    run<FORM, 2>(d, 4);
    run<FORM, 2>(d, 1);
    run<FORM, 1>(d, 1);
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    all<0>(d); all<6>(d); all<1>(d); all<2>(d); all<4>(d); all<5>(d);
    return 0;
}

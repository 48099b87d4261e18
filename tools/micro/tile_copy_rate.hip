// tools/micro/tile_copy_rate.hip -- does the WRITE pattern of k_fwd_cols_r cost anything?  A block reads what that kernel
// reads (1200 rows x 64 bytes at a 9600-byte pitch for the source, 600 rows for the sample) and writes what it writes
// (601 rows x 128 bytes of float2 per spectrum), either row-major as today -- 128-byte pieces at a 19 200-byte pitch -- or
// TILE-major: the 601 pieces of a tile contiguous (76.9 KB per block).  Same block size and LDS request as the kernel.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/tile_copy_rate tile_copy_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float4 lds[];
__global__ __launch_bounds__(512) void k_copy(const float4 *src, const float4 *smp, float4 *cx, float4 *cy, int tile_major)
{
    const int M2 = 2400, M1 = 600, ntiles = M2 / 16;
    const int pair = blockIdx.z, is_smp = blockIdx.y, tile = blockIdx.x;
    const float4 *in = is_smp ? smp + (size_t)pair * (M1 * M2 / 4) : src + (size_t)pair * (2 * M1 * M2 / 4); // floats / 4
    const int rows_in = is_smp ? M1 : 2 * M1;        // real rows with data
    const size_t in_pitch4 = M2 / 4;                  // float4 per input row
    float4 acc = make_float4(0, 0, 0, 0);
    // 64-byte piece = 4 float4 per row
    for (int e = threadIdx.x; e < rows_in * 4; e += 512) {
        const int r = e >> 2, c = e & 3;
        const float4 v = in[(size_t)r * in_pitch4 + tile * 4 + c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        lds[e & 4095] = v;
    }
    __syncthreads();
    float4 *out = (is_smp ? cy : cx) + (size_t)pair * ((M1 + 1) * (size_t)M2 / 2); // float2 count / 2 = float4 count
    // 128-byte piece = 8 float4 per output row, 601 rows
    for (int e = threadIdx.x; e < (M1 + 1) * 8; e += 512) {
        const int r = e >> 3, c = e & 7;
        const float4 v = make_float4(acc.x + lds[e & 4095].x, acc.y, acc.z, acc.w + r);
        if (tile_major) out[(size_t)tile * ((M1 + 1) * 8) + e] = v;
        else out[(size_t)r * (M2 / 2) + tile * 8 + c] = v;
    }
}
int main()
{
    const int pairs = 124, M1 = 600, M2 = 2400;
    const size_t src4 = (size_t)pairs * 2 * M1 * M2 / 4, smp4 = (size_t)pairs * M1 * M2 / 4, c4 = (size_t)pairs * (M1 + 1) * M2 / 2;
    float4 *src, *smp, *cx, *cy;
    (void)hipMalloc(&src, src4 * 16); (void)hipMalloc(&smp, smp4 * 16); (void)hipMalloc(&cx, c4 * 16); (void)hipMalloc(&cy, c4 * 16);
    (void)hipMemset(src, 1, src4 * 16); (void)hipMemset(smp, 1, smp4 * 16);
    (void)hipFuncSetAttribute((const void *)k_copy, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    const double bytes = (double)(src4 + smp4 + 2 * c4) * 16;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++)
        for (int tm = 0; tm < 2; tm++) {
            float best = 1e9f;
            for (int r = 0; r < 5; r++) {
                float ms = 0;
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k_copy, dim3(M2 / 16, 2, pairs), dim3(512), 76800, 0, src, smp, cx, cy, tm);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("k_fwd_cols_r's traffic (%.2f GB), output %s: %.3f ms  %.2f TB/s\n", bytes / 1e9, tm ? "TILE-major (contiguous 76.9 KB per block)" : "row-major (128-byte pieces, 19 200-byte pitch)", best, bytes / best / 1e9);
        }
    return 0;
}

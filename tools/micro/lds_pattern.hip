// tools/micro/lds_pattern.hip -- ground truth for the LDS bank model: replays per-lane LDS address
// patterns (one table per dispatch, written by tools/lds_patterns.py) as ds_read_b128 / ds_write_b128 /
// ds_write_b64 / ds_read_b64 and lets rocprofv3 count SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT
// per dispatch.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_pattern lds_pattern.hip
// Run:   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d out -- ./lds_pattern patterns.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define MAXI 12
#define NT 256
extern __shared__ __attribute__((aligned(16))) char lds[];

// kind: 0 r128, 1 w128, 2 w64, 3 r64
__global__ __launch_bounds__(NT) void k_lds(const int *tab, int ninstr, int kind, int iters, float *sink)
{
    int a[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; k++) a[k] = k < ninstr ? tab[k * NT + threadIdx.x] : -1;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f4 acc = { 0.f, 0.f, 0.f, 0.f };
    const f4 v = { (float)threadIdx.x, 1.f, 2.f, 3.f };
    const f2 v2 = { (float)threadIdx.x, 1.f };
    const int base = (int)(size_t)lds; // LDS byte offset of the dynamic segment (0)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < MAXI; k++) {
            if (a[k] >= 0) {
                const int ad = base + a[k];
                if (kind == 0) { f4 x; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(ad) : "memory"); acc += x; }
                else if (kind == 1) asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(v) : "memory");
                else if (kind == 2) asm volatile("ds_write_b64 %0, %1" :: "v"(ad), "v"(v2) : "memory");
                else { f2 x; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(ad) : "memory"); acc.x += x.x; acc.y += x.y; }
            }
        }
        __syncthreads();
    }
    if (acc.x == 12345.f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char **argv)
{
    FILE *f = fopen(argc > 1 ? argv[1] : "patterns.bin", "rb");
    if (!f) { fprintf(stderr, "no pattern file\n"); return 1; }
    int npat = 0;
    if (fread(&npat, 4, 1, f) != 1) return 1;
    float *sink; hipMalloc(&sink, 16);
    (void)hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int p = 0; p < npat; p++) {
        int hdr[4]; // ninstr, kind, lds_bytes, reserved
        if (fread(hdr, 4, 4, f) != 4) return 1;
        std::vector<int> t((size_t)hdr[0] * NT);
        if (fread(t.data(), 4, t.size(), f) != t.size()) return 1;
        int *d; hipMalloc(&d, t.size() * 4);
        hipMemcpy(d, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_lds, dim3(256), dim3(NT), (size_t)hdr[2], 0, d, hdr[0], hdr[1], 64, sink);
        hipDeviceSynchronize();
        hipFree(d);
    }
    printf("ran %d patterns\n", npat);
    return 0;
}

// tools/micro/rows_copy_rate.hip -- the rate at which this chip moves k_rows_r's traffic and nothing else: per block row k1 of
// both spectra in (2 x 2400 float2 = 38.4 KB, non-temporal 16-byte loads) and one row of Q out (19.2 KB, non-temporal stores),
// 601 rows x 124 pairs = 74 524 blocks of 256 threads, 4.29 GB per launch.  Forms:
//   A = load all / barrier / store all through the kernel's 38.4 KB of LDS (four blocks per CU, the kernel's occupancy)
//   S = every thread streams its pieces straight through (no LDS, no barrier), at 38.4 KB, 19.2 KB and 0 of LDS request
// and 8-byte stores (the kernel's: its last stage leaves single float2 outputs 100 columns apart) against 16-byte stores.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/rows_copy_rate tools/micro/rows_copy_rate.hip && /tmp/rows_copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float4 lds[];
constexpr int M2 = 2400, NROWS = 601, NT = 256;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldnt(const float4 *p)
{
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int FORM, bool ST8>
__global__ __launch_bounds__(NT) void k_rows_copy(const float4 *__restrict__ cx, const float4 *__restrict__ cy, float4 *__restrict__ q, int use_lds)
{
    const size_t row4 = (size_t)blockIdx.x * (M2 / 2); // float4 per row = 1200
    const int tid = threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    if (FORM == 0) {
        float4 a[5], b[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int e = tid + i * NT;
            a[i] = b[i] = make_float4(0, 0, 0, 0);
            if (e < M2 / 2) { a[i] = ldnt(cx + row4 + e); b[i] = ldnt(cy + row4 + e); }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int e = tid + i * NT;
            if (e < M2 / 2) { lds[2 * e] = a[i]; lds[2 * e + 1] = b[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int e = tid + i * NT;
            if (e < M2 / 2) {
                const float4 u = lds[(2 * e + 7) % M2], w = lds[(2 * e + 1001) % M2];
                const float4 v = make_float4(u.x + w.x, u.y, u.z, w.w);
                if (ST8) { // consecutive lanes store consecutive float2, as the kernel's last stage does (two pieces 1280 float2 apart)
                    f2v lo, hi; lo.x = v.x; lo.y = v.y; hi.x = v.z; hi.y = v.w;
                    if (e < 1200) {
                        __builtin_nontemporal_store(lo, reinterpret_cast<f2v *>(q + row4) + e);
                        __builtin_nontemporal_store(hi, reinterpret_cast<f2v *>(q + row4) + 1200 + e);
                    }
                } else {
                    f4v o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
                    __builtin_nontemporal_store(o, reinterpret_cast<f4v *>(q + row4 + e));
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int e = tid + i * NT;
            if (e < M2 / 2) {
                const float4 u = ldnt(cx + row4 + e), w = ldnt(cy + row4 + e);
                const float4 v = make_float4(u.x + w.x, u.y, u.z, w.w);
                if (ST8) { // consecutive lanes store consecutive float2, as the kernel's last stage does (two pieces 1280 float2 apart)
                    f2v lo, hi; lo.x = v.x; lo.y = v.y; hi.x = v.z; hi.y = v.w;
                    if (e < 1200) {
                        __builtin_nontemporal_store(lo, reinterpret_cast<f2v *>(q + row4) + e);
                        __builtin_nontemporal_store(hi, reinterpret_cast<f2v *>(q + row4) + 1200 + e);
                    }
                } else {
                    f4v o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
                    __builtin_nontemporal_store(o, reinterpret_cast<f4v *>(q + row4 + e));
                }
            }
        }
    }
    (void)acc; (void)use_lds;
}
template <int FORM, bool ST8> static void run(const char *name, const float4 *cx, const float4 *cy, float4 *q, int pairs, double bytes)
{
    const void *fn = (const void *)k_rows_copy<FORM, ST8>;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int lds_req[3] = { 38400, 19200, 0 };
    printf("%-40s", name);
    for (int li = 0; li < 3; li++) {
        if (FORM == 0 && li > 0) break;
        int per_cu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NT, lds_req[li]);
        float best = 1e9f;
        for (int r = 0; r < 7; r++) {
            float ms = 0;
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((k_rows_copy<FORM, ST8>), dim3(NROWS * pairs), dim3(NT), lds_req[li], 0, cx, cy, q, lds_req[li]);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
            if (r >= 2 && ms < best) best = ms;
        }
        printf("  %6.3f ms %5.2f TB/s (%d/CU)", best, bytes / best / 1e9, per_cu);
    }
    printf("\n");
}
int main()
{
    const int pairs = 124;
    const size_t c4 = (size_t)pairs * NROWS * M2 / 2;
    float4 *cx, *cy, *q;
    (void)hipMalloc(&cx, c4 * 16); (void)hipMalloc(&cy, c4 * 16); (void)hipMalloc(&q, c4 * 16);
    (void)hipMemset(cx, 1, c4 * 16); (void)hipMemset(cy, 1, c4 * 16);
    const double bytes = 3.0 * c4 * 16;
    printf("k_rows_r's traffic at 600 x 2400, 124 pairs: %.3f GB per launch; columns: LDS request 38.4 KB | 19.2 KB | 0\n", bytes / 1e9);
    for (int rep = 0; rep < 2; rep++) {
        run<0, true>("form A, 8-byte stores (the kernel's)", cx, cy, q, pairs, bytes);
        run<0, false>("form A, 16-byte stores", cx, cy, q, pairs, bytes);
        run<1, true>("form S, 8-byte stores", cx, cy, q, pairs, bytes);
        run<1, false>("form S, 16-byte stores", cx, cy, q, pairs, bytes);
    }
    return 0;
}

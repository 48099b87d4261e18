// tools/micro/hbm_rate.hip -- what this chip's HBM sustains for plain streaming (read-only sum, copy), and for the
// column-tile pattern of the transform kernels (64-byte pieces, one per 9600-byte row), as a yardstick for DESIGN.md.
// build (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_rate hbm_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const float4 *in, float *out, size_t n4)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const float4 *in, float4 *out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// tile pattern: matrix [rows][row_f4] of float4; a block moves a tile of 4 float4 (64 B) x all rows of one matrix
// pitch_f4 >= row_f4: row pitch in float4 (padded pitches: does a row stride that is not 9600 bytes move faster?)
__global__ __launch_bounds__(512) void k_tiles(const float4 *in, float4 *out, int rows, int pitch_f4, int tiles_per_mat, int ntiles)
{
    const int mat = blockIdx.x / tiles_per_mat, b = blockIdx.x % tiles_per_mat;
    const int tile = (b & ~15) + 2 * (b & 7) + ((b >> 3) & 1); // the XCD-aware pairing of xcorr_kernels.hip
    if (tile >= ntiles) return;
    const size_t base = (size_t)mat * rows * pitch_f4 + (size_t)tile * 4;
    for (int e = threadIdx.x; e < rows * 4; e += 512) {
        const int r = e >> 2, c = e & 3;
        out[base + (size_t)r * pitch_f4 + c] = in[base + (size_t)r * pitch_f4 + c];
    }
}
template <typename F> static float ms_of(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 3; r++) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); }
    return ms;
}
int main()
{
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    float4 *a, *b; float *o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 64);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
    for (int blocks : { 2048, 8192, 32768 }) {
        float t = ms_of([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, o, n4); });
        printf("read  %6d blocks: %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
        t = ms_of([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n4); });
        printf("copy  %6d blocks: %.3f ms  %.2f TB/s (read+write)\n", blocks, t, 2.0 * bytes / t / 1e9);
    }
    // 1200 x 1200 complex per matrix = 1200 rows x 600 float4; 150 tiles (padded to 160 blocks); as many matrices as fit in 2 GiB
    const int rows = 1200, row_f4 = 600, tiles = 150, tpm = 160;
    for (int pad : { 0, 4, 8, 16, 24, 40, 72, 136 }) { // extra float4 per row: pitch 9600 B + 64, 128, 256, ... bytes
        const int pitch = row_f4 + pad;
        const int mats = (int)(bytes / ((size_t)rows * pitch * 16));
        float t = ms_of([&] { hipLaunchKernelGGL(k_tiles, dim3(mats * tpm), dim3(512), 0, 0, a, b, rows, pitch, tpm, tiles); });
        printf("tiles pitch %5d B, %6d blocks: %.3f ms  %.2f TB/s (read+write, 64-byte pieces per row, paired tiles on one XCD)\n",
               pitch * 16, mats * tpm, t, 2.0 * mats * (double)rows * row_f4 * 16 / t / 1e9);
    }
    return 0;
}

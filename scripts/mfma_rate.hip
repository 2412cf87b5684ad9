// build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate scripts/mfma_rate.hip && /tmp/mfma_rate
// Issue rate of v_mfma_f32_32x32x16_f16 (and 16x16x32) on gfx950: cycles per MFMA for 1 / 2 / 4 waves per SIMD, independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool SMALL>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[NACC];
    f32x4 acs[NACC];
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) acc[n][r] = 0.f; for (int r = 0; r < 4; ++r) acs[n][r] = 0.f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (SMALL) acs[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acs[n], 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) s += acc[n][r]; for (int r = 0; r < 4; ++r) s += acs[n][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC, bool SMALL>
void run(const char* name, int waves_per_block) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 20);
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (iters * NACC);
    const double flops = (SMALL ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16) * (double)iters * NACC * blocks * waves_per_block;
    printf("%-34s waves/block %2d: %6.1f s_memtime ticks per MFMA per wave, kernel %.3f ms -> %.2f PF/s over %d blocks\n", name, waves_per_block, per, ms, flops / (ms * 1e-3) / 1e15, blocks);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {4, 8, 16}) run<4, false>("32x32x16 f16, 4 independent acc", w);
    for (int w : {4, 8}) run<1, false>("32x32x16 f16, 1 acc (dependent)", w);
    for (int w : {4, 8, 16}) run<8, true>("16x16x32 f16, 8 independent acc", w);
    return 0;
}

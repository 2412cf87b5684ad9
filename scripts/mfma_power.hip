// build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power scripts/mfma_power.hip && /tmp/mfma_power <mode> <seconds>
// What the matrix pipe sustains AT THE PART'S POWER CAP (the headline runs at 1 350-1 370 W of 1 400): a chip-filling loop of MFMAs on operands
// with random bits, for a few seconds, while scripts/mfma_power.sh polls rocm-smi beside it.  mode 0: v_mfma_f32_32x32x16_f16 (the product path's
// instruction); mode 1: v_mfma_i32_32x32x32_i8 (twice the multiply-adds per instruction: what an integer-plane split would run on); mode 2: f16 with ZERO operands
// (how much of the power is data toggling).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    i32x4 ra[4], rb[4];   // four operand pairs with random bits, rotated through the loop
    for (int p = 0; p < 4; ++p)
        for (int i = 0; i < 4; ++i) {
            unsigned ha = hash(id * 64 + p * 8 + i), hb = hash(id * 64 + p * 8 + 4 + i);
            if (MODE == 0) {   // keep the fp16 lanes finite and of order one: clear the top exponent bit of each half
                ha &= 0xbfffbfffU;
                hb &= 0xbfffbfffU;
            }
            if (MODE == 2) ha = hb = 0;
            if (MODE == 3 || MODE == 4 || MODE == 5) {
                ha &= 0xbfffbfffU;
                hb &= 0xbfffbfffU;
            }
            if (MODE == 5) hb &= 0xffe0ffe0U;   // five of ten stored mantissa bits
            ra[p][i] = (int)ha;
            rb[p][i] = (int)hb;
        }
    f32x16 accf[4];
    i32x16 acci[4];
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) { accf[n][r] = 0.f; acci[n][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (MODE == 1) acci[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[p], rb[(p + n) & 3], acci[n], 0, 0, 0);
                else if (MODE == 3) accf[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(p + n) & 3]), __builtin_bit_cast(f16x8, rb[(p + 3 * n + 1) & 3]), accf[n], 0, 0, 0);   // both operands change with every instruction
                else if (MODE == 4) accf[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[p]), __builtin_bit_cast(f16x8, rb[p]), accf[n], 0, 0, 0);   // both operands held over four instructions
                else if (MODE == 5) accf[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[p]), __builtin_bit_cast(f16x8, rb[(p + n) & 3]) , accf[n], 0, 0, 0);   // (mode 5: operand B with only its top five mantissa bits -- see main)
                else accf[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[p]), __builtin_bit_cast(f16x8, rb[(p + n) & 3]), accf[n], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) s += accf[n][r] + (float)acci[n][r];
    out[id] = s;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    float* out;
    hipMalloc(&out, 1 << 24);
    const int blocks = 256 * 2, iters = 20000;   // two four-wave workgroups per CU: two waves per SIMD
    auto launch = [&]() {
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
    };
    launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms1;
    hipEventElapsedTime(&ms1, e0, e1);
    const int reps = (int)(seconds * 1e3 / ms1) + 1;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs_per = mode == 1 ? 32.0 * 32 * 32 : 32.0 * 32 * 16;
    const double ops = 2.0 * macs_per * 16.0 * iters * (double)blocks * 4 * reps;
    printf("mode %d (%s): %d launches in %.1f ms -> %.3f P(FL)OP/s sustained over the run\n", mode,
           mode == 1 ? "v_mfma_i32_32x32x32_i8, random bits" : mode == 2 ? "v_mfma_f32_32x32x16_f16, zero operands" : mode == 3 ? "f16, both operands change with every instruction" :
           mode == 4 ? "f16, both operands held over four instructions" : mode == 5 ? "f16, operand B with five mantissa bits" : "v_mfma_f32_32x32x16_f16, random bits (A held over four instructions)", reps, ms, ops / (ms * 1e-3) / 1e15);
    return 0;
}

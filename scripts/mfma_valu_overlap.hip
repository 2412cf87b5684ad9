// Do VALU instructions (SiLU-class: v_exp / v_rcp / fma) execute in the shadow of fp16 MFMAs on gfx950 -- (a) interleaved in ONE wave, (b) from
// ANOTHER wave of the same SIMD?  Decides whether an epilogue can hide behind a main loop, and how (round 3).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define NM 4   // independent accumulators
template <int MODE, int PRIO = 0, int SEL = 0>   // SEL: which waves run the MFMA loop in mode 3 (0: waves 0-3, 1: even waves, 2: waves with bit 1 clear); PRIO: s_setprio of the VALU waves in mode 3
// 0: MFMA only, 1: VALU only, 2: both interleaved in one wave, 3: even waves MFMA / odd waves VALU (two waves per SIMD)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[NM];
    for (int m = 0; m < NM; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.01f * (lane + i);
    if (MODE == 3) {   // top-level split: waves 0-3 run a pure MFMA loop, waves 4-7 a pure VALU loop (one of each per SIMD)
        if ((SEL % 10) == 0 ? wave < 4 : (SEL % 10) == 1 ? (wave & 1) == 0 : (SEL % 10) == 2 ? (wave & 2) == 0 : (SEL % 10) == 3 ? true : false) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            }
        } else {
            if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (SEL >= 10) {   // plain full-rate VALU work (no transcendental unit): 12 dependent fmas per value
#pragma unroll
                        for (int t = 0; t < 12; ++t) v[q] = __builtin_fmaf(v[q], 0.999f, 0.001f);
                    } else {
                        v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[q] * -1.44269504f)) + 0.25f;
                    }
                }
            }
        }
        float s3 = 0.f;
        for (int m = 0; m < NM; ++m) for (int r = 0; r < 16; ++r) s3 += acc[m][r];
        for (int i = 0; i < 8; ++i) s3 += v[i];
        if (s3 == 123.456f) out[0] = s3;
        return;
    }
    const bool do_m = MODE == 0 || MODE == 2;
    const bool do_v = MODE == 1 || MODE == 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (do_m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            if (do_v) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {   // two SiLUs per MFMA slot: 2 x (mul, exp, add, rcp, mul) ~ 22 issue slots of 4 cycles = 88 cycles vs 32 of MFMA
                    const int q = (2 * m + i) & 7;
                    v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[q] * -1.44269504f)) + 0.25f;
                }
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < NM; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}
template <int MODE, int PRIO = 0, int SEL = 0>
static float run(int waves_per_wg, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, PRIO, SEL>), dim3(256), dim3(64 * waves_per_wg), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const int iters = 20000;
    printf("one wave per SIMD (4-wave workgroups, 256 workgroups), %d iterations x %d MFMA slots:\n", iters, NM);
    const float m1 = run<0>(4, iters), v1 = run<1>(4, iters), b1 = run<2>(4, iters);
    printf("  MFMA only %.2f ms   VALU only (2 SiLU per slot) %.2f ms   interleaved in one wave %.2f ms   (sum %.2f, max %.2f)\n", m1, v1, b1, m1 + v1, m1 > v1 ? m1 : v1);
    printf("two waves per SIMD (8-wave workgroups):\n");
    const float m2 = run<0>(8, iters), v2 = run<1>(8, iters), b2 = run<2>(8, iters), s2 = run<3>(8, iters);
    printf("  MFMA only %.2f ms   VALU only %.2f ms   both in every wave %.2f ms   one wave MFMA + one wave VALU per SIMD %.2f ms\n", m2, v2, b2, s2);
    const float p1 = run<3, 1>(8, iters), p3 = run<3, 3>(8, iters);
    printf("  ... with the VALU waves at s_setprio 1: %.2f ms   at s_setprio 3: %.2f ms\n", p1, p3);
    const float e0 = run<3, 0, 1>(8, iters), e1 = run<3, 0, 2>(8, iters), e3 = run<3, 3, 1>(8, iters);
    printf("  MFMA on the even waves: %.2f ms   on waves 0,1,4,5: %.2f ms   even waves, VALU waves at s_setprio 3: %.2f ms\n", e0, e1, e3);
    // plain fma work instead of exp / rcp in the VALU waves: SEL 13 = MFMA only (all waves), 14 = fma only, 10 / 11 = one of each per SIMD under the two mappings
    const float f_m = run<3, 0, 13>(8, iters), f_v = run<3, 0, 14>(8, iters), f0 = run<3, 0, 10>(8, iters), f1 = run<3, 0, 11>(8, iters);
    printf("plain-fma VALU waves: MFMA in all eight waves %.2f ms   fma in all eight %.2f ms   waves 0-3 MFMA + 4-7 fma %.2f ms   even MFMA + odd fma %.2f ms\n", f_m, f_v, f0, f1);
    return 0;
}

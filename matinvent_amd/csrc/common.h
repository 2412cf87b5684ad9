// Shared device/host helpers for the gfx950 hot path.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/matinvent_hip.h"
#include "../../include/matinvent_hip_debug.h"

namespace mi {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define MI_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            mi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                       \
            return MI_EHIP;                                                                \
        }                                                                                  \
    } while (0)

#define MI_CHECK(cond, code, ...)       \
    do {                                \
        if (!(cond)) {                  \
            mi::set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

#define MI_TRY(expr)            \
    do {                        \
        int _r = (expr);        \
        if (_r != MI_OK) return _r; \
    } while (0)

#define MI_KERNEL_CHECK() MI_HIP(hipGetLastError())

// ---- issued matrix-pipe work (measurement only) -----------------------------------------------------------------------------------
// Every host launcher of a matrix product adds 2 M N K x (MFMA terms per fp32 product) to a process-wide counter (16-bit pipe: fp16 / bf16
// MFMA) or, for the f32-input MFMA kernels, 2 M N K to a second one.  bench.py divides the counters of a timed region by its elapsed time:
// the matrix-pipe rate of the WHOLE step (forwards + backward), not of one bracketed kernel pair.  mi_debug_mfma_flops reads / resets them.
void count_mfma(int64_t M, int64_t N, int64_t K, int terms16);   // terms16 = 0: an f32-input MFMA product
#if !defined(MI_PLANES_TERMS)
#define MI_PLANES_TERMS (MI_PLANES_FP16 ? (MI_TF32_CLASS ? 1 : 3) : 6)   // MFMA terms of a product of two pre-split plane sets in this build
#endif

// ---- roctx ranges (tracing hooks around the sampler step, the fine-tune micro-step and the gradient all-reduce) --------
// Off unless MI_ROCTX=1 is set when the library is loaded; the roctx library is opened at run time (no link dependency), so a
// `rocprofv3 --marker-trace` run shows one named range per denoising step / micro-step / all-reduce over the kernels it enqueues.
void trace_push(const char* name);
void trace_pop();
struct TraceRange {
    explicit TraceRange(const char* name) { trace_push(name); }
    ~TraceRange() { trace_pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device math ----------------------------------------------------------------------
// torch.nn.functional.silu: x / (1 + exp(-x))
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// SiLU on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): x * rcp(1 + 2^(-x log2 e)).
// ~3 ulp from the IEEE-division form, far inside the stated fp32 tolerance; 5 VALU ops instead of ~30,
// which keeps the unrolled edge kernel inside the instruction cache.
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
// silu(x) * s for a power-of-two s, given inv_s = 1 / s, at the cost of silu itself: (1 + e) / s = fma(e, inv_s, inv_s) rounds as 1 + e
// does (scaling by a power of two commutes with rounding), v_rcp_f32 of a power-of-two multiple is the same multiple of the reciprocal
// (it works on the mantissa), and so is the final product.  (The plane-set stores scale every activation: this saves their multiply.)
__device__ __forceinline__ float silu_fast_scaled(float x, float inv_s) {
    return x * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(x * -1.44269504088896340736f), inv_s, inv_s));
}
// d silu / dx given x
__device__ __forceinline__ float silu_grad(float x) {
    float s = 1.0f / (1.0f + expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
// the same on the hardware transcendentals (the backward's streaming passes over [E, H]: a few ulp, far inside the gradient tolerance)
__device__ __forceinline__ float silu_grad_fast(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
    return s * (1.0f + x * (1.0f - s));
}
// python / torch `x % 1.` for floats (torch.remainder): result takes the sign of the divisor
__device__ __forceinline__ float pymod1(float x) {
    // (x - trunc(x) IS fmodf(x, 1): both are exact; the library's fmodf is a thirty-instruction loop, this is two)
    float r = x - truncf(x);
    if (r < 0.0f) r += 1.0f;
    return r;
}

// Branch-free sin & cos for |x| <~ 1e4 (Fourier arguments are bounded by 2*pi*F*|d|): three-term
// Cody-Waite reduction by pi/2 with FMAs, then the classic degree-7/8 minimax kernels on
// [-pi/4, pi/4].  Max abs error ~1.2e-7 over the range (measured against fp64 in
// tests/test_gpu_forward.py), i.e. libm-class; no slow-path branch, ~30 VALU ops.
__device__ __forceinline__ void sincos_bounded(float x, float* sn, float* cs) {
    const float n = rintf(x * 0.63661977236758134308f);  // x * 2/pi
    float r = fmaf(-n, 1.5707962513e+00f, x);            // pi/2 split: 0x3fc90fda, 0x33a22168, 0x27c234c4
    r = fmaf(-n, 7.5497894159e-08f, r);
    r = fmaf(-n, 5.3903029534e-15f, r);
    const float r2 = r * r;
    // sin(r) = r + r^3 * S(r^2),  cos(r) = 1 - r^2/2 + r^4 * C(r^2)   (Cephes sinf/cosf kernels)
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2,
                    fmaf(-0.5f, r2, 1.0f));
    const int q = (int)n;
    const float s_ = (q & 1) ? pc : ps;
    const float c_ = (q & 1) ? ps : pc;
    *sn = (q & 2) ? -s_ : s_;
    *cs = ((q + 1) & 2) ? -c_ : c_;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- Philox4x32-10 (Random123), counter-based noise contract ---------------------------
// key = (seed_lo, seed_hi); counter = (quad_lo, quad_hi, draw_id, step); element e of a draw
// uses quad = e >> 2 and word e & 3.  Normals: Box-Muller on words (0,1) and (2,3).
struct Philox4 {
    uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01_open_low(uint32_t x) { return ((float)(x >> 8) + 1.0f) * 5.9604644775390625e-8f; }  // (0,1]
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }                    // [0,1)

// the 4 normals of quad q of draw (step, draw_id)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint32_t step, uint32_t draw, uint64_t quad, float out[4]) {
    Philox4 r = philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), draw, step, (uint32_t)seed, (uint32_t)(seed >> 32));
    float r0 = sqrtf(-2.0f * logf(u01_open_low(r.x)));
    float t0 = 6.2831853071795864769f * u01(r.y);
    float r1 = sqrtf(-2.0f * logf(u01_open_low(r.z)));
    float t1 = 6.2831853071795864769f * u01(r.w);
    out[0] = r0 * cosf(t0);
    out[1] = r0 * sinf(t0);
    out[2] = r1 * cosf(t1);
    out[3] = r1 * sinf(t1);
}
__device__ __forceinline__ float philox_normal1(uint64_t seed, uint32_t step, uint32_t draw, uint64_t elem) {
    float z[4];
    philox_normal4(seed, step, draw, elem >> 2, z);
    return z[elem & 3];
}
__device__ __forceinline__ float philox_uniform1(uint64_t seed, uint32_t step, uint32_t draw, uint64_t elem) {
    uint64_t quad = elem >> 2;
    Philox4 r = philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), draw, step, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t w = (elem & 3) == 0 ? r.x : (elem & 3) == 1 ? r.y : (elem & 3) == 2 ? r.z : r.w;
    return u01(w);
}

// draw ids (DESIGN.md "RNG"; mirrored in oracle/diffcsp_oracle.py for the checker)
enum : uint32_t {
    DRAW_X_T = 0, DRAW_L_T = 1, DRAW_T_T = 2,
    DRAW_CORR_X = 3, DRAW_PRED_L = 4, DRAW_PRED_T = 5, DRAW_PRED_X = 6,
    DRAW_FT_L = 7, DRAW_FT_X = 8, DRAW_FT_T = 9,
};

}  // namespace mi

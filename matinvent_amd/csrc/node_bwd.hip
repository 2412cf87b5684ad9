// The node-level BACKWARD chain between two edge stages of the fine-tune backward pass as ONE launch per layer boundary (round 6).
//
// Reference: loss.backward() through CSPLayer.node_model / the LayerNorm / the h_i, h_j columns of edge_mlp.0 (models/diffcsp/cspnet.py:59-91),
// driven by MatInvent.ft_step (pipeline/mat_invent.py:150-177).  Per layer boundary the seven-launch form (backward.hip) ran
//     gemm_nt(dPQ Whh) + residual -> layernorm_bwd -> [next layer down] silu_bwd -> silu_fwd -> gemm_nt(dY Wn2) -> silu_bwd -> gemm_nt(dXa Wn0)
// -- three latency-bound 64 x 64-tile products of 0.67 GFLOP each (23 us apiece at 1 280 rows, 11 % of a micro-step), three streaming passes and
// the LayerNorm gradient, every intermediate through HBM.  Here a workgroup owns 32 node rows (like node_chain_kernel, the forward's mirror image)
// and runs, with the intermediates in LDS / registers:
//   phase B (layer l, behind its edge stage):  d hn = d cat[:, :H] + dPQ Whh     (K = 2H: two accumulating passes over the halves of dPQ)
//                                              d h += LayerNorm'(d hn);  partial sums of d ln_w, d ln_b per workgroup
//   phase A (layer l - 1, in front of ITS edge stage):  dY = d h * silu'(Ypre)           -> HBM (operand of node_mlp.2's weight gradient)
//                                              dXa = (dY Wn2) * silu'(Xpre), Xa = silu(Xpre)  -> HBM (operands of the weight gradients)
//                                              d cat = dXa Wn0 (2H columns, two passes)  -> HBM (+ max |d cat| for the dZ2 plane scale)
// Arithmetic: every product runs on the fp16 matrix pipe from two fp16 planes per operand (three MFMA terms, f32 accumulate), the weights from
// fragment-order packs of the TRANSPOSED matrices (scale 2^6, built per parameter update), the gradient operand split on the way into LDS with a
// power-of-two scale taken from the EXACT absmax of the workgroup's own 32-row tile -- no a-priori bound is needed, saturation is impossible by
// construction, and because a tile belongs to one workgroup the scale needs no cross-workgroup exchange (deterministic).  The seven-launch form
// split its fp32 operands into three bf16 planes on the fly (six terms).  Both are fp32-class; the gradient tests run both (mi_debug_set_node_bwd).
#include <mutex>

#include "gemm_split.h"
#include "net.h"

namespace mi {

int g_node_bwd = 1;              // 1 (default): the fused backward chain for batches of at least g_node_bwd_min_blocks row blocks; 0: the seven-launch form
int g_node_bwd_min_blocks = 8;   // (a chain of fewer workgroups streams 5 MB of weights per layer through a handful of CUs: the split-K products win)

#if MI_PLANES_FP16

// fragment-order pack (see pack_frag_kernel, node_chain.hip) of the TRANSPOSE of W[K][rows] (row stride ld): out-column r, k index k  <-  W[k][r]
__global__ void pack_frag_t_kernel(const float* __restrict__ W, int ld, int rows, int K, u16* __restrict__ dst) {
    const int KS = K / 16;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (out-column, 8-k chunk) per thread; consecutive threads = consecutive out-columns (coalesced reads)
    if (idx >= (int64_t)rows * (K / 8)) return;
    const int r = (int)(idx % rows), ch = (int)(idx / rows);
    const int ct = r >> 5, l31 = r & 31, ks = ch >> 1, kg = ch & 1;
    u32x4 pk[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned p[3];
        pl_split_pair(W[(size_t)(ch * 8 + 2 * i) * ld + r], W[(size_t)(ch * 8 + 2 * i + 1) * ld + r], PL_SW, p);
        pk[0][i] = p[0];
        pk[1][i] = p[1];
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(dst + ((((size_t)ct * KS + ks) * 2 + pl) * 64 + kg * 32 + l31) * 8) = pk[pl];
}

struct NodeBwdArgs {
    int N = 0;
    // ---- phase B of layer l (dPQ == nullptr: skipped -- the first launch of a backward pass starts from d h as the final LayerNorm's gradient left it)
    const float* dPQ = nullptr;     // [N][2H] d (P_i | P_j): the edge stage's gradient with respect to the projections of LayerNorm(h)
    const float* dcat = nullptr;    // [N][2H] columns [0, H): d LayerNorm(h) through node_mlp.0 (written by phase A of the PREVIOUS launch)
    const u16* WhhA = nullptr;      // fragment packs (H x H): transpose of edge_mlp.0.weight[:, 0:H] and of [:, H:2H]
    const u16* WhhB = nullptr;
    const float* h = nullptr;       // [N][H] the layer's input (LayerNorm's x)
    const float* lnstat = nullptr;  // [N][2] {mean, 1 / sqrt(var + eps)} kept by the training forward
    const float* ln_w = nullptr;
    float* dh = nullptr;            // [N][H] the residual stream's gradient, updated in place
    float* lnpart = nullptr;        // [workgroups][2H] partial sums of d ln_w | d ln_b (part_reduce_kernel adds them into the gradient)
    // ---- phase A of the layer below (Wn2 == nullptr: skipped -- the last launch: below layer 0 is the embedding)
    const float* Ypre = nullptr;    // [N][H] node_mlp.2's pre-activation
    const float* Xpre = nullptr;    // [N][H] node_mlp.0's pre-activation
    const u16* Wn2 = nullptr;       // fragment pack (H x H): transpose of node_mlp.2.weight
    const u16* Wn0 = nullptr;       // fragment pack (2H x H): transpose of node_mlp.0.weight
    float* dY = nullptr;            // [N][H] out
    float* dXa = nullptr;           // [N][H] out
    float* Xa = nullptr;            // [N][H] out: silu(Xpre)
    float* dcat_out = nullptr;      // [N][2H] out (may alias dcat: a workgroup reads its rows of the old one before it writes the new one)
    unsigned* dcat_absmax = nullptr;   // atomicMax of the bit pattern of max |d cat| (zero before the backward pass)
};

template <int H>
struct NodeBwdCfg {
    static constexpr int KS = H / 16, ROWB = 2 * H + 16, PLB = 32 * ROWB, HLD = H + 4;
    static constexpr int LDS = 2 * PLB + 32 * HLD * 4 + 128;
};

template <int H, int NW, int D>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void node_bwd_kernel(NodeBwdArgs a) {
    using C = NodeBwdCfg<H>;
    constexpr int KS = C::KS, ROWB = C::ROWB, PLB = C::PLB, HLD = C::HLD, CW = H / NW, TW = CW / 32, RPW = 32 / NW;
    static_assert(KS % D == 0 && KS >= 2 * D && CW % 32 == 0 && D % 2 == 0, "ring depth / column split");
    static_assert(RPW * HLD >= 2 * H, "a wave's LayerNorm partial sums live in its own rows of Hs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* P = smem;                                          // gradient planes [2][32][ROWB]
    float* Hs = reinterpret_cast<float*>(smem + 2 * PLB);             // d hn rows [32][HLD]
    unsigned* red = reinterpret_cast<unsigned*>(smem + 2 * PLB + 32 * HLD * 4);   // [3][NW] wave maxima (one row per use: no reuse hazards)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int row0 = blockIdx.x * 32, N = a.N;
    const int c0 = lane * 8;
    const bool act = c0 < H;
    const bool phaseB = a.dPQ != nullptr, phaseA = a.Wn2 != nullptr;
    const int wsz = H * H * 4;   // bytes of one packed H x H operand

    u32x4 ring[D][TW][2];
    f32x16 acc[TW];
    int voff[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) voff[t] = lane * 16 + t * KS * 2048;
    auto ring_load = [&](const __amdgpu_buffer_rsrc_t& rs, int ct0, int ks, u32x4 (&w)[TW][2]) {
        const int so = (ct0 * KS + ks) * 2048;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w[t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[t] + pl * 1024, so, 0);
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rs, int ct0) {
#pragma unroll
        for (int d = 0; d < D; ++d) ring_load(rs, ct0, d, ring[d]);
    };
    auto read_act = [&](int ks, f16x8 (&af)[2]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const f16x8*>(P + pl * PLB + l31 * ROWB + (2 * ks + kg) * 16);
    };
    // term order of the plane GEMMs: (a1, b0), (a0, b1), (a0, b0) with a = gradient operand, b = weight
    auto mma_step = [&](auto tr, const u32x4 (&w)[TW][2], const f16x8 (&af)[2]) {
        constexpr bool TR = decltype(tr)::value;
#pragma unroll
        for (int term = MI_TERM0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const f16x8 wv = __builtin_bit_cast(f16x8, w[t][term == 1 ? 1 : 0]);
                const f16x8 av = af[term == 0 ? 1 : 0];
                if constexpr (TR) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, av, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wv, acc[t], 0, 0, 0);
            }
    };
    // one product pass over K = H; the ring holds k-steps 0 .. D-1 on entry, nothing on exit.  zero: start a new accumulation
    auto run = [&](auto tr, auto zero, const __amdgpu_buffer_rsrc_t& rs, int ct0) {
        if constexpr (decltype(zero)::value) {
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
        f16x8 af[2][2];
        read_act(0, af[0]);
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS - D; ks0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                read_act(ks0 + d + 1, af[(d + 1) & 1]);
                mma_step(tr, ring[d], af[d & 1]);
                ring_load(rs, ct0, ks0 + d + D, ring[d]);
                __builtin_amdgcn_sched_barrier(0);   // (keeps every refill where it is: node_chain.hip)
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d + 1 < D) read_act(KS - D + d + 1, af[(d + 1) & 1]);
            mma_step(tr, ring[d], af[d & 1]);
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    auto wave_max = [&](float m) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        return m;
    };
    // the tile's scale from the NW wave maxima of use `u` (every lane reads the same words: uniform)
    auto tile_scale = [&](int u) {
        unsigned bits = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) bits = max(bits, red[u * NW + w]);   // (bit patterns of non-negative floats order like the floats)
        return f16_scale_from_absmax(bits);
    };
    unsigned sat = 0;
    // eight consecutive values of row `row` (row layout: a lane owns columns c0 .. c0 + 7) -> the two planes
    auto put_row_planes = [&](int row, const f32x4& x, const f32x4& y, float s) {
        u32x4 pk[2];
        unsigned pr[3];
        pl_split_pair_acc(x[0], x[1], s, pr, sat); pk[0][0] = pr[0]; pk[1][0] = pr[1];
        pl_split_pair_acc(x[2], x[3], s, pr, sat); pk[0][1] = pr[0]; pk[1][1] = pr[1];
        pl_split_pair_acc(y[0], y[1], s, pr, sat); pk[0][2] = pr[0]; pk[1][2] = pr[1];
        pl_split_pair_acc(y[2], y[3], s, pr, sat); pk[0][3] = pr[0]; pk[1][3] = pr[1];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4*>(P + pl * PLB + row * ROWB + c0 * 2) = pk[pl];
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 g0[RPW], g1[RPW];   // the workgroup's rows of d h (row layout), then of dY
    if (phaseB) {
        const __amdgpu_buffer_rsrc_t rs_a = uniform_rsrc(a.WhhA, wsz), rs_b = uniform_rsrc(a.WhhB, wsz);
        // ---- B1: the tile's rows of dPQ (both halves), its absmax, the first half as planes ----
        f32x4 pa0[RPW], pa1[RPW], pb0[RPW], pb1[RPW];
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int i = row0 + wave * RPW + r;
            pa0[r] = pa1[r] = pb0[r] = pb1[r] = zero4;
            if (act && i < N) {
                const float* p = a.dPQ + (size_t)i * 2 * H + c0;
                pa0[r] = *reinterpret_cast<const f32x4*>(p);
                pa1[r] = *reinterpret_cast<const f32x4*>(p + 4);
                pb0[r] = *reinterpret_cast<const f32x4*>(p + H);
                pb1[r] = *reinterpret_cast<const f32x4*>(p + H + 4);
            }
        }
        ring_fill(rs_a, wave * TW);   // (behind the row loads: vector loads return in order, the arithmetic below waits for the rows only)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(pa0[r][k]), fabsf(pa1[r][k]))), fmaxf(fabsf(pb0[r][k]), fabsf(pb1[r][k])));
        m = wave_max(m);
        if (lane == 0) red[0 * NW + wave] = __float_as_uint(m);
        __syncthreads();
        const float s1 = tile_scale(0);
        if (act)
#pragma unroll
            for (int r = 0; r < RPW; ++r) put_row_planes(wave * RPW + r, pa0[r], pa1[r], s1);
        __syncthreads();
        // the epilogue's residual rows, requested in front of the product (they land under its MFMAs)
        f32x4 rq[TW][4];
        {
            const int i = row0 + l31, ic = i < N ? i : N - 1;
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) rq[t][q] = *reinterpret_cast<const f32x4*>(a.dcat + (size_t)ic * 2 * H + wave * CW + t * 32 + 8 * q + 4 * kg);
        }
        // ---- B2: d hn = dPQ[:, :H] WhhA^T + dPQ[:, H:] WhhB^T (transposed result: lane = row, registers = columns) ----
        run(Yes{}, Yes{}, rs_a, wave * TW);
        ring_fill(rs_b, wave * TW);
        __syncthreads();                // every wave has read the first half's planes
        if (act)
#pragma unroll
            for (int r = 0; r < RPW; ++r) put_row_planes(wave * RPW + r, pb0[r], pb1[r], s1);
        __syncthreads();
        run(Yes{}, No{}, rs_b, wave * TW);
        if (phaseA) ring_fill(uniform_rsrc(a.Wn2, wsz), wave * TW);   // phase A's first product, in flight under the LayerNorm gradient
        // ---- B3: + d cat[:, :H] -> Hs ----
        {
            const float os = 1.f / (s1 * PL_SW);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = wave * CW + t * 32 + 8 * q + 4 * kg;
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = acc[t][4 * q + k] * os + rq[t][q][k];
                    *reinterpret_cast<f32x4*>(Hs + l31 * HLD + c4) = o;
                }
        }
        __syncthreads();
        // ---- B4: LayerNorm gradient, a wave per row (layernorm_bwd_kernel's arithmetic): dx = rstd (g - mean(g) - xhat mean(g xhat)), g = d hn * w ----
        f32x4 w0 = zero4, w1 = zero4;
        if (act) {
            w0 = *reinterpret_cast<const f32x4*>(a.ln_w + c0);
            w1 = *reinterpret_cast<const f32x4*>(a.ln_w + c0 + 4);
        }
        f32x4 dw0 = zero4, dw1 = zero4, db0 = zero4, db1 = zero4;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r, i = row0 + row;
            const bool live = act && i < N;
            f32x4 d0 = zero4, d1 = zero4, x0 = zero4, x1 = zero4, o0 = zero4, o1 = zero4;
            float mean = 0.f, rstd = 0.f;
            if (live) {
                d0 = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c0);
                d1 = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c0 + 4);
                x0 = *reinterpret_cast<const f32x4*>(a.h + (size_t)i * H + c0);
                x1 = *reinterpret_cast<const f32x4*>(a.h + (size_t)i * H + c0 + 4);
                o0 = *reinterpret_cast<const f32x4*>(a.dh + (size_t)i * H + c0);
                o1 = *reinterpret_cast<const f32x4*>(a.dh + (size_t)i * H + c0 + 4);
            }
            if (i < N) {
                mean = a.lnstat[2 * (size_t)i];
                rstd = a.lnstat[2 * (size_t)i + 1];
            }
            f32x4 xh0, xh1, gg0, gg1;
            float s1g = 0.f, s2g = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xh0[k] = live ? (x0[k] - mean) * rstd : 0.f;
                xh1[k] = live ? (x1[k] - mean) * rstd : 0.f;
                gg0[k] = d0[k] * w0[k];
                gg1[k] = d1[k] * w1[k];
                s1g += gg0[k] + gg1[k];
                s2g += gg0[k] * xh0[k] + gg1[k] * xh1[k];
                dw0[k] += d0[k] * xh0[k];
                dw1[k] += d1[k] * xh1[k];
                db0[k] += d0[k];
                db1[k] += d1[k];
            }
            s1g = wave_sum(s1g) / (float)H;
            s2g = wave_sum(s2g) / (float)H;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o0[k] += rstd * (gg0[k] - s1g - xh0[k] * s2g);
                o1[k] += rstd * (gg1[k] - s1g - xh1[k] * s2g);
            }
            if (live) {
                *reinterpret_cast<f32x4*>(a.dh + (size_t)i * H + c0) = o0;
                *reinterpret_cast<f32x4*>(a.dh + (size_t)i * H + c0 + 4) = o1;
            }
            g0[r] = live ? o0 : zero4;
            g1[r] = live ? o1 : zero4;
        }
        // the wave's partial sums of d ln_w | d ln_b over its rows -> its OWN (consumed) rows of Hs; summed over the waves behind the next barrier
        if (act) {
            float* my = Hs + wave * RPW * HLD;
            *reinterpret_cast<f32x4*>(my + c0) = dw0;
            *reinterpret_cast<f32x4*>(my + c0 + 4) = dw1;
            *reinterpret_cast<f32x4*>(my + H + c0) = db0;
            *reinterpret_cast<f32x4*>(my + H + c0 + 4) = db1;
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int i = row0 + wave * RPW + r;
            g0[r] = g1[r] = zero4;
            if (act && i < N) {
                g0[r] = *reinterpret_cast<const f32x4*>(a.dh + (size_t)i * H + c0);
                g1[r] = *reinterpret_cast<const f32x4*>(a.dh + (size_t)i * H + c0 + 4);
            }
        }
        if (phaseA) ring_fill(uniform_rsrc(a.Wn2, wsz), wave * TW);
    }
    auto ln_partials_out = [&]() {   // (behind a barrier that follows the partial sums' stores)
        for (int c = tid; c < 2 * H; c += 64 * NW) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += Hs[w * RPW * HLD + c];
            a.lnpart[(size_t)blockIdx.x * 2 * H + c] = s;
        }
    };
    if (!phaseA) {
        if (phaseB) {
            __syncthreads();
            ln_partials_out();
        }
        sat_report(sat);
        return;
    }
    // ---- A1: dY = d h * silu'(Ypre) -> HBM and, as planes, LDS ----
    {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int i = row0 + wave * RPW + r;
            if (act && i < N) {
                const f32x4 y0 = *reinterpret_cast<const f32x4*>(a.Ypre + (size_t)i * H + c0);
                const f32x4 y1 = *reinterpret_cast<const f32x4*>(a.Ypre + (size_t)i * H + c0 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    g0[r][k] *= silu_grad_fast(y0[k]);
                    g1[r][k] *= silu_grad_fast(y1[k]);
                    m = fmaxf(m, fmaxf(fabsf(g0[r][k]), fabsf(g1[r][k])));
                }
                *reinterpret_cast<f32x4*>(a.dY + (size_t)i * H + c0) = g0[r];
                *reinterpret_cast<f32x4*>(a.dY + (size_t)i * H + c0 + 4) = g1[r];
            }
        }
        m = wave_max(m);
        if (lane == 0) red[1 * NW + wave] = __float_as_uint(m);
    }
    __syncthreads();   // (also: every wave is through its reads of Hs and of the planes of phase B)
    if (phaseB) ln_partials_out();
    const float s2 = tile_scale(1);
    if (act)
#pragma unroll
        for (int r = 0; r < RPW; ++r) put_row_planes(wave * RPW + r, g0[r], g1[r], s2);
    __syncthreads();
    // node_mlp.0's pre-activation in the product's result layout, requested in front of it
    f32x4 xq[TW][4];
    {
        const int i = row0 + l31, ic = i < N ? i : N - 1;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) xq[t][q] = *reinterpret_cast<const f32x4*>(a.Xpre + (size_t)ic * H + wave * CW + t * 32 + 8 * q + 4 * kg);
    }
    // ---- A2: dY Wn2 (transposed result) ----
    const __amdgpu_buffer_rsrc_t rs_n2 = uniform_rsrc(a.Wn2, wsz);
    run(Yes{}, Yes{}, rs_n2, wave * TW);
    const __amdgpu_buffer_rsrc_t rs_n0 = uniform_rsrc(a.Wn0, 2 * wsz);
    ring_fill(rs_n0, wave * TW);   // pass 0 of the last product, in flight under the epilogue
    // ---- A3: dXa = (.) * silu'(Xpre), Xa = silu(Xpre) -> HBM; dXa -> planes ----
    {
        const float os = 1.f / (s2 * PL_SW);
        const int i = row0 + l31;
        float m = 0.f;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = wave * CW + t * 32 + 8 * q + 4 * kg;
                f32x4 dx, xa;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xp = xq[t][q][k];
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xp * -1.44269504088896340736f));
                    xa[k] = xp * sg;
                    dx[k] = i < N ? acc[t][4 * q + k] * os * (sg * (1.0f + xp * (1.0f - sg))) : 0.f;
                    acc[t][4 * q + k] = dx[k];
                    m = fmaxf(m, fabsf(dx[k]));
                }
                if (i < N) {
                    *reinterpret_cast<f32x4*>(a.dXa + (size_t)i * H + c4) = dx;
                    *reinterpret_cast<f32x4*>(a.Xa + (size_t)i * H + c4) = xa;
                }
            }
        m = wave_max(m);
        if (lane == 0) red[2 * NW + wave] = __float_as_uint(m);
    }
    __syncthreads();   // (also: every wave has read the dY planes)
    const float s3 = tile_scale(2);
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c4 = wave * CW + t * 32 + 8 * q + 4 * kg;
            unsigned p01[3], p23[3];
            pl_split_pair_acc(acc[t][4 * q], acc[t][4 * q + 1], s3, p01, sat);
            pl_split_pair_acc(acc[t][4 * q + 2], acc[t][4 * q + 3], s3, p23, sat);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<uint2*>(P + pl * PLB + l31 * ROWB + c4 * 2) = make_uint2(p01[pl], p23[pl]);
        }
    __syncthreads();
    // ---- A4: d cat = dXa Wn0 (2H columns, two passes; lane = column, registers = rows: coalesced stores) ----
    {
        const float os = 1.f / (s3 * PL_SW);
        float m = 0.f;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int ct0 = (pass * H + wave * CW) / 32;
            run(No{}, Yes{}, rs_n0, ct0);
            if (pass == 0) ring_fill(rs_n0, (H + wave * CW) / 32);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const int col = pass * H + wave * CW + t * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    const float v = acc[t][r] * os;
                    if (i < N) {
                        a.dcat_out[(size_t)i * 2 * H + col] = v;
                        m = fmaxf(m, fabsf(v));
                    }
                }
            }
        }
        if (a.dcat_absmax) {
            m = wave_max(m);
            if (lane == 0) atomicMax(a.dcat_absmax, __float_as_uint(m));
        }
    }
    sat_report(sat);
}

template <int H, int NW, int D>
static int node_bwd_launch(const NodeBwdArgs& a, hipStream_t s) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)node_bwd_kernel<H, NW, D>, hipFuncAttributeMaxDynamicSharedMemorySize, NodeBwdCfg<H>::LDS); });
    MI_HIP(attr_err);
    hipLaunchKernelGGL((node_bwd_kernel<H, NW, D>), dim3(cdiv(a.N, 32)), dim3(64 * NW), NodeBwdCfg<H>::LDS, s, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

size_t node_bwd_pack_elems(int H) { return (size_t)5 * H * H * 2; }   // per layer: [WhhA | WhhB | Wn2^T | Wn0^T (2H rows)] x two planes (u16 elements)

bool node_bwd_supported(const mi_net* net, const mi_batch* b) {
    return g_node_bwd && net->cfg.ln && (net->H == 128 || net->H == 256 || net->H == 512) && net->Wbw != nullptr && g_gemm_mode == MI_GEMM_SPLIT &&
           cdiv(b->N, 32) >= g_node_bwd_min_blocks;
}

// packs of layer l (net_pack_transposes, per parameter update): W1 = edge_mlp.0.weight (row stride edge_in), Wn0 = node_mlp.0.weight [H][2H], Wn2 = node_mlp.2.weight [H][H]
int node_bwd_pack(mi_net* net, int l, const float* W1, const float* Wn0, const float* Wn2, hipStream_t s) {
    const int H = net->H;
    u16* base = net->Wbw + (size_t)l * node_bwd_pack_elems(H);
    const size_t one = (size_t)H * H * 2;
    const int nb = cdiv((int64_t)H * (H / 8), 256);
    hipLaunchKernelGGL(pack_frag_t_kernel, dim3(nb), dim3(256), 0, s, W1, net->edge_in, H, H, base);
    hipLaunchKernelGGL(pack_frag_t_kernel, dim3(nb), dim3(256), 0, s, W1 + H, net->edge_in, H, H, base + one);
    hipLaunchKernelGGL(pack_frag_t_kernel, dim3(nb), dim3(256), 0, s, Wn2, H, H, H, base + 2 * one);
    hipLaunchKernelGGL(pack_frag_t_kernel, dim3(2 * nb), dim3(256), 0, s, Wn0, 2 * H, 2 * H, H, base + 3 * one);
    MI_KERNEL_CHECK();
    return MI_OK;
}

// The launch at the boundary below layer `l` (l = L: the first launch of a backward pass, phase A of layer L - 1 only; l = 0: phase B of layer 0 only).
//   dPQ / cat-side operands of layer l, the window-slot pointers of layer l - 1 (dY, dXa, Xa) and the LayerNorm partial sums' scratch come from the caller.
int node_bwd(mi_net* net, mi_batch* b, int l, const float* dPQ, float* dh, float* dY, float* dXa, float* Xa, float* lnpart, unsigned* dcat_absmax, hipStream_t s) {
    const int H = net->H, L = net->L, N = b->N;
    Tape& t = b->tape;
    const size_t NH = (size_t)N * H, one = (size_t)H * H * 2;
    NodeBwdArgs a;
    a.N = N;
    a.dh = dh;
    if (l < L) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        const u16* base = net->Wbw + (size_t)l * node_bwd_pack_elems(H);
        a.dPQ = dPQ;
        a.dcat = t.dcat;
        a.WhhA = base;
        a.WhhB = base + one;
        a.h = b->h + (size_t)l * NH;
        a.lnstat = t.lnstat + (size_t)l * N * 2;
        a.ln_w = net->p(p + "layer_norm.weight");
        a.lnpart = lnpart;
        count_mfma(N, H, 2 * H, MI_PLANES_TERMS);
    }
    if (l > 0) {
        const u16* base = net->Wbw + (size_t)(l - 1) * node_bwd_pack_elems(H);
        a.Ypre = t.Ypre + (size_t)(l - 1) * NH;
        a.Xpre = t.Xpre + (size_t)(l - 1) * NH;
        a.Wn2 = base + 2 * one;
        a.Wn0 = base + 3 * one;
        a.dY = dY;
        a.dXa = dXa;
        a.Xa = Xa;
        a.dcat_out = t.dcat;
        a.dcat_absmax = dcat_absmax;
        count_mfma(N, 3 * H, H, MI_PLANES_TERMS);
    }
    if (H == 512) return node_bwd_launch<512, 8, 4>(a, s);
    if (H == 256) return node_bwd_launch<256, 8, 4>(a, s);
    return node_bwd_launch<128, 4, 4>(a, s);
}

#else
size_t node_bwd_pack_elems(int) { return 0; }
bool node_bwd_supported(const mi_net*, const mi_batch*) { return false; }
int node_bwd_pack(mi_net*, int, const float*, const float*, const float*, hipStream_t) { return MI_OK; }
int node_bwd(mi_net*, mi_batch*, int, const float*, float*, float*, float*, float*, float*, unsigned*, hipStream_t) { return MI_ESTATE; }
#endif

}  // namespace mi

extern "C" int mi_debug_set_node_bwd(int on, int min_blocks) {
    const int prev = mi::g_node_bwd;
    mi::g_node_bwd = on ? 1 : 0;
    if (min_blocks > 0) mi::g_node_bwd_min_blocks = min_blocks;
    return prev;
}

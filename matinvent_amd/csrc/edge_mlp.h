// Fused edge-message kernel of one CSPLayer (forward):
//
//   reference (models/diffcsp/cspnet.py:59-79):
//     e_in  = cat[hn_i, hn_j, (L L^T)_b, ff(x_j - x_i)]            [E, 2H+9+6F]
//     m     = SiLU(W2 SiLU(W1 e_in + b1) + b2)                      [E, H]
//     agg_i = mean_j m_(i,j)
//
//   here: W1 e_in + b1 = P_i[i] + P_j[j] + G[b] + Wff ff(d)   with  P_i = hn W1[:, :H]^T,
//   P_j = hn W1[:, H:2H]^T (node-level GEMMs, 1/n of the per-edge cost), G = gram W1g^T + b1.
//   Only the Fourier block (K = 6F) and the second linear (K = H) are per-edge work.
//
// One wave owns a tile of 32 consecutive edges and ALL H output features, computed TRANSPOSED
// on v_mfma_f32_32x32x2_f32:  Z1^T[f, e] = sum_k Wff[f, k] ff[e, k].  The C/D fragment of that
// product (lane = edge, registers = features) is exactly the B fragment the next product
// Z2^T = W2 M1^T needs once its k-order is permuted to match, so M1 never leaves the register
// file: no LDS round trip, no [E, H] intermediate in HBM.  The Fourier features are generated
// in registers (bounded-range sincos) under the MFMAs.  Weights are pre-packed (cspnet.hip
// pack_* kernels) so that every A-operand load is one contiguous 1 KiB wave read; waves are
// independent (no barriers) and, being alone on their SIMD (512 registers), hide memory latency
// by explicit software pipelining of the weight stream.
//
// Register plan (H = 512): hipcc places every MFMA accumulator in the 256 AGPRs, so GEMM1 runs
// in two halves of 8 feature tiles (128 accumulators each; the Fourier operand is simply
// regenerated, it is free under the MFMAs).  After bias-gather + SiLU the halves are ordinary
// VALU results, which leaves the accumulator file to GEMM2's output tiles.
//
// Output: per-node partial sums over each node's edge run inside this tile, written to
// part[slot][node][:] with slot = tile - first tile of that node (deterministic; no float
// atomics).  finalize_agg (cspnet.hip) adds the slots and divides by the degree.
#pragma once
#include <type_traits>

#include "common.h"

namespace mi {

struct EdgeFwdArgs {
    const float* PQ;        // [N, 2H]  cols [0,H) = P_i, cols [H,2H) = P_j
    const float* G;         // [B, H]   gram term + b1
    const int* src;         // [E] row node i (sorted ascending)
    const int* dst;         // [E] col node j
    const int* node2graph;  // [N]
    const int* rowptr;      // [N+1] first edge of every node
    const float* Wff_p;     // packed [KP/4][NT][64][4], pair s = c*FP + k
    const float* FFp;       // [tiles][KP/4][64][4] Fourier operand in B-fragment order (built once per evaluation)
    const float* W2_p;      // packed [NT(u)][NT(t)][4(q)][64][4]
    const float* b2;        // [H]
    float* part;            // [nslots][N][H]
    float* Z1;              // SAVE only: [E, H] pre-activation of linear 1 (for backward)
    float* Z2;              // SAVE only: [E, H] pre-activation of linear 2
    unsigned long long* dbg;  // MI_TIMING builds only: [tiles][16] s_memtime stamps
    int64_t E;
    int N, F, KP;           // KP = 3*FP (sin,cos) pairs, FP = F rounded up to a multiple of 8
};

// Pin program order: the asm memory clobber stops IR-level load motion (sched_barrier is IntrNoMem,
// loads float across it), the sched_barrier stops the machine scheduler.
#define MI_PIN()                          \
    do {                                  \
        asm volatile("" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

#ifdef MI_TIMING
#define MI_STAMP(k) do { if (lane == 0) a.dbg[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define MI_STAMP(k) do { } while (0)
#endif

#ifndef MI_UG
#define MI_UG 2
#endif
#ifndef MI_RING
#define MI_RING 4
#endif

template <int H, bool SAVE>
__global__ __launch_bounds__(64, 1) void edge_mlp_fwd_kernel(EdgeFwdArgs a) {
    constexpr int NT = H / 32, NTH = NT / 2;
    constexpr int UG = NT < MI_UG ? NT : MI_UG;  // output tiles in flight in GEMM2
    constexpr int RING = MI_RING, NSTEP = NT * 4;
    __shared__ __attribute__((aligned(16))) float tr[32 * 36];

    const int lane = threadIdx.x, e_l = lane & 31, hi = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * 32;
    const int nvalid = (int)((a.E - e0) < 32 ? (a.E - e0) : 32);
    const int64_t e = e0 + (e_l < nvalid ? e_l : nvalid - 1);
    const int i = a.src[e], j = a.dst[e];
    const int g = a.node2graph[i];

    const float* pi = a.PQ + (size_t)i * (2 * H) + 4 * hi;
    const float* pj = a.PQ + (size_t)j * (2 * H) + H + 4 * hi;
    const float* pg = a.G + (size_t)g * H + 4 * hi;
    float* z1p = SAVE ? a.Z1 + (size_t)e * H + 4 * hi : nullptr;
    const bool row_ok = e_l < nvalid;

    // ---- GEMM1 half: Z1^T[32*(T0+t) .. ] = Wff * ff^T for NTH feature tiles --------------------
    // K runs over (coordinate c, frequency k) pairs, FP = F rounded up to 8 per coordinate (zero
    // weights on the pads), four pairs per packed float4 -> 2*4 k-values per m-step.  Software
    // pipeline, all in registers: weights AND the Fourier operand of step m+1 are loaded under the
    // MFMAs of step m.  (An earlier version generated sin/cos in the loop: on gfx950 VALU work does
    // not overlap f32 MFMAs, so the 12x per-evaluation recomputation cost 13 % of the kernel.)
    auto gemm1_half = [&](auto t0_tag, f32x16 (&acc)[NTH]) {
        constexpr int T0 = decltype(t0_tag)::value;
#pragma unroll
        for (int t = 0; t < NTH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wff_p) + (size_t)T0 * 64 + lane;
        // Fourier operand: precomputed once per network evaluation in fragment order (fourier_pack_kernel,
        // cspnet.hip): FFp[tile][m][lane] = float4 of the four pairs of m-step m for lane (edge, hi).
        const f32x4* fp = reinterpret_cast<const f32x4*>(a.FFp) + ((size_t)blockIdx.x * (a.KP / 4)) * 64 + lane;
        const int nm = a.KP / 4;  // even
        auto mfma_block = [&](const f32x4 (&w)[NTH], const f32x4& bv) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < NTH; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t][q], bv[q], acc[t], 0, 0, 0);
        };
        f32x4 wa[NTH], wb[NTH], bva, bvb;
#pragma unroll
        for (int t = 0; t < NTH; ++t) wa[t] = wp[(size_t)t * 64];
        bva = fp[0];
        for (int m = 0; m < nm; m += 2) {
            MI_PIN();
#pragma unroll
            for (int t = 0; t < NTH; ++t) wb[t] = wp[((size_t)(m + 1) * NT + t) * 64];
            bvb = fp[(size_t)(m + 1) * 64];
            MI_PIN();
            mfma_block(wa, bva);
            MI_PIN();
            const int mn = (m + 2 < nm) ? m + 2 : m;  // last step: harmless reload
#pragma unroll
            for (int t = 0; t < NTH; ++t) wa[t] = wp[((size_t)mn * NT + t) * 64];
            bva = fp[(size_t)mn * 64];
            MI_PIN();
            mfma_block(wb, bvb);
        }
        MI_PIN();
    };

    // lane (e, hi), register r of feature tile t  <->  feature 32t + 8(r>>2) + 4hi + (r&3)
    struct Gath { f32x4 x, y, z; };
    auto gather = [&](int qq) {  // quarter qq = 4*t + q  -> features 32t + 8q + 4hi .. +3
        const int off = 32 * (qq >> 2) + 8 * (qq & 3);
        return Gath{*reinterpret_cast<const f32x4*>(pi + off), *reinterpret_cast<const f32x4*>(pj + off),
                    *reinterpret_cast<const f32x4*>(pg + off)};
    };
    // pre-activation = Fourier part + node terms -> (save) -> SiLU, for quarter q of global tile t
    auto finish_quarter = [&](f32x16& tile, int t, int q, const Gath& gq) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = tile[4 * q + c] + ((gq.x[c] + gq.y[c]) + gq.z[c]);
        if constexpr (SAVE) {
            if (row_ok) *reinterpret_cast<f32x4*>(z1p + 32 * t + 8 * q) = v;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) tile[4 * q + c] = silu_fast(v[c]);
    };

    f32x16 m1a[NTH], m1b[NTH];
    MI_STAMP(0);
    gemm1_half(std::integral_constant<int, 0>{}, m1a);
    MI_STAMP(1);
    // first half: node terms + SiLU, four quarters (12 float4 gathers) in flight at a time
#pragma unroll
    for (int t = 0; t < NTH; ++t) {
        Gath gq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] = gather(4 * t + q);
#pragma unroll
        for (int q = 0; q < 4; ++q) finish_quarter(m1a[t], t, q, gq[q]);
        asm volatile("" : "+v"(m1a[t]));  // materialise here: keeps the SiLU block from sinking past GEMM1's second half
        MI_PIN();
    }
    MI_STAMP(2);
    gemm1_half(std::integral_constant<int, NTH>{}, m1b);
    MI_STAMP(3);

    // ---- segment structure of this tile (runs of equal src) -------------------------------
    const int i_prev = __shfl_up(i, 1, 64);
    const bool is_start = (hi == 0) && row_ok && (e_l == 0 || i != i_prev);
    const uint32_t starts = (uint32_t)__ballot(is_start);
    const int tile = (int)blockIdx.x;

    // ---- GEMM2: Z2^T = W2 * M1^T + b2, then SiLU and the per-node partial sums ------------
    // Weight stream through a RING-deep register ring (one slot = the UG float4 of one (t,q)
    // step = 4*UG MFMAs).  During the FIRST pass the second half of M1 is still raw: its
    // quarters are finished (gather-add + SiLU) under the MFMAs that consume the first half.
    const f32x4* w2p = reinterpret_cast<const f32x4*>(a.W2_p) + lane;
    auto pass = [&](int ug, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        f32x16 o[UG];
#pragma unroll
        for (int uu = 0; uu < UG; ++uu)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b = *reinterpret_cast<const f32x4*>(a.b2 + 32 * (ug + uu) + 8 * q + 4 * hi);
#pragma unroll
                for (int c = 0; c < 4; ++c) o[uu][4 * q + c] = b[c];
            }
        f32x4 ring[RING][UG];
        auto wload = [&](int step, int uu) { return w2p[(((size_t)(ug + uu) * NT) * 4 + step) * 64]; };
#pragma unroll
        for (int st = 0; st < RING; ++st)
#pragma unroll
            for (int uu = 0; uu < UG; ++uu) ring[st][uu] = wload(st, uu);
        Gath gr[2];  // gathers for second-half quarters st (finished this step) and st+1
        if constexpr (FIRST) {
            gr[0] = gather(4 * NTH + 0);
            gr[1] = gather(4 * NTH + 1);
        }
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int t = st >> 2, q = st & 3;
            MI_PIN();
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int uu = 0; uu < UG; ++uu) {
                    const float bop = t < NTH ? m1a[t < NTH ? t : 0][4 * q + c] : m1b[t < NTH ? 0 : t - NTH][4 * q + c];
                    o[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[st % RING][uu][c], bop, o[uu], 0, 0, 0);
                }
            if constexpr (FIRST) {
                if (st < 4 * NTH) {  // second-half quarter st: tile NTH + st/4, under these MFMAs
                    finish_quarter(m1b[st >> 2], NTH + (st >> 2), st & 3, gr[st & 1]);
                    if (st + 2 < 4 * NTH) gr[st & 1] = gather(4 * NTH + st + 2);
                }
            }
            if (st + RING < NSTEP) {
#pragma unroll
                for (int uu = 0; uu < UG; ++uu) ring[st % RING][uu] = wload(st + RING, uu);
            }
        }
        MI_PIN();
        if (ug == 0) MI_STAMP(4);
        if (ug == UG) MI_STAMP(6);

#pragma unroll
        for (int uu = 0; uu < UG; ++uu) {
            const int u = ug + uu;
            if constexpr (SAVE) {
                if (row_ok) {
                    float* z = a.Z2 + (size_t)e * H + 32 * u + 4 * hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {o[uu][4 * q], o[uu][4 * q + 1], o[uu][4 * q + 2], o[uu][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(z + 8 * q) = v;
                    }
                }
            }
            // transpose through LDS: tr[edge][feature]
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {silu_fast(o[uu][4 * q]), silu_fast(o[uu][4 * q + 1]), silu_fast(o[uu][4 * q + 2]), silu_fast(o[uu][4 * q + 3])};
                *reinterpret_cast<f32x4*>(&tr[e_l * 36 + 8 * q + 4 * hi]) = v;
            }
            __syncthreads();
            // Segmented sums over edges (runs of equal src), deterministic order.  Lane (f = e_l, hi)
            // pulls its 16 rows [16hi, 16hi+16) of column f in one pipelined burst of LDS reads, then
            // walks the (wave-uniform) segments: predicated adds over its rows in edge order, one
            // cross-half add, and the half that owns the segment's parity stores the partial sum.
            float col[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) col[r] = tr[(16 * hi + r) * 36 + e_l];
            uint32_t rem = starts;
            int seg = 0;
            while (rem) {
                const int s = __builtin_ctz(rem);
                rem &= rem - 1;
                const int end = rem ? __builtin_ctz(rem) : nvalid;
                const int node = __builtin_amdgcn_readlane(i, s);  // s is wave-uniform
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 16 * hi + r;
                    sum += (row >= s && row < end) ? col[r] : 0.f;
                }
                const float other = __shfl_xor(sum, 32, 64);
                sum = hi ? other + sum : sum + other;  // rows [0,16) first, then [16,32): edge order
                if ((seg & 1) == hi) {
                    const int slot = tile - (a.rowptr[node] >> 5);
                    a.part[((size_t)slot * a.N + node) * H + 32 * u + e_l] = sum;
                }
                ++seg;
            }
        }
    };
    pass(0, std::true_type{});
    MI_STAMP(5);
    for (int ug = UG; ug < NT; ug += UG) {
        pass(ug, std::false_type{});
        if (ug == UG) MI_STAMP(7);
    }
    MI_STAMP(8);
}

}  // namespace mi

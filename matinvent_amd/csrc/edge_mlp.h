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
// in registers (sincosf) under the MFMAs.  Weights are pre-packed (pack.h) so that every
// A-operand load is one contiguous 1 KiB wave read; waves are independent (no barriers).
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
    const float* frac;      // [N, 3]
    const int* src;         // [E] row node i (sorted ascending)
    const int* dst;         // [E] col node j
    const int* node2graph;  // [N]
    const int* rowptr;      // [N+1] first edge of every node
    const float* freqs;     // [F]  2*pi*k table
    const float* Wff_p;     // packed [KP/4][NT][64][4]
    const float* W2_p;      // packed [NT(u)][NT(t)][4(q)][64][4]
    const float* b2;        // [H]
    float* part;            // [nslots][N][H]
    float* Z1;              // optional [E, H] pre-activation of linear 1 (saved for backward)
    float* Z2;              // optional [E, H] pre-activation of linear 2
    int64_t E;
    int N, F, KP;           // KP = number of (sin,cos) pairs padded to a multiple of 4
};

template <int H>
__global__ __launch_bounds__(64, 1) void edge_mlp_fwd_kernel(EdgeFwdArgs a) {
    constexpr int NT = H / 32;
#ifndef MI_UG
#define MI_UG 4
#endif
#ifndef MI_RING
#define MI_RING 4
#endif
    constexpr int UG = NT < MI_UG ? NT : MI_UG;  // output tiles in flight in GEMM2
    __shared__ __attribute__((aligned(16))) float tr[32 * 36];

    const int lane = threadIdx.x, e_l = lane & 31, hi = lane >> 5;
    const int64_t e0 = (int64_t)blockIdx.x * 32;
    const int nvalid = (int)((a.E - e0) < 32 ? (a.E - e0) : 32);
    const int64_t e = e0 + (e_l < nvalid ? e_l : nvalid - 1);
    const int i = a.src[e], j = a.dst[e];
    const int g = a.node2graph[i];

    // fractional difference (x_j - x_i) % 1   (cspnet.py:242)
    float d0 = pymod1(a.frac[j * 3 + 0] - a.frac[i * 3 + 0]);
    float d1 = pymod1(a.frac[j * 3 + 1] - a.frac[i * 3 + 1]);
    float d2 = pymod1(a.frac[j * 3 + 2] - a.frac[i * 3 + 2]);

    // ---- accumulators start at P_i[i] + P_j[j] + G[g]  (C-in of the first MFMA) ----------
    f32x16 acc[NT];
    {
        const float* pi = a.PQ + (size_t)i * (2 * H) + 4 * hi;
        const float* pj = a.PQ + (size_t)j * (2 * H) + H + 4 * hi;
        const float* pg = a.G + (size_t)g * H + 4 * hi;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if ((t & 1) == 0) __builtin_amdgcn_sched_barrier(0);  // two tiles (24 float4) in flight at a time
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 x = *reinterpret_cast<const f32x4*>(pi + 32 * t + 8 * q);
                f32x4 y = *reinterpret_cast<const f32x4*>(pj + 32 * t + 8 * q);
                f32x4 z = *reinterpret_cast<const f32x4*>(pg + 32 * t + 8 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][4 * q + c] = (x[c] + y[c]) + z[c];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- GEMM1: Z1^T += Wff * ff^T,  K = 2*KP, four k-steps per packed float4 -------------
    // One wave per SIMD: nothing else hides memory latency, so the weight stream is software
    // pipelined through two register buffers (the loads of step m+1 fly under the 64 MFMAs of
    // step m); sched_barriers keep the compiler from hoisting or sinking the stream.
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wff_p) + lane;
        int c = 0, k = 0;  // pair s = c*F + k
        const int nm = a.KP / 4;  // even (KP % 8 == 0)
        auto fourier4 = [&](float (&bv)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dc = c == 0 ? d0 : (c == 1 ? d1 : d2);
                float arg = dc * a.freqs[k];  // emb = x * freq  (cspnet.py:21)
                float sn, cs;
                sincos_bounded(arg, &sn, &cs);
                bv[q] = hi ? cs : sn;         // hi = 0 lanes carry sin(c,k), hi = 1 lanes cos(c,k)
                if (++k == a.F) { k = 0; ++c; }
                if (c > 2) { c = 2; k = a.F - 1; }  // padding pairs: weights are zero
            }
        };
        f32x4 wa[NT], wb[NT];
        float bv[4];
#pragma unroll
        for (int t = 0; t < NT; ++t) wa[t] = wp[(size_t)t * 64];
        fourier4(bv);
        for (int m = 0; m < nm; m += 2) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) wb[t] = wp[((size_t)(m + 1) * NT + t) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t][q], bv[q], acc[t], 0, 0, 0);
            fourier4(bv);
            __builtin_amdgcn_sched_barrier(0);
            const int mn = (m + 2 < nm) ? m + 2 : m;  // tail: harmless reload
#pragma unroll
            for (int t = 0; t < NT; ++t) wa[t] = wp[((size_t)mn * NT + t) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[t][q], bv[q], acc[t], 0, 0, 0);
            fourier4(bv);
        }
    }

    // lane (e, hi), register r of tile t  <->  feature 32t + 8(r>>2) + 4hi + (r&3)
    if (a.Z1) {
        float* z = a.Z1 + (size_t)e * H + 4 * hi;
        if (e_l < nvalid)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(z + 32 * t + 8 * q) = v;
                }
    }

    // ---- segment structure of this tile (runs of equal src) -------------------------------
    const int i_prev = __shfl_up(i, 1, 64);
    const bool is_start = (hi == 0) && (e_l < nvalid) && (e_l == 0 || i != i_prev);
    const uint32_t starts = (uint32_t)__ballot(is_start);
    const int tile = (int)blockIdx.x;

    // ---- GEMM2: Z2^T = W2 * M1^T + b2, then SiLU and the per-node partial sums ------------
    // Weight stream through a RING-deep register ring (one ring slot = the UG float4 of one
    // (t,q) step = 4*UG MFMAs); SiLU of M1 tile t+1 is issued under the MFMAs of tile t during
    // the first pass.
    constexpr int RING = MI_RING, NSTEP = NT * 4;
    const f32x4* w2p = reinterpret_cast<const f32x4*>(a.W2_p) + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = silu(acc[0][r]);

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
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int t = st >> 2, q = st & 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int uu = 0; uu < UG; ++uu)
                    o[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[st % RING][uu][c], acc[t][4 * q + c], o[uu], 0, 0, 0);
            if constexpr (FIRST) {
                if (t + 1 < NT) {  // SiLU of the next M1 tile, a quarter per step, under these MFMAs
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[t + 1][4 * q + c] = silu(acc[t + 1][4 * q + c]);
                }
            }
            if (st + RING < NSTEP) {
#pragma unroll
                for (int uu = 0; uu < UG; ++uu) ring[st % RING][uu] = wload(st + RING, uu);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

#pragma unroll
        for (int uu = 0; uu < UG; ++uu) {
            const int u = ug + uu;
            if (a.Z2 && e_l < nvalid) {
                float* z = a.Z2 + (size_t)e * H + 32 * u + 4 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {o[uu][4 * q], o[uu][4 * q + 1], o[uu][4 * q + 2], o[uu][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(z + 8 * q) = v;
                }
            }
            // transpose through LDS: tr[edge][feature]
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {silu(o[uu][4 * q]), silu(o[uu][4 * q + 1]), silu(o[uu][4 * q + 2]), silu(o[uu][4 * q + 3])};
                *reinterpret_cast<f32x4*>(&tr[e_l * 36 + 8 * q + 4 * hi]) = v;
            }
            __syncthreads();
            // lane (f = e_l, hi) sums the segments of parity hi in edge order
            uint32_t rem = starts;
            int seg = 0;
            while (rem) {
                const int s = __builtin_ctz(rem);
                rem &= rem - 1;
                const int end = rem ? __builtin_ctz(rem) : nvalid;
                const int node = __builtin_amdgcn_readlane(i, s);  // s is wave-uniform
                if ((seg & 1) == hi) {
                    float sum = 0.f;
                    for (int x = s; x < end; ++x) sum += tr[x * 36 + e_l];
                    const int slot = tile - (a.rowptr[node] >> 5);
                    a.part[((size_t)slot * a.N + node) * H + 32 * u + e_l] = sum;
                }
                ++seg;
            }
        }
    };
    pass(0, std::true_type{});
    for (int ug = UG; ug < NT; ug += UG) pass(ug, std::false_type{});
}

}  // namespace mi

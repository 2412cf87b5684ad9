// MatterGen-shaped path on gfx950: a GemNet-T-shaped denoiser (periodic radius graph, Gaussian radial basis x polynomial
// envelope, spherical-harmonic triplet basis with the efficient bilinear contraction, 4 interaction blocks, per-edge force /
// stress heads, type logits), its parameter-gradient backward, the three corruptions and the predictor-corrector sampler.
//
// PARITY UNPINNED: the reference adapts these from the un-vendored package `mattergen @ 5bb2b397` (models/mattergen/*.py);
// the arithmetic here is the one restated definition by definition in oracle/mattergen_oracle.py, which the GPU tests
// compare against.
//
// Structure: the network is written ONCE as a program over eight tensor ops (dense, multiply, scaled add with optional row
// permutation, segmented sum, triplet contraction, weighted row dot, embedding lookup, heads).  Running the program
// executes the forward kernels and -- in training mode -- records a tape; the backward walks the tape in reverse with each
// op's gradient kernels.  All intermediates live in one arena per batch handle (a dry run of the program sizes it), the
// gradient arena mirrors it.  Dense layers run on the fp32-operand split GEMM (gemm_nt: three bf16 planes, six MFMA terms,
// fp32-class accuracy), weight gradients on gemm_tn_auto, data gradients on gemm_nt against transposed weight copies.
#include <string.h>

#include <chrono>
#include <map>
#include <set>
#include <string>
#include <vector>

#include <mutex>

#include "gemm_split.h"

namespace mi {

constexpr int GN_NMAX = 64;    // atoms per crystal (LDS-resident coordinates)
constexpr int GN_CAND = 512;   // candidates within the cutoff kept per target atom (more: the radius is bisected down)
constexpr int GN_DEG = 128;    // in-degree capacity after symmetrisation
constexpr int GN_CODE_BITS = 11;  // image code < (2 * 5 + 1)^3 = 1331
constexpr float GN_ACT = 1.66666666666666667f;
constexpr float GN_ISQ2 = 0.70710678118654752440f;
// large dense layers of the forward: two fp16 planes split on the fly (3 MFMA terms) instead of three bf16 planes (6).  OFF by default:
// measured at 256k edges the fp32-operand kernel is bound by loading / splitting / staging its operands, not by the matrix pipe
// (741 -> 722 us per product), and the absmax passes the scales need cost 12 % of the step -- kept as a tested option
int g_mg_f16 = 0;
// edge-level dense layers on the pre-split plane-set kernel (gemm_planes: operands split ONCE where they are produced -- weights at
// mi_gemnet_set_params, activations in their producer's epilogue, scaled by a power of two from a rigorous one-layer bound on the exact
// absmax of the producer's inputs): measured 2.0-2.6x the fp32-operand kernel on every shape of this network
int g_mg_planes = 1;
// training: the data gradients of the edge-level dense layers (dX += dZ W) on the plane-set kernel as well -- dZ written as a plane set by
// the activation-gradient pass (scale from the exact absmax of dY: |dZ| <= |s| max|silu'| max|dY|), against plane sets of the transposed
// weight blocks built at first use after mi_gemnet_set_params; three fp16 terms instead of the fp32-operand kernel's six bf16 terms
int g_mg_bwd_planes = 1;
// inference forwards in plane mode keep every edge-level tensor in ONE format -- the plane set where a dense layer reads it, fp32 rows
// otherwise -- instead of both (elementwise consumers and residual merges reconstruct x = (h0 + h1) / scale, exact in fp32), fold the
// skip-connection merges into the last layer of the residual stack they close and the radial weighting into the edge -> atom sum.
// Training forwards keep both formats (the tape's gradient kernels read fp32 rows).
int g_mg_nosync = 1;   // the sampler's forwards run without a host round trip per evaluation (forward_impl); 0: the synchronising form (mi_debug_set_mg_nosync)
int g_mg_deg_cap = GN_DEG;   // in-degree capacity a crystal may reach before it is taken out of the graph (<= GN_DEG; mi_debug_set_mg_deg_cap)
int g_mg_lean = 63;
static const bool g_optime = getenv("MI_DEBUG_OPTIME") != nullptr;   // bit 0: one format per tensor, bit 1: folded skip merges, bit 2: weighted edge -> atom sum in one pass, bit 3: residuals read from plane sets, bit 4: scalar heads through the derived tensors, bit 5: radial multiplicand / reversed-edge merge in the dense epilogues
constexpr int64_t MG_PLANES_MIN_ROWS = 4096;
constexpr int AMAX_SLOTS = 2048;
// an absmax slot is a row of AMAX_W sub-slots: a producer's waves spread their atomicMax over them (32k same-address atomics of one
// [256k, 512] product were a 270 us floor under every edge-level layer), readers fold the row
constexpr int AMAX_W = 64;
constexpr int LOGIT_LD = 104;  // row stride of the logits buffer (101 padded to a multiple of 4: GEMM operand alignment)

enum : uint32_t {  // Philox draw ids of this path (DESIGN.md "RNG"; mirrored by oracle-side helpers in the tests)
    DRAW_MG_POS = 10, DRAW_MG_CELL = 11, DRAW_MG_TYPES = 12, DRAW_MG_CORR_POS = 13, DRAW_MG_CORR_CELL = 14, DRAW_MG_PRED_POS = 15,
    DRAW_MG_PRED_CELL = 16, DRAW_MG_PRED_U1 = 17, DRAW_MG_PRED_U2 = 18, DRAW_MG_INIT_POS = 19, DRAW_MG_INIT_CELL = 20,
};

// ================================================================================================================================
// graph
// ================================================================================================================================
struct GraphArgs {
    const float *pos, *cell;
    const int* node_off;
    float cutoff;
    int maxnb, R, cap;
    int *ent, *acnt, *deg, *mcount, *meta;
    int* bad;   // [B] per-crystal capacity flags (1 / 2 / 4 as in meta[2]); a flagged crystal gets NO edges (gg_fix_kernel)
};

// entries of target a: key = (c << 11) | code for each selected neighbour whose pair is represented by (a, c, code).
// grid (crystals, ceil(max atoms / 4)): a WAVE per target atom (one block per crystal left 256 four-wave blocks on 256 CUs for 1.5 ms:
// a wave walked five atoms' candidate lists one after the other); the in-degree counts of the crystal's atoms are integer atomics on
// g.deg (zeroed by the caller; order-independent), the degree-capacity flag is raised by gg_scan_kernel.
__global__ __launch_bounds__(256) void gg_select_kernel(GraphArgs g) {
#pragma clang fp contract(off)
    __shared__ float cart[GN_NMAX * 3];
    __shared__ float Ls[9];
    __shared__ int reps[3];
    __shared__ float dl_all[4][GN_CAND];
    __shared__ unsigned kl_all[4][GN_CAND];
    const int b = blockIdx.x, n0 = g.node_off[b], n = g.node_off[b + 1] - n0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* dl = dl_all[wave];
    unsigned* kl = kl_all[wave];
    const float* Lm = g.cell + (size_t)b * 9;
    if (tid < 9) Ls[tid] = Lm[tid];
    for (int i = tid; i < n; i += 256) {
        const float* f = g.pos + (size_t)(n0 + i) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) cart[i * 3 + c] = __builtin_fmaf(f[2], Lm[6 + c], __builtin_fmaf(f[1], Lm[3 + c], f[0] * Lm[c]));
    }
    if (tid == 0) {  // images per dimension: ceil(cutoff / inter-plane spacing), at most R
        const double a0 = Lm[0], a1 = Lm[1], a2 = Lm[2], b0 = Lm[3], b1 = Lm[4], b2 = Lm[5], c0 = Lm[6], c1 = Lm[7], c2 = Lm[8];
        const double x23[3] = {b1 * c2 - b2 * c1, b2 * c0 - b0 * c2, b0 * c1 - b1 * c0};
        const double x31[3] = {c1 * a2 - c2 * a1, c2 * a0 - c0 * a2, c0 * a1 - c1 * a0};
        const double x12[3] = {a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0};
        const double vol = fabs(a0 * x23[0] + a1 * x23[1] + a2 * x23[2]);
        const double* xs[3] = {x23, x31, x12};
        for (int k = 0; k < 3; ++k) {
            const double nrm = sqrt(xs[k][0] * xs[k][0] + xs[k][1] * xs[k][1] + xs[k][2] * xs[k][2]);
            const double spacing = vol / fmax(nrm, 1e-30);
            double r = ceil((double)g.cutoff / fmax(spacing, 1e-30) - 1e-9);
            if (!(r >= 1.0)) r = 1.0;
            if (!(r <= (double)g.R)) r = (double)g.R;
            reps[k] = (int)r;
        }
    }
    __syncthreads();
    const int ra = reps[0], rb = reps[1], rc = reps[2], wb = 2 * rb + 1, wc = 2 * rc + 1, W = 2 * g.R + 1;
    const int nimg = (2 * ra + 1) * wb * wc, nq = nimg * n;
    const float r2 = g.cutoff * g.cutoff;
    const int zero_code = (g.R * W + g.R) * W + g.R;
    const int a = blockIdx.y * 4 + wave;
    if (a < n) {
        const float cx = cart[a * 3], cy = cart[a * 3 + 1], cz = cart[a * 3 + 2];
        // d^2 and key of candidate q = (image, source), or d^2 = -1 outside (1e-6, limit]
        auto cand = [&](int q, float limit, unsigned* key) -> float {
            const int ii = q / n, c = q - ii * n;
            const int ia = ii / (wb * wc) - ra, ib = (ii / wc) % wb - rb, ic = ii % wc - rc;
            const float fa = (float)ia, fb = (float)ib, fc = (float)ic;
            const float ox = (fa * Ls[0] + fb * Ls[3]) + fc * Ls[6], oy = (fa * Ls[1] + fb * Ls[4]) + fc * Ls[7],
                        oz = (fa * Ls[2] + fb * Ls[5]) + fc * Ls[8];
            const float dx = (cart[c * 3] + ox) - cx, dy = (cart[c * 3 + 1] + oy) - cy, dz = (cart[c * 3 + 2] + oz) - cz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            *key = ((unsigned)c << GN_CODE_BITS) | (unsigned)(((ia + g.R) * W + (ib + g.R)) * W + (ic + g.R));
            return (d <= limit && d > 1e-6f) ? d : -1.f;
        };
        auto collect = [&](float limit) -> int {
            int m = 0;
            for (int q0 = 0; q0 < nq; q0 += 64) {
                const int q = q0 + lane;
                unsigned key = 0;
                const float d = q < nq ? cand(q, limit, &key) : -1.f;
                const bool pass = d >= 0.f;
                const uint64_t mk = __ballot(pass);
                if (pass) {
                    const int p = m + __popcll(mk & ((1ull << lane) - 1ull));
                    if (p < GN_CAND) {
                        dl[p] = d;
                        kl[p] = key;
                    }
                }
                m += __popcll(mk);
            }
            return m;
        };
        int m = collect(r2);
        if (m > GN_CAND) {  // a very dense cell: shrink the radius until the list fits but still holds the maxnb nearest
            float lo = 0.f, hi = r2, mid = r2;
            bool ok = false;
            for (int it = 0; it < 24 && !ok; ++it) {
                mid = 0.5f * (lo + hi);
                m = collect(mid);
                if (m > GN_CAND) hi = mid;
                else if (m < g.maxnb) lo = mid;
                else ok = true;
            }
            if (!ok && lane == 0) {
                atomicOr(&g.meta[2], 2);
                atomicOr(&g.bad[b], 2);
            }
            if (m > GN_CAND) m = GN_CAND;
        }
        __builtin_amdgcn_wave_barrier();
        int cnt = 0;
        for (int p0 = 0; p0 < m; p0 += 64) {
            const int p = p0 + lane;
            bool sel = false;
            unsigned key = 0;
            if (p < m) {
                const float v = dl[p];
                key = kl[p];
                int rank = 0;
                for (int k = 0; k < m; ++k) {
                    const float x = dl[k];
                    rank += (x < v) || (x == v && kl[k] < key);
                }
                const int c = (int)(key >> GN_CODE_BITS), code = (int)(key & ((1u << GN_CODE_BITS) - 1u));
                sel = rank < g.maxnb && (c < a || (c == a && code < zero_code));
            }
            const uint64_t mk = __ballot(sel);
            if (sel) {
                const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (pos < g.cap) g.ent[(size_t)(n0 + a) * g.cap + pos] = (int)key;
                atomicAdd(&g.deg[n0 + (int)(key >> GN_CODE_BITS)], 1);
            }
            cnt += __popcll(mk);
        }
        if (lane == 0) {
            g.acnt[n0 + a] = cnt;
            atomicAdd(&g.deg[n0 + a], cnt);
            if (cnt > g.cap) {
                atomicOr(&g.meta[2], 1);
                atomicOr(&g.bad[b], 1);
            }
        }
    }
}

// A crystal whose neighbour lists do not fit the capacities (flags 1 / 2 from gg_select_kernel, 4 = an in-degree above `dg_cap`, the
// triplet kernels' LDS capacity of this launch) is taken OUT of the graph: its atoms get no edges in this evaluation and the flag
// stays in bad[] (sticky until the caller clears it), so the rest of the batch goes on and the caller drops exactly the offending
// crystals afterwards -- what the reference's invalid_filter does with collapsed cells (pipeline/filters/opt_filter.py:49-61).  It is
// also what makes every later kernel safe without a host-side check: the edge count of the crystals that remain is at most
// 2 cap N = E_cap.  One wave per crystal.
__global__ __launch_bounds__(64) void gg_fix_kernel(const int* __restrict__ node_off, int* __restrict__ deg, int* __restrict__ acnt, int* __restrict__ bad,
                                                    int* __restrict__ meta, int dg_cap) {
    const int b = blockIdx.x, lane = threadIdx.x, n0 = node_off[b], n1 = node_off[b + 1];
    int mx = 0;
    for (int i = n0 + lane; i < n1; i += 64) mx = deg[i] > mx ? deg[i] : mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
    }
    int flags = bad[b];
    if (mx > dg_cap) flags |= 4;
    if (flags == 0) return;
    if (lane == 0) {
        bad[b] = flags;
        atomicOr(&meta[2], flags);
    }
    for (int i = n0 + lane; i < n1; i += 64) {
        deg[i] = 0;
        acnt[i] = 0;
    }
}

// rowptr = exclusive scan of deg; meta = {E, max degree, flags}
__global__ __launch_bounds__(1024) void gg_scan_kernel(const int* __restrict__ deg, int N, int* __restrict__ rowptr, int* __restrict__ meta) {
    __shared__ int part[1024], pmax[1024];
    const int tid = threadIdx.x, chunk = (N + 1023) / 1024, lo = tid * chunk, hi = lo + chunk < N ? lo + chunk : N;
    int s = 0, mx = 0;
    for (int k = lo; k < hi; ++k) {
        s += deg[k];
        mx = deg[k] > mx ? deg[k] : mx;
    }
    part[tid] = s;
    pmax[tid] = mx;
    __syncthreads();
    if (tid == 0) {
        int run = 0, m2 = 0;
        for (int k = 0; k < 1024; ++k) {
            const int v = part[k];
            part[k] = run;
            run += v;
            m2 = pmax[k] > m2 ? pmax[k] : m2;
        }
        rowptr[N] = run;
        meta[0] = run;
        meta[1] = m2;
        if (m2 > GN_DEG) atomicOr(&meta[2], 4);   // in-degree above the triplet kernels' capacity
    }
    __syncthreads();
    int run = part[tid];
    for (int k = lo; k < hi; ++k) {
        rowptr[k] = run;
        run += deg[k];
    }
}

struct EmitArgs {
    const int *node_off, *ent, *acnt, *rowptr, *meta;
    int cap, R;
    int64_t E_cap;
    int *src, *dst, *code, *ekey, *swap, *edge_graph;
};

// rows sorted by (source, image code); both directions of every kept pair; then the index of each edge's reverse
__global__ __launch_bounds__(256) void gg_emit_kernel(EmitArgs g) {
    extern __shared__ int el[];  // [n * cap] entries (a << 17 | key)
    __shared__ int aoff[GN_NMAX + 1];
    __shared__ unsigned rowbuf[4][GN_DEG];
    if ((int64_t)g.meta[0] > g.E_cap) return;   // (cannot happen once gg_fix_kernel has removed the crystals over capacity: E <= 2 cap N)
    const int b = blockIdx.x, n0 = g.node_off[b], n = g.node_off[b + 1] - n0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = 2 * g.R + 1;
    if (tid == 0) {
        int s = 0;
        for (int i = 0; i < n; ++i) {
            aoff[i] = s;
            s += g.acnt[n0 + i];
        }
        aoff[n] = s;
    }
    __syncthreads();
    const int M = aoff[n];
    for (int i = 0; i < n; ++i)
        for (int k = tid; k < aoff[i + 1] - aoff[i]; k += 256) el[aoff[i] + k] = (i << 17) | g.ent[(size_t)(n0 + i) * g.cap + k];
    __syncthreads();
    const unsigned cmask = (1u << GN_CODE_BITS) - 1u;
    for (int v = wave; v < n; v += 4) {
        unsigned* rb = rowbuf[wave];
        int cnt = 0;
        for (int t0 = 0; t0 < M; t0 += 64) {
            const int t = t0 + lane;
            bool hit = false;
            unsigned key = 0;
            if (t < M) {
                const int w = el[t], a = w >> 17, c = (w >> GN_CODE_BITS) & 63;
                const unsigned code = (unsigned)w & cmask;
                if (a == v) {  // c -> v with this image
                    hit = true;
                    key = ((unsigned)c << GN_CODE_BITS) | code;
                } else if (c == v) {  // the reverse: a -> v with the negated image
                    const int ia = (int)(code / (W * W)) - g.R, ib = (int)((code / W) % W) - g.R, ic = (int)(code % W) - g.R;
                    hit = true;
                    key = ((unsigned)a << GN_CODE_BITS) | (unsigned)(((-ia + g.R) * W + (-ib + g.R)) * W + (-ic + g.R));
                }
            }
            const uint64_t mk = __ballot(hit);
            if (hit) {
                const int p = cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (p < GN_DEG) rb[p] = key;
            }
            cnt += __popcll(mk);
            // second direction of self pairs
            bool hit2 = false;
            unsigned key2 = 0;
            if (t < M) {
                const int w = el[t], a = w >> 17, c = (w >> GN_CODE_BITS) & 63;
                if (a == v && c == v) {
                    const unsigned code = (unsigned)w & cmask;
                    const int ia = (int)(code / (W * W)) - g.R, ib = (int)((code / W) % W) - g.R, ic = (int)(code % W) - g.R;
                    hit2 = true;
                    key2 = ((unsigned)a << GN_CODE_BITS) | (unsigned)(((-ia + g.R) * W + (-ib + g.R)) * W + (-ic + g.R));
                }
            }
            const uint64_t mk2 = __ballot(hit2);
            if (hit2) {
                const int p = cnt + __popcll(mk2 & ((1ull << lane) - 1ull));
                if (p < GN_DEG) rb[p] = key2;
            }
            cnt += __popcll(mk2);
        }
        __builtin_amdgcn_wave_barrier();
        const int r0 = g.rowptr[n0 + v];
        for (int p = lane; p < cnt && p < GN_DEG; p += 64) {
            const unsigned key = rb[p];
            int rank = 0;
            for (int k = 0; k < cnt; ++k) rank += rb[k] < key;
            const int e = r0 + rank;
            g.src[e] = n0 + (int)(key >> GN_CODE_BITS);
            g.dst[e] = n0 + v;
            g.code[e] = (int)(key & cmask);
            g.ekey[e] = (int)key;
            g.edge_graph[e] = b;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence();
    __syncthreads();
    // swap[e]: in row src(e), the edge coming from dst(e) with the negated image (rows are sorted by key: binary search)
    const int e0 = g.rowptr[n0], e1 = g.rowptr[n0 + n];
    for (int e = e0 + tid; e < e1; e += 256) {
        const int s = g.src[e] - n0, v = g.dst[e] - n0;
        const unsigned code = (unsigned)g.code[e];
        const int ia = (int)(code / (W * W)) - g.R, ib = (int)((code / W) % W) - g.R, ic = (int)(code % W) - g.R;
        const int want = (int)(((unsigned)v << GN_CODE_BITS) | (unsigned)(((-ia + g.R) * W + (-ib + g.R)) * W + (-ic + g.R)));
        int lo = g.rowptr[n0 + s], hi = g.rowptr[n0 + s + 1] - 1, found = -1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1, k = g.ekey[mid];
            if (k == want) {
                found = mid;
                break;
            }
            if (k < want) lo = mid + 1;
            else hi = mid - 1;
        }
        g.swap[e] = found;
    }
}

// Row count of an edge-level launch: `rows` is what the launch was sized for; when the graph's edge count is only known on the device
// (forwards without a host round trip, see forward_impl) it is a capacity and *mdev (= meta[0]) the number of rows that exist.
__device__ __forceinline__ int64_t dev_rows(int64_t rows, const int* __restrict__ mdev) {
    if (!mdev) return rows;
    const int64_t m = *mdev;
    return m < rows ? m : rows;
}

// D, V and the radial basis (polynomial envelope p = 5 x Gaussian smearing on d = D / cutoff): one thread per (edge, radial index)
__global__ void edge_geom_rbf_kernel(const float* __restrict__ pos, const float* __restrict__ cell, const int* __restrict__ src,
                                     const int* __restrict__ dst, const int* __restrict__ code, const int* __restrict__ edge_graph, int R_img,
                                     float cutoff, int NR, int64_t E, float* __restrict__ D, float* __restrict__ V, float* __restrict__ rbf,
                                     const int* __restrict__ mdev = nullptr) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    E = dev_rows(E, mdev);
    if (idx >= E * NR) return;
    const int e = (int)(idx / NR), r = (int)(idx % NR);
    const float* L = cell + (size_t)edge_graph[e] * 9;
    const int W = 2 * R_img + 1, cd = code[e];
    const float fa = (float)(cd / (W * W) - R_img), fb = (float)((cd / W) % W - R_img), fc = (float)(cd % W - R_img);
    const float *ps = pos + (size_t)src[e] * 3, *pd = pos + (size_t)dst[e] * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float cs = fmaf(ps[2], L[6 + c], fmaf(ps[1], L[3 + c], ps[0] * L[c]));
        const float cdn = fmaf(pd[2], L[6 + c], fmaf(pd[1], L[3 + c], pd[0] * L[c]));
        const float o = (fa * L[c] + fb * L[3 + c]) + fc * L[6 + c];
        v[c] = (cs + o) - cdn;
    }
    const float dd = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (r == 0) {
        D[e] = dd;
        V[(size_t)e * 3] = v[0] / dd;
        V[(size_t)e * 3 + 1] = v[1] / dd;
        V[(size_t)e * 3 + 2] = v[2] / dd;
    }
    const float d = dd / cutoff;
    const float d2 = d * d, d5 = d2 * d2 * d;
    const float env = d < 1.f ? 1.f + d5 * (-21.f + d * (35.f - 15.f * d)) : 0.f;
    const float step = 1.f / (float)(NR - 1), off = (float)r * step, coeff = -0.5f / (step * step);
    rbf[idx] = env * expf(coeff * (d - off) * (d - off));
}

// noise-level encoding of t in (0, 1]: [sin(1000 t div_k) | cos(1000 t div_k)], div_k = exp(-k ln(1e4) / half)
__global__ void nle_kernel(const float* __restrict__ t, float* __restrict__ z, int B, int A) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * A) return;
    const int b = idx / A, k = idx % A, half = A / 2, kk = k < half ? k : k - half;
    const float div = expf((float)kk * (-9.21034037197618273607f / (float)half));
    const float arg = (t[b] * 1000.0f) * div;
    z[idx] = k < half ? sinf(arg) : cosf(arg);
}

// ================================================================================================================================
// op kernels
// ================================================================================================================================
__global__ void embed_fwd_kernel(const float* __restrict__ table, const int* __restrict__ types, float* __restrict__ Y, int N, int A) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * A) return;
    Y[idx] = table[(size_t)(types[idx / A] - 1) * A + idx % A];
}
// one block per table row: rows are added in atom order (deterministic)
__global__ void embed_bwd_kernel(const float* __restrict__ dY, const int* __restrict__ types, float* __restrict__ dT, int N, int A) {
    const int r = blockIdx.x;
    for (int k = threadIdx.x; k < A; k += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < N; ++i)
            if (types[i] - 1 == r) s += dY[(size_t)i * A + k];
        dT[(size_t)r * A + k] += s;
    }
}
__global__ void mul_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] * b[i];
}
__global__ void mul_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dy, float* __restrict__ da,
                               float* __restrict__ db, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = dy[i];
    if (da) da[i] += g * b[i];
    if (db) db[i] += g * a[i];
}
// y = sf[0] * x and its gradient (GemNet's ScalingFactor with a fitted value: op_scale; identity factors launch nothing)
__global__ void scale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ sf, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * sf[0];
}
__global__ void scale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ sf, float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] += dy[i] * sf[0];
}
// y = (a + b[perm]) * s   (perm: row permutation of b, or NULL)
__global__ void axpby_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const int* __restrict__ perm, float s, float* __restrict__ y,
                                 int64_t rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i % cols);
    y[i] = (a[i] + b[(perm ? (int64_t)perm[r] : r) * cols + c]) * s;
}
// da += s dy;  db[r] += s dy[perm[r]]  (perm is an involution)
__global__ void axpby_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ perm, float s, float* __restrict__ da, float* __restrict__ db,
                                 int64_t rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i % cols);
    if (da) da[i] += s * dy[i];
    if (db) db[i] += s * dy[(perm ? (int64_t)perm[r] : r) * cols + c];
}
__device__ __forceinline__ float ssilu_grad(float z) { return silu_grad(z) * GN_ACT; }
// y = (act(z) + res) * s:   dres += s dY,   dZ = s dY act'(Z)   (z == NULL: no activation)
__global__ void act_bwd_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ z, float s, float* __restrict__ dres,
                               float* __restrict__ dz, int64_t rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i % cols);
    const float g = dy[r * ldy + c] * s;
    if (dres) dres[i] += g;
    dz[i] = z ? g * ssilu_grad(z[i]) : g;
}
// Y[v] (+)= sum over the rows of segment v of X[perm ? perm[e] : e]
__global__ void segsum_kernel(const float* __restrict__ X, int ldx, const int* __restrict__ segptr, const int* __restrict__ perm, float* __restrict__ Y,
                              int nseg, int cols, int acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nseg * cols) return;
    const int v = (int)(i / cols), c = (int)(i % cols);
    float s = 0.f;
    for (int e = segptr[v]; e < segptr[v + 1]; ++e) s += X[(size_t)(perm ? perm[e] : e) * ldx + c];
    Y[i] = acc ? Y[i] + s : s;
}
// dX[e] += dY[seg_of[e]]
__global__ void gather_add_kernel(const float* __restrict__ dY, const int* __restrict__ seg_of, float* __restrict__ dX, int64_t rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    dX[i] += dY[(size_t)seg_of[i / cols] * cols + i % cols];
}

// ---- plane-set plumbing of the forward (fp16 two-plane / bf16 three-plane format of gemm_split.h) ---------------------------------
// the value of an absmax slot: the largest of its sub-slots (all lanes of the calling wave get it)
__device__ __forceinline__ float amax_read(const unsigned* slot) {
    float m = __uint_as_float(slot[threadIdx.x & (AMAX_W - 1)]);   // (non-negative floats order like their bit patterns)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}
// slot[0] = the folded value (for readers that take one word: the on-the-fly fp16 split of the fp32-operand kernel)
__global__ void amax_fold_kernel(unsigned* slot) {
    const float m = amax_read(slot);
    if (threadIdx.x == 0) slot[0] = __float_as_uint(m);
}
// {scale, 1 / scale} of an output plane set from a rigorous bound:  bound = (fa (a * rs * b2 * deg * kmul + g1 + g2) + res) * s
// with a, b2, g1, g2, res = exact absmax bit patterns of the inputs (NULL = absent), rs = largest row sum of |W| of the layer
__global__ void mg_scale_kernel(const unsigned* a, const float* rs, const unsigned* b2, const int* degp, float kmul, const unsigned* g1, const unsigned* g2,
                                const unsigned* res, float fa, float s, float* dsc, const unsigned* res2 = nullptr, float s2 = 1.f, const unsigned* pmul = nullptr) {
    // (one wave: every input is a row of AMAX_W sub-slots)
    float t = amax_read(a) * (rs ? *rs : 1.f) * (b2 ? amax_read(b2) : 1.f) * (degp ? (float)*degp : 1.f) * kmul;
    if (g1) t += amax_read(g1);
    if (g2) t += amax_read(g2);
    t = fa * t * (pmul ? amax_read(pmul) : 1.f) + (res ? amax_read(res) : 0.f);   // (pmul: the multiplicand applied right after the activation)
    t *= s;
    if (res2) t = (t + amax_read(res2)) * s2;   // the folded second merge
    int e = 14 - (int)ceilf(log2f(fmaxf(t, 1e-30f)));
    if (!(t == t) || t > 3e38f) e = -100;
    e = e > 30 ? 30 : (e < -100 ? -100 : e);
    if (threadIdx.x == 0) {
        dsc[0] = exp2f((float)e);
        dsc[1] = exp2f(-(float)e);
    }
}
// max over rows of sum_k |W[row][k]|, k in [0, K): a wave per row, the blocks' maxima merged by atomicMax on the bit pattern (non-negative
// floats order like unsigned integers; *out must be zero before the launch).  One block for the whole matrix took 200 us per weight --
// 21 ms after every parameter update of a fine-tune run.
__global__ __launch_bounds__(256) void rowsum_max_kernel(const float* __restrict__ W, int ldw, int rows, int K, float* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += 4 * gridDim.x) {
        float sacc = 0.f;
        for (int k = threadIdx.x & 63; k < K; k += 64) sacc += fabsf(W[(size_t)r * ldw + k]);
        m = fmaxf(m, wave_sum(sacc));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
__device__ __forceinline__ void store_pl_pair(const Planes& P, int64_t row, int col, float x, float y) {
    unsigned pr[3];
    pl_split_pair(x, y, P.s(), pr);
#pragma unroll
    for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(P.base + P.elem((int)row, col, k)) = pr[k];
}
// act_bwd_kernel that also leaves dZ as a plane set (the A operand of the data-gradient product); a thread per column pair.
// dz == NULL: planes only (layers without activation / merge, whose dZ is dY itself).
__global__ void act_bwd_pl_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ z, float s, float* __restrict__ dres,
                                  float* __restrict__ dz, Planes P, int64_t rows, int cols) {
    const int cp = cols >> 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cp) return;
    const int64_t r = i / cp;
    const int c = (int)(i - r * cp) * 2;
    const float2 g2 = *reinterpret_cast<const float2*>(dy + r * ldy + c);
    float g0 = g2.x * s, g1 = g2.y * s;
    if (dres) {
        float2* d = reinterpret_cast<float2*>(dres + r * cols + c);
        float2 v = *d;
        v.x += g0;
        v.y += g1;
        *d = v;
    }
    if (z) {
        const float2 zz = *reinterpret_cast<const float2*>(z + r * cols + c);
        g0 *= ssilu_grad(zz.x);
        g1 *= ssilu_grad(zz.y);
    }
    if (dz) *reinterpret_cast<float2*>(dz + r * cols + c) = make_float2(g0, g1);
    store_pl_pair(P, r, c, g0, g1);
}
__device__ __forceinline__ void note_absmax(unsigned* slot, float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && slot) atomicMax(slot + (blockIdx.x & (AMAX_W - 1)), __float_as_uint(m));
}
// An edge-level operand as its consumer finds it: fp32 rows, or -- where an inference run kept only the plane set -- the planes
struct Src {
    const float* f = nullptr;
    Planes P;
};
__device__ __forceinline__ f32x2 src_load2(const Src& s, int64_t r, int c, int cols) {
    if (s.f) return *reinterpret_cast<const f32x2*>(s.f + r * cols + c);
    const float inv = s.P.dscale ? s.P.dscale[1] : 1.f / s.P.scale;
    float x = 0.f, y = 0.f;
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl) {   // smallest plane first; the sum is exact in fp32
        const unsigned w = *reinterpret_cast<const unsigned*>(s.P.base + s.P.elem((int)r, c, pl));
#if MI_PLANES_FP16
        const f16x2 h = __builtin_bit_cast(f16x2, w);
        x += (float)h[0];
        y += (float)h[1];
#else
        x += __uint_as_float(w << 16);
        y += __uint_as_float(w & 0xFFFF0000u);
#endif
    }
    return f32x2{x * inv, y * inv};
}
// plane set -> fp32 rows (a consumer without a plane-reading form, or a debug tap, asked for a tensor kept as planes only)
__global__ __launch_bounds__(256) void pl_to_f32_kernel(Src a, float* __restrict__ y, int64_t rows, int cols, const int* __restrict__ mdev = nullptr) {
    const int64_t n = dev_rows(rows, mdev) * (cols / 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (cols / 2);
        const int c = (int)(i % (cols / 2)) * 2;
        *reinterpret_cast<f32x2*>(y + r * cols + c) = src_load2(a, r, c, cols);
    }
}
// Y[v] = sum over the in-edges e of atom v of X[e] * Wt[e]  (the radially weighted edge -> atom sum in one pass: the product is never
// written); a block per atom, a thread per column pair
__global__ __launch_bounds__(256) void segsum_mul_kernel(Src X, const float* __restrict__ Wt, const int* __restrict__ segptr, float* __restrict__ Y, int cols) {
    const int v = blockIdx.x;
    const int e0 = segptr[v], e1 = segptr[v + 1];
    for (int c = 2 * threadIdx.x; c < cols; c += 2 * blockDim.x) {
        float s0 = 0.f, s1 = 0.f;
        for (int e = e0; e < e1; ++e) {
            const f32x2 x = src_load2(X, e, c, cols), w = *reinterpret_cast<const f32x2*>(Wt + (size_t)e * cols + c);
            s0 += x[0] * w[0];
            s1 += x[1] * w[1];
        }
        *reinterpret_cast<f32x2*>(Y + (size_t)v * cols + c) = f32x2{s0, s1};
    }
}
// y = a * b (fp32) + plane set + absmax; grid-stride over column pairs (one atomic per wave of a FIXED-size grid: a wave-per-pair-block
// launch serialised a million atomics on one address -- 8.8 ms instead of 0.3)
constexpr int EW_GRID = 4096;
__global__ __launch_bounds__(256) void mul_pl_kernel(Src a, Src b, float* __restrict__ y, Planes P, unsigned* amax, int64_t rows, int cols,
                                                     const int* __restrict__ mdev = nullptr) {
    float m = 0.f;
    const int64_t n = dev_rows(rows, mdev) * (cols / 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (cols / 2);
        const int c = (int)(i % (cols / 2)) * 2;
        const f32x2 av = src_load2(a, r, c, cols), bv = src_load2(b, r, c, cols);
        const float y0 = av[0] * bv[0], y1 = av[1] * bv[1];
        if (y) *reinterpret_cast<f32x2*>(y + r * cols + c) = f32x2{y0, y1};
        store_pl_pair(P, r, c, y0, y1);
        m = fmaxf(m, fmaxf(fabsf(y0), fabsf(y1)));
    }
    note_absmax(amax, m);
}
// y = (a + b[perm]) * s (fp32) + optional plane set + absmax
__global__ __launch_bounds__(256) void axpby_pl_kernel(Src a, Src b, const int* __restrict__ perm, float s, float* __restrict__ y, Planes P, unsigned* amax,
                                                       int64_t rows, int cols, const int* __restrict__ mdev = nullptr) {
    float m = 0.f;
    const int64_t n = dev_rows(rows, mdev) * (cols / 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (cols / 2);
        const int c = (int)(i % (cols / 2)) * 2;
        const f32x2 av = src_load2(a, r, c, cols);
        const f32x2 bv = src_load2(b, perm ? (int64_t)perm[r] : r, c, cols);
        const float y0 = (av[0] + bv[0]) * s, y1 = (av[1] + bv[1]) * s;
        if (y) *reinterpret_cast<f32x2*>(y + r * cols + c) = f32x2{y0, y1};
        if (P.base) store_pl_pair(P, r, c, y0, y1);
        m = fmaxf(m, fmaxf(fabsf(y0), fabsf(y1)));
    }
    note_absmax(amax, m);
}
// fp32 [rows][cols] -> plane set with the fixed scale of P (the radial basis, |x| <= 1); column padding written as zero
__global__ void split_fixed_kernel(const float* __restrict__ x, Planes P, int64_t rows, int cols, const int* __restrict__ mdev = nullptr) {
    const int cp = P.KT * 16;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dev_rows(rows, mdev) * cp) return;
    const int64_t r = i / cp;
    const int c = (int)(i % cp) * 2;
    store_pl_pair(P, r, c, c < cols ? x[r * cols + c] : 0.f, c + 1 < cols ? x[r * cols + c + 1] : 0.f);
}

// ---- triplet contraction ---------------------------------------------------------------------------------------------------
// One block per target atom a; its in-edges e (rows lo..hi) see each other as triplets (e, k), k != e, with angle cos = V_e . V_k.
//   Tm[e][i][j] = sum_l cbfW[e][l][i] * sum_{k != e} Y_l(cos_ek) xd[k][j],   Y_l = sqrt((2l+1)/(4 pi)) P_l
// Lane j holds feature j (TR <= 64 features), a wave walks the edges of the atom.
template <int S>
__device__ __forceinline__ void sph_l(float c, float (&y)[S]) {
    float pm2 = 1.f, pm1 = c;
    y[0] = 0.28209479177387814f;
    if (S > 1) y[1] = 0.48860251190291992f * c;
#pragma unroll
    for (int l = 2; l < S; ++l) {
        const float p = ((float)(2 * l - 1) * c * pm1 - (float)(l - 1) * pm2) / (float)l;
        y[l] = sqrtf((float)(2 * l + 1) * 0.07957747154594767f) * p;
        pm2 = pm1;
        pm1 = p;
    }
}
// Edges are processed in chunks of TFC: phase 1 -- every thread takes (edge, neighbour) pairs of the chunk and writes Y_l(cos) to LDS
// (the angle work is per PAIR, not per feature: done once instead of once per lane); phase 2 -- a wave per edge, lane = feature,
// reads the pair's eight Y values as two broadcast ds_read_b128 and one feature value per neighbour.
constexpr int TFC = 8;
template <int S>
__global__ __launch_bounds__(512) void triplet_fwd_kernel(const float* __restrict__ xd, const float* __restrict__ V, const float* __restrict__ cbfW,
                                                          const int* __restrict__ rowptr, float* __restrict__ Tm, int TR, int CB, Planes P, unsigned* amax, int DG) {
    // DG = in-degree capacity of THIS graph (its largest in-degree rounded up to a multiple of 4; at most GN_DEG): the LDS image is sized
    // by it, so that three or four workgroups share a CU at the usual 50-70 in-edges instead of two at the 128-edge worst case
    extern __shared__ __attribute__((aligned(16))) float sm[];  // Y [TFC][DG][8] | V [DG][3] | xd [DG][TR] | cbfW [TFC][S * CB]
    float* Ys = sm;
    float* Vs = sm + TFC * DG * 8;
    float* xs = Vs + DG * 3;
    float* ws = xs + DG * TR;
    const int a = blockIdx.x, lo = rowptr[a], deg = rowptr[a + 1] - lo, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < deg * 3; i += 512) Vs[i] = V[(size_t)lo * 3 + i];
    for (int i = tid; i < deg * TR; i += 512) xs[i] = xd[(size_t)lo * TR + i];
    __syncthreads();
    float tmax = 0.f;
    for (int c0 = 0; c0 < deg; c0 += TFC) {
        const int ne = deg - c0 < TFC ? deg - c0 : TFC;
        for (int p = tid; p < ne * deg; p += 512) {
            const int ee = p / deg, k = p - ee * deg, e = c0 + ee;
            float c = Vs[e * 3] * Vs[k * 3] + Vs[e * 3 + 1] * Vs[k * 3 + 1] + Vs[e * 3 + 2] * Vs[k * 3 + 2];
            c = fminf(fmaxf(c, -1.f), 1.f);
            float y[S];
            sph_l<S>(c, y);
            float o[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) o[l] = (l < S && k != e) ? y[l < S ? l : 0] : 0.f;
            f32x4* dst = reinterpret_cast<f32x4*>(Ys + ((size_t)ee * DG + k) * 8);
            dst[0] = f32x4{o[0], o[1], o[2], o[3]};
            dst[1] = f32x4{o[4], o[5], o[6], o[7]};
        }
        for (int i = tid; i < ne * S * CB; i += 512) ws[i] = cbfW[(size_t)(lo + c0) * S * CB + i];  // (read through LDS: per-lane global loads of
        __syncthreads();                                                                              //  wave-uniform weights were pure latency)
        for (int ee = wave; ee < ne; ee += 8) {
            const int e = c0 + ee;
            float acc[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[l] = 0.f;
            const f32x4* yr = reinterpret_cast<const f32x4*>(Ys + (size_t)ee * DG * 8);
#pragma unroll 4
            for (int k = 0; k < deg; ++k) {
                const f32x4 y0 = yr[2 * k], y1 = yr[2 * k + 1];
                const float x = lane < TR ? xs[k * TR + lane] : 0.f;
                acc[0] += y0[0] * x;
                acc[1] += y0[1] * x;
                acc[2] += y0[2] * x;
                acc[3] += y0[3] * x;
                if (S > 4) {
                    acc[4] += y1[0] * x;
                    acc[5] += y1[1] * x;
                    acc[6] += y1[2] * x;
                    acc[7] += y1[3] * x;
                }
            }
            const float* w = ws + ee * S * CB;
            if (lane < TR) {
                // plane address of (row, column i * TR + lane): the row part once per edge, TR columns = TR / 32 column tiles per step of i
                const int row = lo + e;
                u16* prow = P.base ? P.base + P.tile(row >> 7, 0) + (size_t)(row & 127) * 32 + (size_t)(lane >> 5) * 12288 + (lane & 31) : nullptr;
                const float pscale = P.base ? P.s() : 1.f;
                auto emit = [&](int i, float sacc) {
                    if (Tm) Tm[((size_t)row * CB + i) * TR + lane] = sacc;
                    if (P.base) {   // the operand of the bilinear product: lanes pair up, the even one stores both halves of a 32-bit plane word
                        const float nb = __shfl_down(sacc, 1, 64);
                        if ((lane & 1) == 0) {
                            unsigned pr[3];
                            pl_split_pair(sacc, nb, pscale, pr);
                            u16* d = prow + (size_t)i * (TR >> 5) * 12288;
#pragma unroll
                            for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(d + k * 4096) = pr[k];
                        }
                        tmax = fmaxf(tmax, fabsf(sacc));
                    }
                };
                if ((CB & 3) == 0 && (TR & 31) == 0) {   // four output features per step: the weights as 16-byte LDS reads
                    for (int i = 0; i < CB; i += 4) {
                        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int l = 0; l < S; ++l) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + l * CB + i);
#pragma unroll
                            for (int q = 0; q < 4; ++q) s4[q] += wv[q] * acc[l];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) emit(i + q, s4[q]);
                    }
                } else {
                    for (int i = 0; i < CB; ++i) {
                        float sacc = 0.f;
#pragma unroll
                        for (int l = 0; l < S; ++l) sacc += w[l * CB + i] * acc[l];
                        if (Tm) Tm[((size_t)row * CB + i) * TR + lane] = sacc;
                        if (P.base) {
                            const float nb = __shfl_down(sacc, 1, 64);
                            if ((lane & 1) == 0) store_pl_pair(P, row, i * TR + lane, sacc, nb);
                            tmax = fmaxf(tmax, fabsf(sacc));
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (P.base) {   // one atomic per workgroup
        __shared__ float wmax[8];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
        if (lane == 0) wmax[wave] = tmax;
        __syncthreads();
        if (tid == 0) {
            float m = wmax[0];
            for (int k = 1; k < 8; ++k) m = fmaxf(m, wmax[k]);
            atomicMax(amax + (blockIdx.x & (AMAX_W - 1)), __float_as_uint(m));
        }
    }
}
// backward: dxd[k][j] += sum_{e != k} sum_l Y_l(cos_ek) dacc_e[l][j],  dacc_e[l][j] = sum_i cbfW[e][l][i] dTm[e][i][j];
//           dcbfW[e][l][i] += sum_j acc_e[l][j] dTm[e][i][j]   (acc recomputed).
// Same structure as the forward: edges in chunks of TBC (a wave per edge), the chunk's Y_l(cos) computed once per PAIR into LDS and used
// twice (acc of the chunk's edges; the chunk's contribution to every row k), the chunk's weights and dTm rows staged through LDS by
// coalesced loads.  The per-edge [S x TR] x [TR x CB] product that is dcbfW runs from LDS (acc rows and dTm rows padded to TAS floats:
// conflict-free 16-byte reads), one output per thread -- the first version reduced every output with a wave-wide sum and added it to
// global memory from lane 0 (112 dependent read-modify-writes per edge: 3 ms per call at 64k edges, 13x the forward).
constexpr int TBC = 8, TAS = 68;
static inline size_t triplet_bwd_lds_floats(int TR, int CB, int DG) {
    return (size_t)TBC * DG * 8 + DG * 3 + (size_t)DG * TR + TBC * 8 * 64 + (size_t)TBC * CB * 8 + TBC * 8 * TAS + (size_t)TBC * CB * TAS;
}
template <int S>
__global__ __launch_bounds__(512) void triplet_bwd_kernel(const float* __restrict__ xd, const float* __restrict__ V, const float* __restrict__ cbfW,
                                                          const int* __restrict__ rowptr, const float* __restrict__ dTm, float* __restrict__ dxd,
                                                          float* __restrict__ dcbfW, int TR, int CB, int DG) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ys = sm;                         // [TBC][DG][8]       Y_l(cos_ek), zero for k == e and l >= S   (DG: see the forward)
    float* Vs = Ys + TBC * DG * 8;          // [DG][3]
    float* xs = Vs + DG * 3;                // [DG][TR]
    float* da = xs + DG * TR;               // [TBC][8][64]       dacc of the chunk's edges
    float* wT = da + TBC * 8 * 64;          // [TBC][CB][8]       the chunk's weights, (i, l) order
    float* as = wT + TBC * CB * 8;          // [TBC][8][TAS]      acc of the chunk's edges
    float* gs = as + TBC * 8 * TAS;         // [TBC][CB][TAS]     the chunk's dTm rows
    const int a = blockIdx.x, lo = rowptr[a], deg = rowptr[a + 1] - lo, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < deg * 3; i += 512) Vs[i] = V[(size_t)lo * 3 + i];
    for (int i = tid; i < deg * TR; i += 512) xs[i] = xd[(size_t)lo * TR + i];
    float gk[GN_DEG / 8];  // this wave's rows k = wave, wave + 8, ...: accumulated gradient of feature `lane`
#pragma unroll
    for (int q = 0; q < GN_DEG / 8; ++q) gk[q] = 0.f;
    const int SC = S * CB, CT = CB * TR;
    __syncthreads();
    for (int c0 = 0; c0 < deg; c0 += TBC) {
        const int ne = deg - c0 < TBC ? deg - c0 : TBC;
        for (int p = tid; p < ne * deg; p += 512) {
            const int ee = p / deg, k = p - ee * deg, e = c0 + ee;
            float c = Vs[e * 3] * Vs[k * 3] + Vs[e * 3 + 1] * Vs[k * 3 + 1] + Vs[e * 3 + 2] * Vs[k * 3 + 2];
            c = fminf(fmaxf(c, -1.f), 1.f);
            float y[S];
            sph_l<S>(c, y);
            float o[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) o[l] = (l < S && k != e) ? y[l < S ? l : 0] : 0.f;
            f32x4* dst = reinterpret_cast<f32x4*>(Ys + ((size_t)ee * DG + k) * 8);
            dst[0] = f32x4{o[0], o[1], o[2], o[3]};
            dst[1] = f32x4{o[4], o[5], o[6], o[7]};
        }
        for (int i = tid; i < ne * SC; i += 512) {
            const int ee = i / SC, r = i - ee * SC, l = r / CB, ii = r - l * CB;
            wT[(ee * CB + ii) * 8 + l] = cbfW[(size_t)(lo + c0) * SC + i];
        }
        for (int i = tid; i < ne * CT; i += 512) {
            const int ee = i / CT, r = i - ee * CT, ii = r / TR, j = r - ii * TR;
            gs[(ee * CB + ii) * TAS + j] = dTm[(size_t)(lo + c0) * CT + i];
        }
        __syncthreads();
        if (wave < ne) {   // phase A: acc and dacc of edge c0 + wave, lane = feature
            const int ee = wave;
            float acc[8], dacc[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[l] = dacc[l] = 0.f;
            const f32x4* yr = reinterpret_cast<const f32x4*>(Ys + (size_t)ee * DG * 8);
#pragma unroll 4
            for (int k = 0; k < deg; ++k) {
                const f32x4 y0 = yr[2 * k], y1 = yr[2 * k + 1];
                const float x = lane < TR ? xs[k * TR + lane] : 0.f;
                acc[0] += y0[0] * x;
                acc[1] += y0[1] * x;
                acc[2] += y0[2] * x;
                acc[3] += y0[3] * x;
                if (S > 4) {
                    acc[4] += y1[0] * x;
                    acc[5] += y1[1] * x;
                    acc[6] += y1[2] * x;
                    acc[7] += y1[3] * x;
                }
            }
            for (int i = 0; i < CB; ++i) {
                const float g = lane < TR ? gs[(ee * CB + i) * TAS + lane] : 0.f;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wT + (ee * CB + i) * 8), w1 = *reinterpret_cast<const f32x4*>(wT + (ee * CB + i) * 8 + 4);
                dacc[0] += w0[0] * g;
                dacc[1] += w0[1] * g;
                dacc[2] += w0[2] * g;
                dacc[3] += w0[3] * g;
                if (S > 4) {
                    dacc[4] += w1[0] * g;
                    dacc[5] += w1[1] * g;
                    dacc[6] += w1[2] * g;
                    dacc[7] += w1[3] * g;
                }
            }
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                da[(ee * 8 + l) * 64 + lane] = l < S ? dacc[l] : 0.f;
                as[(ee * 8 + l) * TAS + lane] = l < S ? acc[l] : 0.f;
            }
        }
        __syncthreads();
        // dcbfW[e][l][i] += sum_j acc_e[l][j] dTm[e][i][j]: one output per thread, coalesced read-modify-write
        for (int o = tid; o < ne * SC; o += 512) {
            const int ee = o / SC, r = o - ee * SC, l = r / CB, ii = r - l * CB;
            const float* ar = as + (ee * 8 + l) * TAS;
            const float* gr = gs + (ee * CB + ii) * TAS;
            float sacc = 0.f;
            int j = 0;
            for (; j + 4 <= TR; j += 4) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ar + j), gv = *reinterpret_cast<const f32x4*>(gr + j);
                sacc += av[0] * gv[0] + av[1] * gv[1] + av[2] * gv[2] + av[3] * gv[3];
            }
            for (; j < TR; ++j) sacc += ar[j] * gr[j];
            dcbfW[(size_t)(lo + c0) * SC + o] += sacc;
        }
        // every row k collects the chunk's contributions (Y is zero where e == k)
        int q = 0;
        for (int k = wave; k < deg; k += 8, ++q) {
            float sk = 0.f;
            for (int ee = 0; ee < ne; ++ee) {
                const f32x4* yr = reinterpret_cast<const f32x4*>(Ys + ((size_t)ee * DG + k) * 8);
                const f32x4 y0 = yr[0], y1 = yr[1];
                const float* d = da + ee * 8 * 64 + lane;
                sk += y0[0] * d[0] + y0[1] * d[64] + y0[2] * d[128] + y0[3] * d[192];
                if (S > 4) sk += y1[0] * d[256] + y1[1] * d[320] + y1[2] * d[384] + y1[3] * d[448];
            }
            gk[q] += sk;
        }
        __syncthreads();
    }
    int q = 0;
    for (int k = wave; k < deg; k += 8, ++q)
        if (lane < TR) dxd[(size_t)(lo + k) * TR + lane] += gk[q];
}

// ---- weighted row dot: y[e] (+)= sum_k A[e][k] B[e][k] w[k]  (the per-edge scalar heads) -----------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ w,
                                                         float* __restrict__ y, int64_t rows, int K, int acc) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += A[r * K + k] * Bm[r * K + k] * (w ? w[k] : 1.f);
    s = wave_sum(s);
    if (lane == 0) y[r] = acc ? y[r] + s : s;
}
// short rows (K <= 16, a power of two): y[e] (+)= sum_k A[e][k] B[e][k], 64 / K rows per wave
__global__ __launch_bounds__(256) void rowdot_short_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ y, int64_t rows, int K, int acc,
                                                           const int* __restrict__ mdev = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    rows = dev_rows(rows, mdev);
    const int64_t r = i / K;
    float s = r < rows ? A[i] * Bm[i] : 0.f;
    for (int o = K >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (r < rows && (i % K) == 0) y[r] = acc ? y[r] + s : s;
}
__global__ void rowdot_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ w, const float* __restrict__ dy,
                                  float* __restrict__ dA, float* __restrict__ dB, int64_t rows, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const float g = dy[i / K] * w[i % K];
    dA[i] += g * Bm[i];
    dB[i] += g * A[i];
}
// P[chunk][k] = sum over the chunk's rows of dy[e] A[e][k] B[e][k]   (then part_reduce_kernel adds the chunks into dw)
__global__ __launch_bounds__(256) void rowdot_dw_kernel(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ dy,
                                                        float* __restrict__ P, int64_t rows, int K, int rows_per) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += dy[r] * A[r * K + k] * Bm[r * K + k];
    P[(size_t)blockIdx.y * K + k] = s;
}

// ---- heads -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv3(const float* L, float* I) {
    const float a = L[0], b = L[1], c = L[2], d = L[3], e = L[4], f = L[5], g = L[6], h = L[7], i = L[8];
    const float A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * Bc + c * C, id = 1.0f / det;
    I[0] = A * id;
    I[1] = -(b * i - c * h) * id;
    I[2] = (b * f - c * e) * id;
    I[3] = Bc * id;
    I[4] = (a * i - c * g) * id;
    I[5] = -(a * f - c * d) * id;
    I[6] = C * id;
    I[7] = -(a * h - b * g) * id;
    I[8] = (a * e - b * d) * id;
}
// pos[a] = (sum_{e into a} F[e] V[e]) @ inv(L)
__global__ void force_fwd_kernel(const float* __restrict__ F, const float* __restrict__ V, const int* __restrict__ rowptr, const int* __restrict__ n2g,
                                 const float* __restrict__ cell, float* __restrict__ pos, int N) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N) return;
    float f[3] = {0.f, 0.f, 0.f};
    for (int e = rowptr[a]; e < rowptr[a + 1]; ++e) {
        const float s = F[e];
        f[0] += s * V[(size_t)e * 3];
        f[1] += s * V[(size_t)e * 3 + 1];
        f[2] += s * V[(size_t)e * 3 + 2];
    }
    float I[9];
    inv3(cell + (size_t)n2g[a] * 9, I);
#pragma unroll
    for (int c = 0; c < 3; ++c) pos[(size_t)a * 3 + c] = f[0] * I[c] + f[1] * I[3 + c] + f[2] * I[6 + c];
}
// dF[e] += (dpos[dst] @ inv(L)^T) . V[e]
__global__ void force_bwd_kernel(const float* __restrict__ dpos, const float* __restrict__ V, const int* __restrict__ dst, const int* __restrict__ n2g,
                                 const float* __restrict__ cell, float* __restrict__ dF, int64_t E) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int a = dst[e];
    float I[9];
    inv3(cell + (size_t)n2g[a] * 9, I);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float df = dpos[(size_t)a * 3] * I[c * 3] + dpos[(size_t)a * 3 + 1] * I[c * 3 + 1] + dpos[(size_t)a * 3 + 2] * I[c * 3 + 2];
        s += df * V[(size_t)e * 3 + c];
    }
    dF[e] += s;
}
// stress[b] = (1 / max(E_b, 1)) sum_{e in crystal b} Sc[e] V[e] (x) V[e]
__global__ __launch_bounds__(256) void stress_fwd_kernel(const float* __restrict__ Sc, const float* __restrict__ V, const int* __restrict__ rowptr,
                                                         const int* __restrict__ node_off, float* __restrict__ out) {
    __shared__ float red[4][9];
    const int b = blockIdx.x, e0 = rowptr[node_off[b]], e1 = rowptr[node_off[b + 1]], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) s[q] = 0.f;
    for (int e = e0 + tid; e < e1; e += 256) {
        const float w = Sc[e], x = V[(size_t)e * 3], y = V[(size_t)e * 3 + 1], z = V[(size_t)e * 3 + 2];
        const float v[3] = {x, y, z};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) s[i * 3 + j] += w * v[i] * v[j];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const float r = wave_sum(s[q]);
        if (lane == 0) red[wave][q] = r;
    }
    __syncthreads();
    if (tid < 9) {
        const float cnt = (float)(e1 - e0 > 0 ? e1 - e0 : 1);
        out[(size_t)b * 9 + tid] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) / cnt;
    }
}
__global__ void stress_bwd_kernel(const float* __restrict__ dS, const float* __restrict__ V, const int* __restrict__ edge_graph,
                                  const int* __restrict__ rowptr, const int* __restrict__ node_off, float* __restrict__ dSc, int64_t E) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int b = edge_graph[e];
    const int cnt = rowptr[node_off[b + 1]] - rowptr[node_off[b]];
    const float v[3] = {V[(size_t)e * 3], V[(size_t)e * 3 + 1], V[(size_t)e * 3 + 2]};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s += dS[(size_t)b * 9 + i * 3 + j] * v[i] * v[j];
    dSc[e] += s / (float)(cnt > 0 ? cnt : 1);
}
// W [rows][cols] (row stride cols) -> WT [cols][ldt]
// q[k][c] = w_out[c] * Wr[c][k]   (Wr [Ed][Rb], q [Rb][Ed]; see mi_gemnet::dtheta)
__global__ void derive_q_kernel(const float* __restrict__ w_out, const float* __restrict__ Wr, float* __restrict__ q, int Ed, int Rb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Ed * Rb) return;
    const int k = (int)(i / Ed), c = (int)(i % Ed);
    q[i] = w_out[c] * Wr[(size_t)c * Rb + k];
}
__global__ void gn_transpose_kernel(const float* __restrict__ W, float* __restrict__ WT, int rows, int cols, int ldt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    WT[(size_t)c * ldt + r] = W[i];
}
__global__ void copy_ld_kernel(const float* __restrict__ X, int ldx, float* __restrict__ Y, int ldy, int64_t rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    Y[(i / cols) * ldy + i % cols] = X[(i / cols) * ldx + i % cols];
}
__global__ void wrap_pos_kernel(float* x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = pymod1(x[i]);
}
__global__ void fill_kernel(float* p, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace mi

// ================================================================================================================================
// host objects
// ================================================================================================================================
struct GParam {
    std::string name;
    int64_t off, numel, toff;  // toff: offset of the transposed copy
    int rows, cols, ldt;
    bool derived = false;      // an inference-only tensor computed from the parameters at mi_gemnet_set_params; `off` then indexes mi_gemnet::dtheta
};

struct mi_gemnet {
    mi_gemnet_config cfg;
    std::vector<GParam> params;
    std::map<std::string, int> index;
    int64_t nparams = 0, ntrans = 0;
    const float* theta = nullptr;
    float* thetaT = nullptr;
    // derived tensors (entries of `params` behind the first n_real, not part of theta / the gradient / the public parameter list):
    //   out_blocks.<i>.q_F / q_S [emb_rbf, emb_edge]:  q[k][c] = out_weight[c] * rbf_weight[c][k].  The per-edge scalar heads are
    //   y[e] = sum_c x[e][c] (sum_k rbf[e][k] Wr[c][k]) w[c] = sum_k rbf[e][k] (x Q^T)[e][k]: a 16-wide product and a 16-term row dot
    //   instead of materialising the [E, emb_edge] radial weights and reading them back beside x.
    float* dtheta = nullptr;
    int64_t ndtheta = 0;
    int n_real = 0;
    const float* wptr(const GParam& p) const { return (p.derived ? dtheta : theta) + p.off; }
    unsigned* wamax = nullptr;   // [tensors] max |w| bit patterns (scales of the on-the-fly fp16 split)
    // plane sets of the weight blocks the edge-level dense layers use, built lazily after every mi_gemnet_set_params
    struct WPl {
        mi::u16* pl;
        float* rowsum;   // device: largest row sum of |W| of the block (output bounds)
        float scale;     // host: power-of-two scale of the block's plane set
        const mi::u16* frag = nullptr;   // the same block in MFMA fragment order (Planes::frag) where the register-tile GEMM can use it
        // the build is enqueued on the stream of the first use: a use on ANOTHER stream waits for `ready` until it has been seen complete
        hipEvent_t ready = nullptr;
        hipStream_t built_on = nullptr;
        bool done = false;
    };
    std::map<int64_t, WPl> wplanes;
    // Forwards of one network run from several host threads (MatterGenModule.sample(chains > 1): a chain per thread and stream), and plane
    // mode is decided per batch (E >= 4096, E rebuilt by every forward), so any forward may be the first use of a block: the map and the
    // two bump allocators below are guarded by this mutex, and cross-stream readers by the blocks' events.
    std::mutex wmu;
    std::vector<hipEvent_t> wevents;   // owned: destroyed when the plane sets are dropped
    mi::u16* warena = nullptr;
    size_t warena_elems = 0, warena_top = 0;
    float* wrowsum = nullptr;          // [1024] device floats handed out with the blocks
    int wrowsum_used = 0;
    std::vector<float> wscale_h;       // per tensor, from its absmax (host copy refreshed by set_params)
    // GemNet's ScalingFactor entries of the parameter list (`*.scale_factor`, 1 x 1, never trained): host copies refreshed by set_params.  A factor
    // equal to one costs nothing (op_scale returns its input); any other value runs as an elementwise pass and switches the sampler's forwards to
    // the synchronising form (sf_all_unit: the plane-only program has no such pass).  sf_pattern = which factors differ from one (program shape).
    std::vector<float> sf_h;           // per tensor (1.f for everything that is not a scale factor)
    bool sf_all_unit = true;
    uint64_t sf_pattern = 0;
    const GParam& P(const std::string& n) const {
        auto it = index.find(n);
        if (it == index.end()) {
            fprintf(stderr, "matinvent_hip: unknown gemnet parameter %s\n", n.c_str());
            abort();
        }
        return params[it->second];
    }
};

enum { OP_DENSE = 1, OP_MUL, OP_AXPBY, OP_SEGSUM, OP_TRIPLET, OP_ROWDOT, OP_EMBED, OP_FORCE, OP_STRESS, OP_SCALE };
enum { GK_NONE = 0, GK_SRC = 1, GK_DST = 2, GK_NODE = 3 };

struct GOp {
    int type = 0;
    const float *X = nullptr, *X2 = nullptr;  // inputs (arena or constant)
    float *Y = nullptr, *Z = nullptr;         // output, saved pre-activation
    int64_t M = 0;
    int N = 0, K = 0, ldy = 0;
    // dense
    int pidx = -1, wcol0 = 0, bidx = -1, act = 0;
    bool x_grad = true;
    const float *G1 = nullptr, *G2 = nullptr;
    int gk1 = 0, gk2 = 0;
    // axpby / rowdot
    float s = 1.f;
    bool perm = false;
    int widx = -1;
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, top = 0;
    float* take(size_t nfloats) {
        const size_t bytes = (nfloats * sizeof(float) + 255) / 256 * 256;
        float* p = reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(base) + top);
        top += bytes;
        return p;
    }
};

struct mi_gbatch {
    int B = 0, N = 0;
    int64_t E = 0, E_cap = 0;
    int deg_max = mi::GN_DEG;   // largest in-degree of the current graph (host copy: sizes the triplet kernels' LDS)
    // Forwards without a host round trip (the sampler's, see forward_impl): E is then the CAPACITY the launches are sized for and the
    // kernels read the edge count from meta[0]; crystals over a graph capacity are taken out on the device and remembered in bad[].
    bool nosync = false;
    int* bad = nullptr;            // [B] device: sticky per-crystal capacity flags (gg_fix_kernel)
    int* status_h = nullptr;       // pinned host mirror of bad[] (mi_gbatch_graph_status)
    // the activation arena of such forwards is COMPACT: an fp32 tensor that lives as a plane set only reserves no rows (ordinal of its
    // take call not in mat_ord, the set of tensors some consumer materialises); both found by the dry passes, cached per program shape
    bool compact = false, collect = false;
    int take_idx = 0;
    std::set<int> mat_ord;
    std::map<const float*, int> ord_of;
    std::set<const float*> placeholder;   // tensors of THIS forward that got a 1-float placeholder instead of rows (compact arena): materialising one is MI_ESTATE, never a write
    uint64_t dry_key = 0;
    size_t dry_need = 0;
    int64_t node_offset = 0, graph_offset = 0;
    std::vector<int> num_atoms_h, node_off_h;
    int *num_atoms = nullptr, *node_off = nullptr, *node2graph = nullptr;
    // graph
    int cap = 0, R_img = 0;
    int *ent = nullptr, *acnt = nullptr, *deg = nullptr, *mcount = nullptr, *meta = nullptr, *rowptr = nullptr, *src = nullptr, *dst = nullptr,
        *code = nullptr, *ekey = nullptr, *swap = nullptr, *edge_graph = nullptr;
    float *D = nullptr, *V = nullptr;
    // runtime
    Arena fwd, grad;
    float* scratch = nullptr;  // dZ buffer [E][max N] + reduction scratch
    size_t scratch_floats = 0, dz_floats = 0;
    mi::u16* dzpl = nullptr;   // dZ as a plane set (backward on the plane-set kernel), its absmax slot row and {scale, 1 / scale}
    size_t dzpl_elems = 0;
    unsigned* bwd_amax = nullptr;
    float* bwd_dsc = nullptr;
    std::vector<std::pair<size_t, size_t>> pl_ranges;   // (offset, bytes) of the plane sets in the activation arena of the last training forward
    std::vector<GOp> tape;
    bool tape_valid = false;
    std::map<std::string, std::pair<const float*, int64_t>> taps;
    float *out_pos = nullptr, *out_cell = nullptr, *out_logits = nullptr, *Fe = nullptr, *Se = nullptr;  // arena pointers of the last forward
    const int* types_in = nullptr;   // inputs of the last training forward (device copies)
    int* types_copy = nullptr;
    float *pos_copy = nullptr, *cell_copy = nullptr, *t_buf = nullptr;
    // sampler scratch
    float *sp_pos = nullptr, *sp_cell = nullptr, *sp_logits = nullptr, *ts_dev = nullptr;
    unsigned* amax_pool = nullptr;                  // [AMAX_SLOTS] absmax bit patterns of this forward's tensors
    std::map<const float*, unsigned*> amax_of;      // tensor -> its slot (several layers read the same tensor)
    int amax_used = 0;
    struct PlInfo {
        mi::u16* pl;
        float* dsc;     // device {scale, 1 / scale}, or NULL for a fixed scale
        float scale;
    };
    std::map<const float*, PlInfo> pl_of;           // fp32 tensor -> the plane set its producer wrote next to it
    struct Dims {
        int64_t rows;
        int cols;
    };
    std::map<const float*, Dims> absent;            // tensors of the last (inference) forward whose fp32 rows were NOT written: plane set only
    float* dsc_pool = nullptr;                      // [AMAX_SLOTS][2]
    int dsc_used = 0;
    bool planes_mode = false;
    std::vector<float> ts_h;
    std::vector<void*> allocs;
};

namespace mi {

template <typename T>
static int galloc(mi_gbatch* b, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        return MI_ENOMEM;
    }
    b->allocs.push_back(q);
    *p = (T*)q;
    return MI_OK;
}

static int arena_ensure(Arena& a, size_t bytes) {
    if (a.cap >= bytes) return MI_OK;
    if (a.base) (void)hipFree(a.base);
    a.base = nullptr;
    a.cap = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for the activation arena failed: %s", want, hipGetErrorString(e));
        return MI_ENOMEM;
    }
    a.base = (char*)q;
    a.cap = want;
    return MI_OK;
}

// ---- the program ---------------------------------------------------------------------------------------------------------------
struct Ctx {
    mi_gemnet* net;
    mi_gbatch* b;
    hipStream_t s;
    bool dry, train;
    int rc = MI_OK;
    float* take(size_t n) {
        ++b->take_idx;
        return b->fwd.take(n);
    }
    // the fp32 rows of an op's output.  plane_only: the producer writes the plane set alone (lean inference); in a compact arena such
    // a tensor gets a 256-byte placeholder (its address is still the key of the per-forward tables) unless a consumer materialises it
    float* take_y(size_t n, bool plane_only) {
        const int ord = b->take_idx;
        const bool ph = plane_only && b->compact && !b->collect && !b->mat_ord.count(ord);
        float* y = take(ph ? 1 : n);
        if (b->collect) b->ord_of[y] = ord;
        if (ph && !dry) b->placeholder.insert(y);
        return y;
    }
    // device-side row count of an edge-level launch (NULL: the host's count is exact)
    const int* mdev(int64_t M) const { return (b->nosync && M == b->E) ? b->meta : nullptr; }
    u16* take_planes(int64_t rows, int cols) {
        static const bool dbg_arena = getenv("MI_DEBUG_ARENA") != nullptr;
        if (dbg_arena) fprintf(stderr, "[arena %s] planes %lld x %d at %zu\n", dry ? "dry" : "run", (long long)rows, cols, b->fwd.top);
        u16* p = reinterpret_cast<u16*>(b->fwd.take((planes_elems(rows, cols) + 1) / 2));
        if (!dry && train) b->pl_ranges.emplace_back((size_t)(reinterpret_cast<char*>(p) - b->fwd.base), b->fwd.top - (size_t)(reinterpret_cast<char*>(p) - b->fwd.base));
        return p;
    }
    bool pm() const { return b->planes_mode; }
    // the exact absmax slot of a tensor: the producer's if it tracked one, otherwise one extra pass over the tensor (node-level tensors)
    unsigned* amax(const float* x, int64_t rows, int64_t cols) {   // (rows given explicitly: "edge-level" is rows == E, never inferred from the element count)
        const int64_t n = rows * cols;
        unsigned*& slot = b->amax_of[x];
        if (!slot && b->amax_used < AMAX_SLOTS) {
            need_f32(x);
            slot = b->amax_pool + (size_t)AMAX_W * b->amax_used++;
            const bool edge_rows = b->nosync && b->E > 0 && rows == b->E;
            if (!dry && n > 0)
                hipLaunchKernelGGL(absmax_bits_kernel<>, dim3((unsigned)std::min<int64_t>(1024, (n + 1023) / 1024)), dim3(256), 0, s, x, n, slot, AMAX_W - 1,
                                   edge_rows ? b->meta : (const int*)nullptr, edge_rows ? (int)cols : 0);
        }
        return slot;
    }
    unsigned* new_amax(const float* y) {   // a fresh (zeroed) slot the producer of y fills itself
        unsigned* slot = b->amax_used < AMAX_SLOTS ? b->amax_pool + (size_t)AMAX_W * b->amax_used++ : nullptr;
        if (slot) b->amax_of[y] = slot;
        return slot;
    }
    float* new_dsc() { return b->dsc_used < AMAX_SLOTS ? b->dsc_pool + 2 * b->dsc_used++ : nullptr; }
    // lean inference (see g_mg_lean): one format per edge-level tensor
    bool lean() const { return !train && b->planes_mode && (g_mg_lean & 1); }
    bool lean_fold() const { return lean() && (g_mg_lean & 2); }
    bool lean_segsum() const { return lean() && (g_mg_lean & 4); }
    bool lean_heads() const { return lean() && (g_mg_lean & 16); }
    bool lean_mul() const { return lean() && (g_mg_lean & 32); }
    void drop_f32(const float* Y, int64_t rows, int cols) { b->absent[Y] = mi_gbatch::Dims{rows, cols}; }
    bool is_absent(const float* X) const { return b->absent.find(X) != b->absent.end(); }
    Planes planes_of(const float* X, int cols) const {
        const mi_gbatch::PlInfo& pi = b->pl_of.at(X);
        return make_planes(pi.pl, cols, pi.scale, pi.dsc);
    }
    Src src(const float* X, int cols) const { return is_absent(X) ? Src{nullptr, planes_of(X, cols)} : Src{X, Planes()}; }
    // a consumer that reads fp32 rows: materialise them from the plane set if the producer skipped them
    void need_f32(const float* X);
    const int* gidx(int kind) const { return kind == GK_SRC ? b->src : kind == GK_DST ? b->dst : b->node2graph; }
};

static int materialize_f32(mi_gbatch* b, const float* X, hipStream_t s) {
    auto it = b->absent.find(X);
    if (it == b->absent.end()) return MI_OK;
    // (a tensor the cached dry passes left without rows: the program's shape changed behind the cache key -- fail, do not write rows x cols
    //  floats over the neighbouring tensors of the compact arena)
    MI_CHECK(!b->placeholder.count(X), MI_ESTATE, "MatterGen-shaped forward: a plane-only tensor without reserved rows is being materialised "
             "(a kernel-selection knob changed between two forwards of one batch handle without the dry passes being redone)");
    const mi_gbatch::Dims d = it->second;
    const mi_gbatch::PlInfo& pi = b->pl_of.at(X);
    if (d.rows > 0)
        hipLaunchKernelGGL(pl_to_f32_kernel, dim3((unsigned)std::min<int64_t>(4096, (d.rows * (d.cols / 2) + 255) / 256)), dim3(256), 0, s,
                           Src{nullptr, make_planes(pi.pl, d.cols, pi.scale, pi.dsc)}, const_cast<float*>(X), d.rows, d.cols,
                           (b->nosync && d.rows == b->E) ? b->meta : (const int*)nullptr);
    b->absent.erase(it);
    return MI_OK;
}
void Ctx::need_f32(const float* X) {
    if (!X || !is_absent(X)) return;
    if (dry) {
        if (b->collect) {
            auto it = b->ord_of.find(X);
            if (it != b->ord_of.end()) b->mat_ord.insert(it->second);
        }
        b->absent.erase(X);
    } else {
        const int r = materialize_f32(b, X, s);
        if (r != MI_OK && rc == MI_OK) rc = r;
    }
}

// MI_DEBUG_OPTIME=1: every op of the program is bracketed by stream synchronisations and its wall time printed (a per-layer profile
// with the layer's NAME, which a kernel trace does not carry); serialises the stream, never on in a timed run
struct OpTimer {
    Ctx& c;
    std::string what;
    bool on;
    std::chrono::steady_clock::time_point t0;
    OpTimer(Ctx& c_, const std::string& w, int64_t M, int N, int K) : c(c_), on(!c_.dry && g_optime) {
        if (!on) return;
        what = w + " [" + std::to_string(M) + " x " + std::to_string(N) + " x " + std::to_string(K) + "]";
        (void)hipStreamSynchronize(c.s);
        t0 = std::chrono::steady_clock::now();
    }
    ~OpTimer() {
        if (!on) return;
        (void)hipStreamSynchronize(c.s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[optime] %-52s %9.1f us\n", what.c_str(), us);
    }
};

#define CTX_OK(c) ((c).rc == MI_OK)
#define MI_HIP_VOID(call)                                                                   \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess) {                                                             \
            mi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            c.rc = MI_EHIP;                                                                 \
        }                                                                                   \
    } while (0)
#define CTX_TRY(c, expr)                      \
    do {                                      \
        if ((c).rc == MI_OK) (c).rc = (expr); \
    } while (0)

static inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

// (both under net->wmu)  publish: the block's build has just been enqueued on `s`.  join: a later use on stream `s`.
static int wpl_publish(mi_gemnet* net, mi_gemnet::WPl& e, hipStream_t s) {
    MI_HIP(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
    net->wevents.push_back(e.ready);
    MI_HIP(hipEventRecord(e.ready, s));
    e.built_on = s;
    return MI_OK;
}
static int wpl_join(mi_gemnet::WPl& e, hipStream_t s) {
    if (e.done || e.built_on == s) return MI_OK;   // (same stream: ordered behind the build anyway)
    if (hipEventQuery(e.ready) == hipSuccess) e.done = true;
    else MI_HIP(hipStreamWaitEvent(s, e.ready, 0));
    return MI_OK;
}

// plane set (+ largest |row| sum) of the weight block W[:, wcol0 : wcol0 + K], built on first use after mi_gemnet_set_params
static int get_wplanes(Ctx& c, int pidx, int wcol0, int K, mi_gemnet::WPl* out) {
    mi_gemnet* net = c.net;
    const int64_t key = ((int64_t)pidx << 40) | ((int64_t)wcol0 << 16) | (int64_t)K;
    std::lock_guard<std::mutex> guard(net->wmu);
    auto it = net->wplanes.find(key);
    if (it != net->wplanes.end()) {
        MI_TRY(wpl_join(it->second, c.s));
        *out = it->second;
        return MI_OK;
    }
    const GParam& w = net->params[pidx];
    if (!net->warena) {
        size_t tot = 0;
        // (+ the transposed blocks the backward's data-gradient products read: [cols, rows] per tensor, sliced by rows)
        // (+ the fragment-order copies of both for the register-tile GEMM: 2 N K elements each, + slack for the 16-byte alignment)
        for (const GParam& q : net->params)
            tot += planes_elems(q.rows, q.cols) + 3 * planes_elems(q.rows, 32) + planes_elems(q.cols, q.rows) + 3 * planes_elems(128, q.rows) + 2 * frag_elems(q.rows, q.cols) + 64;
        MI_HIP(hipMalloc((void**)&net->warena, tot * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&net->wrowsum, 1024 * sizeof(float)));
        net->warena_elems = tot;
    }
    const size_t need = planes_elems(w.rows, K);
    MI_CHECK(net->warena_top + need <= net->warena_elems && net->wrowsum_used < 1024, MI_ENOMEM, "weight plane arena exhausted");
    mi_gemnet::WPl e{net->warena + net->warena_top, net->wrowsum + net->wrowsum_used++, net->wscale_h[pidx]};
    net->warena_top += need;
    Planes P = make_planes(e.pl, K, e.scale);
    const int64_t nthr = (int64_t)((w.rows + 127) / 128 * 128) * P.KT * 16;
    hipLaunchKernelGGL(split_planes_kernel<>, dim3(nblk(nthr)), dim3(256), 0, c.s, net->wptr(w) + wcol0, w.cols, w.rows, K, P, 0);
    MI_HIP(hipMemsetAsync(e.rowsum, 0, sizeof(float), c.s));
    hipLaunchKernelGGL(rowsum_max_kernel, dim3((unsigned)std::min(64, (w.rows + 3) / 4)), dim3(256), 0, c.s, net->wptr(w) + wcol0, w.cols, w.rows, K, e.rowsum);
    MI_KERNEL_CHECK();
    if (MI_PLANES_FP16 && (w.rows & 255) == 0 && (K & 63) == 0 && K >= 128) {
        net->warena_top = (net->warena_top + 7) & ~(size_t)7;
        MI_CHECK(net->warena_top + frag_elems(w.rows, K) <= net->warena_elems, MI_ENOMEM, "weight plane arena exhausted");
        u16* f = net->warena + net->warena_top;
        net->warena_top += frag_elems(w.rows, K);
        MI_TRY(pack_frag_from_planes(P, w.rows, K, f, c.s));
        e.frag = f;
    }
    MI_TRY(wpl_publish(net, e, c.s));
    net->wplanes[key] = e;
    *out = e;
    return MI_OK;
}

// plane set of the TRANSPOSED weight block (W[:, wcol0 : wcol0 + K])^T = rows wcol0 .. wcol0 + K of thetaT's copy, [K, N]: the W operand of
// dX[M, K] += dZ[M, N] W[:, wcol0 : wcol0 + K]; built on first use after mi_gemnet_set_params (the forward has allocated the arena)
static int get_wtplanes(mi_gemnet* net, hipStream_t s, int pidx, int wcol0, int K, mi_gemnet::WPl* out) {
    const int64_t key = ((int64_t)1 << 62) | ((int64_t)pidx << 40) | ((int64_t)wcol0 << 16) | (int64_t)K;
    std::lock_guard<std::mutex> guard(net->wmu);
    auto it = net->wplanes.find(key);
    if (it != net->wplanes.end()) {
        MI_TRY(wpl_join(it->second, s));
        *out = it->second;
        return MI_OK;
    }
    const GParam& w = net->params[pidx];
    MI_CHECK(net->warena && net->thetaT && !w.derived, MI_ESTATE, "transposed weight planes before a plane-mode forward");
    const size_t need = planes_elems(K, w.rows);
    MI_CHECK(net->warena_top + need <= net->warena_elems, MI_ENOMEM, "weight plane arena exhausted");
    mi_gemnet::WPl e{net->warena + net->warena_top, nullptr, net->wscale_h[pidx]};
    net->warena_top += need;
    Planes P = make_planes(e.pl, w.rows, e.scale);
    const int64_t nthr = (int64_t)((K + 127) / 128 * 128) * P.KT * 16;
    hipLaunchKernelGGL(split_planes_kernel<>, dim3(nblk(nthr)), dim3(256), 0, s, net->thetaT + w.toff + (size_t)wcol0 * w.ldt, w.ldt, K, w.rows, P, 0);
    MI_KERNEL_CHECK();
    if (MI_PLANES_FP16 && (K & 255) == 0 && (w.rows & 63) == 0 && w.rows >= 128) {   // (this product's N is K, its contraction length w.rows)
        net->warena_top = (net->warena_top + 7) & ~(size_t)7;
        MI_CHECK(net->warena_top + frag_elems(K, w.rows) <= net->warena_elems, MI_ENOMEM, "weight plane arena exhausted");
        u16* f = net->warena + net->warena_top;
        net->warena_top += frag_elems(K, w.rows);
        MI_TRY(pack_frag_from_planes(P, K, w.rows, f, s));
        e.frag = f;
    }
    MI_TRY(wpl_publish(net, e, s));
    net->wplanes[key] = e;
    *out = e;
    return MI_OK;
}

// inference-only extensions of a dense layer's epilogue (the plane-set kernel's EXT instantiation):
//   y = ((act(z) * post_mul + res) * scale + res2[res2_rows]) * scale2
struct DenseExtra {
    const float* res2 = nullptr;      // second merge folded in behind the first (a skip connection closed by this layer)
    float scale2 = 1.f;
    const int* res2_rows = nullptr;   // row map of res2 (the edge <-> reversed-edge permutation)
    const float* post_mul = nullptr;  // [M, N] multiplicand applied right after the activation (radial weights)
};
// Y[M,N] = act( X[M,K] W[:, wcol0 : wcol0+K]^T + bias + G1[idx1] + G2[idx2] )
static float* op_dense(Ctx& c, const float* X, int64_t M, int K, const std::string& wname, int wcol0 = 0, int act = ACT_NONE, bool x_grad = true,
                       const std::string& bname = "", const float* G1 = nullptr, int gk1 = GK_NONE, const float* G2 = nullptr, int gk2 = GK_NONE,
                       int ldy = 0, const float* res = nullptr, float scale = 1.f, bool want_pl = false, const DenseExtra& ex = DenseExtra()) {
    const float* const res2 = ex.res2;
    const float scale2 = ex.scale2;
    const GParam& w = c.net->P(wname);
    const int N = w.rows;
    if (ldy == 0) ldy = N;
    OpTimer optimer(c, "dense " + wname, M, N, K);
    // edge-level layers whose input carries a plane set run on the pre-split plane kernel
    auto xin = c.b->pl_of.find(X);
    const bool planes = c.pm() && M >= MG_PLANES_MIN_ROWS && (N & 7) == 0 && ldy == N && bname.empty() && xin != c.b->pl_of.end();
    float* Y = c.take_y((size_t)M * ldy, planes && want_pl && c.lean());
    float* Z = (act != ACT_NONE && c.train) ? c.take((size_t)M * N) : nullptr;
    u16* Ypl = (planes && want_pl) ? c.take_planes(M, N) : nullptr;
    if (Ypl) c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, nullptr, 1.f};   // (registered in the dry run too: both runs must take the same decisions)
    const bool lean_y = Ypl && c.lean();   // the consumers of Y read the plane set: its fp32 rows are not written
    if (!planes) {   // the fp32-operand kernels read rows
        c.need_f32(X);
        c.need_f32(res);
        c.need_f32(res2);
    }
    if (ex.res2_rows) c.need_f32(res2);   // (the row-mapped form reads fp32 rows)
    c.need_f32(ex.post_mul);
    if (lean_y) c.drop_f32(Y, M, N);
    if (c.dry || !CTX_OK(c) || M == 0) return Y;
    if (planes) {
        const int pidx = c.net->index.at(wname);
        mi_gemnet::WPl wp;
        CTX_TRY(c, get_wplanes(c, pidx, wcol0, K, &wp));
        if (!CTX_OK(c)) return Y;
        const mi_gbatch::PlInfo& xi = xin->second;
        PlanesEpilogue pe;
        if (G1) {
            pe.ep.row_bias = G1;
            pe.ep.row_group = c.gidx(gk1);
            pe.ep.ld_row_bias = N;
        }
        if (G2) {
            pe.ep.row_bias2 = G2;
            pe.ep.row_group2 = c.gidx(gk2);
            pe.ep.ld_row_bias2 = N;
        }
        pe.ep.act = act;
        if (Z) {
            pe.ep.pre_act = Z;
            pe.ep.ld_pre = N;
        }
        if (res) {
            if (!(g_mg_lean & 8)) c.need_f32(res);
            if (c.is_absent(res)) pe.res_pl = c.planes_of(res, N);
            else {
                pe.ep.residual = res;
                pe.ep.ld_res = N;
            }
        }
        pe.ep.out_scale = scale;
        if (res2) {
            if (c.is_absent(res2)) pe.res2_pl = c.planes_of(res2, N);
            else {
                pe.residual2 = res2;
                pe.ld_res2 = N;
                pe.res2_rows = ex.res2_rows;
            }
            pe.out_scale2 = scale2;
        }
        if (ex.post_mul) {
            pe.post_mul = ex.post_mul;
            pe.ld_post_mul = N;
        }
        pe.C = lean_y ? nullptr : Y;
        pe.ldc = N;
        pe.absmax = c.new_amax(Y);
        pe.absmax_mask = AMAX_W - 1;
        if (Ypl) {   // scale of the output plane set from the one-layer bound on the exact absmax of everything that enters
            float* dsc = c.new_dsc();
            const int rows_g = gk1 == GK_NODE ? c.b->B : c.b->N;
            hipLaunchKernelGGL(mg_scale_kernel, dim3(1), dim3(AMAX_W), 0, c.s, c.amax(X, M, K), wp.rowsum, (const unsigned*)nullptr, (const int*)nullptr, 1.f,
                               G1 ? c.amax(G1, rows_g, N) : (const unsigned*)nullptr, G2 ? c.amax(G2, c.b->N, N) : (const unsigned*)nullptr,
                               res ? c.amax(res, M, N) : (const unsigned*)nullptr, act == ACT_SSILU ? GN_ACT : 1.f, scale, dsc,
                               res2 ? c.amax(res2, M, N) : (const unsigned*)nullptr, scale2, ex.post_mul ? c.amax(ex.post_mul, M, N) : (const unsigned*)nullptr);
            if (N % 32 != 0) MI_HIP_VOID(hipMemsetAsync(Ypl, 0, planes_elems(M, N) * sizeof(u16), c.s));   // the k-padding of the next product must be zero
            pe.Cp = make_planes(Ypl, N, 1.f, dsc);
            c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, dsc, 1.f};
        }
        Planes Wp = make_planes(wp.pl, K, wp.scale);
        Wp.frag = wp.frag;
        pe.m_dev = c.mdev(M);
        CTX_TRY(c, gemm_planes(make_planes(xi.pl, K, xi.scale, xi.dsc), Wp, (int)M, N, K, pe, c.s));
        if (c.train) {
            GOp o;
            o.type = OP_DENSE;
            o.X = X;
            o.Y = Y;
            o.Z = Z;
            o.M = M;
            o.N = N;
            o.K = K;
            o.ldy = ldy;
            o.pidx = pidx;
            o.wcol0 = wcol0;
            o.act = act;
            o.x_grad = x_grad;
            o.G1 = G1;
            o.G2 = G2;
            o.gk1 = gk1;
            o.gk2 = gk2;
            o.X2 = res;
            o.s = scale;
            c.b->tape.push_back(o);
        }
        return Y;
    }
    if (c.mdev(M)) {   // (every edge-level layer of a lean plane-mode forward takes the branch above)
        set_error("forward without a host round trip: edge-level layer %s is not on the plane-set kernel", wname.c_str());
        c.rc = MI_ESTATE;
        return Y;
    }
    GemmEpilogue ep;
    if (!bname.empty()) ep.bias = c.net->theta + c.net->P(bname).off;
    if (G1) {
        ep.row_bias = G1;
        ep.row_group = c.gidx(gk1);
        ep.ld_row_bias = N;
    }
    if (G2) {
        ep.row_bias2 = G2;
        ep.row_group2 = c.gidx(gk2);
        ep.ld_row_bias2 = N;
    }
    ep.act = act;
    if (Z) {
        ep.pre_act = Z;
        ep.ld_pre = N;
    }
    if (res) {   // y = (act(z) + res) * scale: the residual merge folded into the product's epilogue (one pass over [M, N] less)
        ep.residual = res;
        ep.ld_res = N;
    }
    ep.out_scale = scale;
    if (ldy != N) MI_HIP_VOID(hipMemsetAsync(Y, 0, (size_t)M * ldy * sizeof(float), c.s));
    const bool use16 = g_mg_f16 && g_gemm_mode != 0 && (int64_t)cdiv(M, 128) * cdiv(N, 128) >= 256 && c.b->amax_used < AMAX_SLOTS;
    if (use16) {
        unsigned*& slot = c.b->amax_of[X];
        if (!slot) {   // the operand's exact absmax, once per tensor and forward (one extra read of it; the product then issues half the MFMA work)
            slot = c.b->amax_pool + (size_t)AMAX_W * c.b->amax_used++;
            hipLaunchKernelGGL(absmax_bits_kernel<>, dim3((unsigned)std::min<int64_t>(2048, cdiv(M * K, 1024))), dim3(256), 0, c.s, X, M * K, slot, AMAX_W - 1);
        }
        hipLaunchKernelGGL(amax_fold_kernel, dim3(1), dim3(AMAX_W), 0, c.s, slot);   // (the kernel reads one word)
        CTX_TRY(c, gemm_nt_split(X, K, c.net->wptr(w) + wcol0, w.cols, Y, ldy, (int)M, N, K, ep, c.s, nullptr, slot, c.net->wamax + c.net->index.at(wname)));
    } else {
        CTX_TRY(c, gemm_nt(X, K, c.net->wptr(w) + wcol0, w.cols, Y, ldy, (int)M, N, K, ep, c.s));
    }
    if (res2) {   // (the plane-set kernel folds this in; here it is one more pass, in place)
        if (ldy != N || c.train || ex.res2_rows || ex.post_mul) c.rc = MI_EINVAL;
        else hipLaunchKernelGGL(axpby_fwd_kernel, dim3(nblk(M * N)), dim3(256), 0, c.s, Y, res2, (const int*)nullptr, scale2, Y, M, N);
    }
    if (c.train) {
        GOp o;
        o.type = OP_DENSE;
        o.X = X;
        o.Y = Y;
        o.Z = Z;
        o.M = M;
        o.N = N;
        o.K = K;
        o.ldy = ldy;
        o.pidx = c.net->index.at(wname);
        o.wcol0 = wcol0;
        o.bidx = bname.empty() ? -1 : c.net->index.at(bname);
        o.act = act;
        o.x_grad = x_grad;
        o.G1 = G1;
        o.G2 = G2;
        o.gk1 = gk1;
        o.gk2 = gk2;
        o.X2 = res;
        o.s = scale;
        c.b->tape.push_back(o);
    }
    return Y;
}
static float* op_mul(Ctx& c, const float* A, const float* Bm, int64_t M, int N, bool want_pl = false) {
    OpTimer optimer(c, "mul", M, N, 0);
    const bool pl = c.pm() && want_pl && M >= MG_PLANES_MIN_ROWS && (N & 1) == 0;
    float* Y = c.take_y((size_t)M * N, pl && c.lean());
    u16* Ypl = pl ? c.take_planes(M, N) : nullptr;
    if (Ypl) c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, nullptr, 1.f};
    const bool lean_y = Ypl && c.lean();
    if (!pl) {
        c.need_f32(A);
        c.need_f32(Bm);
    }
    const Src sa = c.src(A, N), sb = c.src(Bm, N);
    if (lean_y) c.drop_f32(Y, M, N);
    if (c.dry || !CTX_OK(c) || M == 0) return Y;
    if (pl) {
        float* dsc = c.new_dsc();
        hipLaunchKernelGGL(mg_scale_kernel, dim3(1), dim3(AMAX_W), 0, c.s, c.amax(A, M, N), (const float*)nullptr, c.amax(Bm, M, N), (const int*)nullptr, 1.f,
                           (const unsigned*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr, 1.f, 1.f, dsc);
        if (N % 32 != 0) MI_HIP_VOID(hipMemsetAsync(Ypl, 0, planes_elems(M, N) * sizeof(u16), c.s));
        c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, dsc, 1.f};
        hipLaunchKernelGGL(mul_pl_kernel, dim3((unsigned)std::min<int64_t>(EW_GRID, nblk(M * (N / 2)))), dim3(256), 0, c.s, sa, sb, lean_y ? (float*)nullptr : Y,
                           make_planes(Ypl, N, 1.f, dsc), c.new_amax(Y), M, N, c.mdev(M));
    } else if (c.mdev(M)) {
        c.rc = MI_ESTATE;
    } else
    hipLaunchKernelGGL(mul_fwd_kernel, dim3(nblk(M * N)), dim3(256), 0, c.s, A, Bm, Y, M * N);
    if (c.train) {
        GOp o;
        o.type = OP_MUL;
        o.X = A;
        o.X2 = Bm;
        o.Y = Y;
        o.M = M;
        o.N = N;
        c.b->tape.push_back(o);
    }
    return Y;
}
static float* op_axpby(Ctx& c, const float* A, const float* Bm, int64_t M, int N, bool perm = false, bool want_pl = false) {
    OpTimer optimer(c, perm ? "axpby (row-permuted)" : "axpby", M, N, 0);
    const bool big = c.pm() && M >= MG_PLANES_MIN_ROWS && (N & 1) == 0;   // edge-level: the output's absmax is tracked on the way
    float* Y = c.take_y((size_t)M * N, big && want_pl && c.lean());
    u16* Ypl = (big && want_pl) ? c.take_planes(M, N) : nullptr;
    if (Ypl) c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, nullptr, 1.f};
    const bool lean_y = Ypl && c.lean();
    if (!big) {
        c.need_f32(A);
        c.need_f32(Bm);
    }
    const Src sa = c.src(A, N), sb = c.src(Bm, N);
    if (lean_y) c.drop_f32(Y, M, N);
    if (c.dry || !CTX_OK(c) || M == 0) return Y;
    if (big) {
        float* dsc = nullptr;
        if (Ypl) {
            dsc = c.new_dsc();
            hipLaunchKernelGGL(mg_scale_kernel, dim3(1), dim3(AMAX_W), 0, c.s, c.amax(A, M, N), (const float*)nullptr, (const unsigned*)nullptr, (const int*)nullptr, 1.f,
                               (const unsigned*)nullptr, (const unsigned*)nullptr, c.amax(Bm, M, N), 1.f, GN_ISQ2, dsc);
            if (N % 32 != 0) MI_HIP_VOID(hipMemsetAsync(Ypl, 0, planes_elems(M, N) * sizeof(u16), c.s));
            c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, dsc, 1.f};
        }
        hipLaunchKernelGGL(axpby_pl_kernel, dim3((unsigned)std::min<int64_t>(EW_GRID, nblk(M * (N / 2)))), dim3(256), 0, c.s, sa, sb, perm ? c.b->swap : (const int*)nullptr, GN_ISQ2,
                           lean_y ? (float*)nullptr : Y, Ypl ? make_planes(Ypl, N, 1.f, dsc) : Planes(), c.new_amax(Y), M, N, c.mdev(M));
    } else if (c.mdev(M)) {
        c.rc = MI_ESTATE;
    } else
    hipLaunchKernelGGL(axpby_fwd_kernel, dim3(nblk(M * N)), dim3(256), 0, c.s, A, Bm, perm ? c.b->swap : (const int*)nullptr, GN_ISQ2, Y, M, N);
    if (c.train) {
        GOp o;
        o.type = OP_AXPBY;
        o.X = A;
        o.X2 = Bm;
        o.Y = Y;
        o.M = M;
        o.N = N;
        o.s = GN_ISQ2;
        o.perm = perm;
        c.b->tape.push_back(o);
    }
    return Y;
}
// GemNet's ScalingFactor `name`.scale_factor applied to X [M, N]: the input itself while the factor is one (the identity-initialised case: no launch,
// no tape entry), otherwise one elementwise pass on fp32 rows (its result carries no plane set: the consumer splits on the fly)
static float* op_scale(Ctx& c, float* X, int64_t M, int N, const std::string& name) {
    const int idx = c.net->index.at(name + ".scale_factor");
    if (c.net->sf_h.empty() || c.net->sf_h[idx] == 1.f) return X;
    OpTimer optimer(c, "scale " + name, M, N, 0);
    float* Y = c.take((size_t)M * N);
    c.need_f32(X);
    if (c.dry || !CTX_OK(c) || M == 0) return Y;
    if (c.mdev(M)) {   // (forward_impl keeps such networks off the no-round-trip form)
        c.rc = MI_ESTATE;
        return Y;
    }
    const float* sf = c.net->theta + c.net->params[idx].off;
    hipLaunchKernelGGL(scale_fwd_kernel, dim3(nblk(M * N)), dim3(256), 0, c.s, X, sf, Y, M * N);
    if (c.train) {
        GOp o;
        o.type = OP_SCALE;
        o.X = X;
        o.Y = Y;
        o.M = M;
        o.N = N;
        o.widx = idx;
        c.b->tape.push_back(o);
    }
    return Y;
}
// edges -> atoms by target with the per-edge weights Wt multiplied in on the way (inference: the weighted messages are never written)
static float* op_segsum_mul(Ctx& c, const float* X, const float* Wt, int cols) {
    OpTimer optimer(c, "segsum_mul", c.b->E, cols, 0);
    float* Y = c.take((size_t)c.b->N * cols);
    c.need_f32(Wt);
    const Src sx = c.src(X, cols);
    if (c.dry || !CTX_OK(c) || c.train) return Y;
    hipLaunchKernelGGL(segsum_mul_kernel, dim3(c.b->N), dim3(256), 0, c.s, sx, Wt, c.b->rowptr, Y, cols);
    return Y;
}
static float* op_segsum(Ctx& c, const float* X, int cols) {  // edges -> atoms by target
    OpTimer optimer(c, "segsum", c.b->E, cols, 0);
    float* Y = c.take((size_t)c.b->N * cols);
    c.need_f32(X);
    if (c.dry || !CTX_OK(c)) return Y;
    hipLaunchKernelGGL(segsum_kernel, dim3(nblk((int64_t)c.b->N * cols)), dim3(256), 0, c.s, X, cols, c.b->rowptr, (const int*)nullptr, Y, c.b->N, cols, 0);
    if (c.train) {
        GOp o;
        o.type = OP_SEGSUM;
        o.X = X;
        o.Y = Y;
        o.M = c.b->E;
        o.N = cols;
        c.b->tape.push_back(o);
    }
    return Y;
}
// dynamic-LDS ceiling of a kernel: only ever RAISED, under a lock, and remembered per function -- never set per launch (see op_triplet)
static void raise_lds_ceiling(const void* fn, int bytes) {
    static std::mutex mu;
    static std::map<const void*, int> cur;
    std::lock_guard<std::mutex> g(mu);
    int& c = cur[fn];
    if (bytes > c && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) c = bytes;
}

static float* op_triplet(Ctx& c, const float* xd, const float* cbfW) {
    OpTimer optimer(c, "triplet", c.b->E, 0, 0);
    const mi_gemnet_config& g = c.net->cfg;
    const int64_t E = c.b->E;
    const int NT = g.emb_cbf * g.emb_trip;
    const bool pl = c.pm() && E >= MG_PLANES_MIN_ROWS && (g.emb_trip & 1) == 0;
    float* Y = c.take_y((size_t)E * g.emb_cbf * g.emb_trip, pl && !c.train);
    u16* Ypl = pl ? c.take_planes(E, NT) : nullptr;
    if (Ypl) c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, nullptr, 1.f};
    c.need_f32(xd);
    c.need_f32(cbfW);
    if (c.dry || !CTX_OK(c) || E == 0) return Y;
    Planes P;
    unsigned* ymax = nullptr;
    if (pl) {   // |Tm| <= max|cbfW| max|xd| S max|Y_l| deg_max
        float* dsc = c.new_dsc();
        hipLaunchKernelGGL(mg_scale_kernel, dim3(1), dim3(AMAX_W), 0, c.s, c.amax(cbfW, E, (int64_t)g.num_spherical * g.emb_cbf), (const float*)nullptr, c.amax(xd, E, g.emb_trip),
                           (const int*)(c.b->meta + 1), 1.1f * (float)g.num_spherical, (const unsigned*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr,
                           1.f, 1.f, dsc);
        if (NT % 32 != 0) MI_HIP_VOID(hipMemsetAsync(Ypl, 0, planes_elems(E, NT) * sizeof(u16), c.s));
        P = make_planes(Ypl, NT, 1.f, dsc);
        ymax = c.new_amax(Y);
        c.b->pl_of[Y] = mi_gbatch::PlInfo{Ypl, dsc, 1.f};
    }
    const int DG = std::min(GN_DEG, (c.b->deg_max + 3) / 4 * 4);
    const size_t sh = (size_t)(TFC * DG * 8 + DG * 3 + DG * g.emb_trip + TFC * g.num_spherical * g.emb_cbf) * sizeof(float);
    // (the kernel's dynamic-LDS ceiling is raised once per network shape, to what the largest in-degree needs: forwards of one network run
    //  from several host threads -- concurrent sampler chains -- and a per-launch hipFuncSetAttribute with THIS launch's size raced with the
    //  other threads' launches of the same function)
    const size_t sh_max = (size_t)(TFC * GN_DEG * 8 + GN_DEG * 3 + GN_DEG * g.emb_trip + TFC * g.num_spherical * g.emb_cbf) * sizeof(float);
#define TRIP_FWD(SS)                                                                                                                       \
    case SS:                                                                                                                               \
        raise_lds_ceiling(reinterpret_cast<const void*>(&triplet_fwd_kernel<SS>), (int)sh_max);                                           \
        hipLaunchKernelGGL((triplet_fwd_kernel<SS>), dim3(c.b->N), dim3(512), sh, c.s, xd, c.b->V, cbfW, c.b->rowptr, (pl && !c.train) ? (float*)nullptr : Y, g.emb_trip, g.emb_cbf, P, ymax, DG); \
        break;
    switch (g.num_spherical) {
        TRIP_FWD(1) TRIP_FWD(2) TRIP_FWD(3) TRIP_FWD(4) TRIP_FWD(5) TRIP_FWD(6) TRIP_FWD(7) TRIP_FWD(8)
        default: c.rc = MI_EINVAL;
    }
#undef TRIP_FWD
    if (c.train) {
        GOp o;
        o.type = OP_TRIPLET;
        o.X = xd;
        o.X2 = cbfW;
        o.Y = Y;
        o.M = E;
        c.b->tape.push_back(o);
    }
    return Y;
}
static void op_rowdot(Ctx& c, const float* A, const float* Bm, const std::string& wname, float* y, int K, bool acc) {
    OpTimer optimer(c, "rowdot " + wname, c.b->E, 1, K);
    c.need_f32(A);
    c.need_f32(Bm);
    if (c.dry || !CTX_OK(c) || c.b->E == 0) return;
    if (c.mdev(c.b->E)) {
        c.rc = MI_ESTATE;
        return;
    }
    const GParam& w = c.net->P(wname);
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(nblk(c.b->E, 4)), dim3(256), 0, c.s, A, Bm, c.net->theta + w.off, y, c.b->E, K, acc ? 1 : 0);
    if (c.train) {
        GOp o;
        o.type = OP_ROWDOT;
        o.X = A;
        o.X2 = Bm;
        o.Y = y;
        o.M = c.b->E;
        o.K = K;
        o.widx = c.net->index.at(wname);
        c.b->tape.push_back(o);
    }
}

// `out_pl`: the stack's result feeds another dense layer (its plane set is wanted)
static void op_rowdot_short(Ctx& c, const float* A, const float* Bm, float* y, int K, bool acc) {
    OpTimer optimer(c, "rowdot (short rows)", c.b->E, 1, K);
    c.need_f32(A);
    c.need_f32(Bm);
    if (c.dry || !CTX_OK(c) || c.b->E == 0 || c.train) return;
    hipLaunchKernelGGL(rowdot_short_kernel, dim3(nblk(c.b->E * K)), dim3(256), 0, c.s, A, Bm, y, c.b->E, K, acc ? 1 : 0, c.mdev(c.b->E));
}

// `outer` (lean inference, n > 0): the skip connection the stack closes, (outer + stack(x)) / sqrt(2), folded into its last layer
static float* res_stack(Ctx& c, const std::string& prefix, int n, float* x, int64_t M, int W, bool out_pl = false, const float* outer = nullptr) {
    for (int k = 0; k < n; ++k) {
        const std::string p = prefix + "." + std::to_string(k);
        float* y1 = op_dense(c, x, M, W, p + ".0.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true);
        x = op_dense(c, y1, M, W, p + ".1.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, x, GN_ISQ2,
                     k + 1 < n || out_pl, DenseExtra{k + 1 == n ? outer : nullptr, GN_ISQ2, nullptr, nullptr});  // (x + f(x)) / sqrt(2)
    }
    return x;
}

static void out_block(Ctx& c, int i, const float* m, const float* rbf_out, bool first) {
    const mi_gemnet_config& g = c.net->cfg;
    const int64_t E = c.b->E;
    const int Ed = g.emb_edge;
    const std::string p = "out_blocks." + std::to_string(i);
    float* t1 = op_dense(c, m, E, Ed, p + ".dense_F.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true);
    const int Rb = g.emb_rbf;
    // (the energy path -- dense_rbf, scale_sum, seq_energy, out_energy -- is not evaluated: E_t feeds no output, see the parameter list)
    if (c.lean_heads() && (Rb & (Rb - 1)) == 0 && Rb <= 64) {   // (see mi_gemnet::dtheta: the heads through the derived [emb_rbf, emb_edge] tensors)
        float* xF = res_stack(c, p + ".res_F", g.num_atom, t1, E, Ed, true);
        xF = op_scale(c, xF, E, Ed, p + ".scale_rbf_F");
        float* pF = op_dense(c, xF, E, Ed, p + ".q_F");
        op_rowdot_short(c, rbf_out, pF, c.b->Fe, Rb, !first);
        float* xS = op_dense(c, m, E, Ed, p + ".dense_S.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true);
        float* pS = op_dense(c, xS, E, Ed, p + ".q_S");
        op_rowdot_short(c, rbf_out, pS, c.b->Se, Rb, !first);
        return;
    }
    float* xF = res_stack(c, p + ".res_F", g.num_atom, t1, E, Ed);
    xF = op_scale(c, xF, E, Ed, p + ".scale_rbf_F");
    float* rF = op_dense(c, rbf_out, E, g.emb_rbf, p + ".rbf_F.weight");
    op_rowdot(c, xF, rF, p + ".out_F.weight", c.b->Fe, Ed, !first);
    float* xS = op_dense(c, m, E, Ed, p + ".dense_S.weight", 0, ACT_SSILU);
    float* rS = op_dense(c, rbf_out, E, g.emb_rbf, p + ".rbf_S.weight");
    op_rowdot(c, xS, rS, p + ".out_S.weight", c.b->Se, Ed, !first);
}

// the whole denoiser; pos / cell / types / t are the (device) inputs
static void run_program(Ctx& c, const float* pos, const float* cell, const int* types, const float* t) {
    const mi_gemnet_config& g = c.net->cfg;
    mi_gbatch* b = c.b;
    const int A = g.emb_atom, Ed = g.emb_edge, R = g.num_radial, Rb = g.emb_rbf, S = g.num_spherical, Cb = g.emb_cbf, Tr = g.emb_trip;
    const int64_t E = b->E;
    const int N = b->N, B = b->B;
    b->fwd.top = 0;
    b->take_idx = 0;
    b->tape.clear();
    b->taps.clear();
    b->amax_of.clear();
    b->amax_used = 0;
    b->pl_of.clear();
    b->absent.clear();
    b->placeholder.clear();
    b->dsc_used = 0;
    b->planes_mode = g_mg_planes && g_gemm_mode != 0 && E >= MG_PLANES_MIN_ROWS;
    if (!c.dry && CTX_OK(c)) MI_HIP_VOID(hipMemsetAsync(b->amax_pool, 0, (size_t)AMAX_SLOTS * AMAX_W * sizeof(unsigned), c.s));
    float* rbf = c.take((size_t)E * R);
    u16* rbf_pl = c.pm() ? c.take_planes(E, R) : nullptr;
    if (rbf_pl) b->pl_of[rbf] = mi_gbatch::PlInfo{rbf_pl, nullptr, PL_S_UNIT};   // |rbf| <= 1: the fixed unit-range scale
    float* z = c.take((size_t)B * A);
    float* H0 = c.take((size_t)N * A);
    b->Fe = c.take((size_t)E);
    b->Se = c.take((size_t)E);
    b->out_pos = c.take((size_t)N * 3);
    b->out_cell = c.take((size_t)B * 9);
    if (!c.dry && CTX_OK(c)) {
        if (E > 0)
            hipLaunchKernelGGL(edge_geom_rbf_kernel, dim3(nblk(E * R)), dim3(256), 0, c.s, pos, cell, b->src, b->dst, b->code, b->edge_graph, g.max_images,
                               g.cutoff, R, E, b->D, b->V, rbf, c.mdev(E));
        if (rbf_pl && E > 0) {
            const Planes P = make_planes(rbf_pl, R, PL_S_UNIT);
            hipLaunchKernelGGL(split_fixed_kernel, dim3(nblk(E * P.KT * 16)), dim3(256), 0, c.s, rbf, P, E, R, c.mdev(E));
            unsigned* one = c.new_amax(rbf);   // bound 1 (envelope x Gaussian)
            if (one) MI_HIP_VOID(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(one), 0x3f800000, 1, c.s));
        }
        hipLaunchKernelGGL(nle_kernel, dim3(nblk((int64_t)B * A)), dim3(256), 0, c.s, t, z, B, A);
        hipLaunchKernelGGL(embed_fwd_kernel, dim3(nblk((int64_t)N * A)), dim3(256), 0, c.s, c.net->theta + c.net->P("atom_emb.weight").off, types, H0, N, A);
        if (c.train) {
            GOp o;
            o.type = OP_EMBED;
            o.Y = H0;
            o.M = N;
            o.N = A;
            b->tape.push_back(o);
        }
    }
    float* ZP = op_dense(c, z, B, A, "atom_latent_emb.weight", A, ACT_NONE, false);
    float* h = op_dense(c, H0, N, A, "atom_latent_emb.weight", 0, ACT_NONE, true, "atom_latent_emb.bias", ZP, GK_NODE);
    float* HS = op_dense(c, h, N, A, "edge_emb.weight", 0);
    float* HT = op_dense(c, h, N, A, "edge_emb.weight", A);
    float* m = op_dense(c, rbf, E, R, "edge_emb.weight", 2 * A, ACT_SSILU, false, "", HS, GK_SRC, HT, GK_DST, 0, nullptr, 1.f, true);
    float* rbf3 = op_dense(c, rbf, E, R, "mlp_rbf3.weight", 0, ACT_NONE, false, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true);
    float* cbfW = op_dense(c, rbf, E, R, "mlp_cbf3.weight", 0, ACT_NONE, false);
    float* rbf_h = op_dense(c, rbf, E, R, "mlp_rbf_h.weight", 0, ACT_NONE, false, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true);
    const bool heads_short = c.lean_heads() && (Rb & (Rb - 1)) == 0 && Rb <= 64;   // (out_block: rbf_out then feeds a row dot, not a dense layer)
    float* rbf_out = op_dense(c, rbf, E, R, "mlp_rbf_out.weight", 0, ACT_NONE, false, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, !heads_short);
    b->taps["rbf"] = {rbf, E * R};
    b->taps["h0"] = {h, (int64_t)N * A};
    b->taps["m0"] = {m, E * Ed};
    out_block(c, 0, m, rbf_out, true);
    for (int i = 0; i < g.num_blocks; ++i) {
        const std::string p = "int_blocks." + std::to_string(i);
        float* rr = op_dense(c, rbf3, E, Rb, p + ".mlp_rbf.weight");
        float* x_ba;
        if (c.lean_mul()) {   // the radial weighting as the multiplicand of dense_ba's epilogue
            x_ba = op_dense(c, m, E, Ed, p + ".dense_ba.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true,
                            DenseExtra{nullptr, 1.f, nullptr, rr});
        } else {
            float* tb = op_dense(c, m, E, Ed, p + ".dense_ba.weight", 0, ACT_SSILU);
            x_ba = op_mul(c, tb, rr, E, Ed, true);
        }
        x_ba = op_scale(c, x_ba, E, Ed, p + ".scale_rbf");
        float* xd = op_dense(c, x_ba, E, Ed, p + ".down_projection.weight");
        float* Tm = op_triplet(c, xd, cbfW);
        float* x3 = op_dense(c, Tm, E, Cb * Tr, p + ".bilinear.weight", 0, ACT_NONE, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, 1.f, true);
        x3 = op_scale(c, x3, E, g.emb_bil, p + ".scale_cbf_sum");
        b->taps["x3_" + std::to_string(i)] = {x3, E * g.emb_bil};
        float* x;
        if (c.lean_mul()) {
            // x = (act(m W_ca) + (u1 + u2[swap]) / sqrt(2)) / sqrt(2): both up-projections leave pre-scaled, the edge <-> reversed-edge
            // merge and the skip merge ride in dense_ca's epilogue (second residual through the row map)
            float* u1 = op_dense(c, x3, E, g.emb_bil, p + ".up_projection_ca.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, GN_ISQ2);
            float* u2 = op_dense(c, x3, E, g.emb_bil, p + ".up_projection_ac.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, nullptr, GN_ISQ2);
            x = op_dense(c, m, E, Ed, p + ".dense_ca.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, u1, 1.f, g.num_before_skip > 0,
                         DenseExtra{u2, GN_ISQ2, c.b->swap, nullptr});
        } else {
            float* u1 = op_dense(c, x3, E, g.emb_bil, p + ".up_projection_ca.weight", 0, ACT_SSILU);
            float* u2 = op_dense(c, x3, E, g.emb_bil, p + ".up_projection_ac.weight", 0, ACT_SSILU);
            float* x3b = op_axpby(c, u1, u2, E, Ed, true);
            x = op_dense(c, m, E, Ed, p + ".dense_ca.weight", 0, ACT_SSILU, true, "", nullptr, GK_NONE, nullptr, GK_NONE, 0, x3b, GN_ISQ2,
                         g.num_before_skip > 0);  // (x_ca + x3) / sqrt(2)
        }
        if (c.lean_fold() && g.num_before_skip > 0) m = res_stack(c, p + ".before_skip", g.num_before_skip, x, E, Ed, true, m);
        else {
            x = res_stack(c, p + ".before_skip", g.num_before_skip, x, E, Ed);
            m = op_axpby(c, m, x, E, Ed, false, true);
        }
        m = res_stack(c, p + ".after_skip", g.num_after_skip, m, E, Ed, true);
        float* ru = op_dense(c, rbf_h, E, Rb, p + ".atom_update.rbf.weight");
        float* h2;
        if (c.lean_segsum()) h2 = op_segsum_mul(c, m, ru, Ed);
        else {
            float* mm = op_mul(c, m, ru, E, Ed);
            h2 = op_segsum(c, mm, Ed);
        }
        h2 = op_scale(c, h2, N, Ed, p + ".atom_update.scale_sum");
        h2 = op_dense(c, h2, N, Ed, p + ".atom_update.dense.weight", 0, ACT_SSILU);
        h2 = res_stack(c, p + ".atom_update.res", g.num_atom, h2, N, A);
        h = op_axpby(c, h, h2, N, A);
        HS = op_dense(c, h, N, A, p + ".concat.weight", 0);
        HT = op_dense(c, h, N, A, p + ".concat.weight", A);
        float* m2 = op_dense(c, m, E, Ed, p + ".concat.weight", 2 * A, ACT_SSILU, true, "", HS, GK_SRC, HT, GK_DST, 0, nullptr, 1.f, g.num_concat > 0);
        if (c.lean_fold() && g.num_concat > 0) m = res_stack(c, p + ".residual_m", g.num_concat, m2, E, Ed, true, m);
        else {
            m2 = res_stack(c, p + ".residual_m", g.num_concat, m2, E, Ed);
            m = op_axpby(c, m, m2, E, Ed, false, true);
        }
        b->taps["h" + std::to_string(i + 1)] = {h, (int64_t)N * A};
        b->taps["m" + std::to_string(i + 1)] = {m, E * Ed};
        out_block(c, i + 1, m, rbf_out, false);
    }
    b->out_logits = op_dense(c, h, N, A, "fc_atom.weight", 0, ACT_NONE, true, "fc_atom.bias", nullptr, GK_NONE, nullptr, GK_NONE, LOGIT_LD);
    b->taps["Fe"] = {b->Fe, E};
    b->taps["Se"] = {b->Se, E};
    b->taps["out_pos"] = {b->out_pos, (int64_t)N * 3};
    b->taps["V"] = {b->V, E * 3};
    if (!c.dry && CTX_OK(c)) {
        if (E == 0) {
            MI_HIP_VOID(hipMemsetAsync(b->Fe, 0, sizeof(float), c.s));
        }
        hipLaunchKernelGGL(force_fwd_kernel, dim3(nblk(N)), dim3(256), 0, c.s, b->Fe, b->V, b->rowptr, b->node2graph, cell, b->out_pos, N);
        hipLaunchKernelGGL(stress_fwd_kernel, dim3(B), dim3(256), 0, c.s, b->Se, b->V, b->rowptr, b->node_off, b->out_cell);
    }
}

// nosync: no host round trip -- the launches that follow are sized by the capacity E_cap and read the edge count from meta[0]; crystals
// over a capacity keep their flag in b->bad (sticky: the caller clears it at the start of a chain and reads it back at its end)
static int graph_build(mi_gemnet* net, mi_gbatch* b, const float* pos, const float* cell, hipStream_t s, bool nosync = false) {
    const mi_gemnet_config& g = net->cfg;
    MI_HIP(hipMemsetAsync(b->meta, 0, 4 * sizeof(int), s));
    if (!nosync) MI_HIP(hipMemsetAsync(b->bad, 0, (size_t)std::max(b->B, 1) * sizeof(int), s));
    GraphArgs ga{pos, cell, b->node_off, g.cutoff, g.max_neighbors, g.max_images, b->cap, b->ent, b->acnt, b->deg, b->mcount, b->meta, b->bad};
    int nmax_sel = 1;
    for (int v : b->num_atoms_h) nmax_sel = std::max(nmax_sel, v);
    MI_HIP(hipMemsetAsync(b->deg, 0, (size_t)std::max(b->N, 1) * sizeof(int), s));
    hipLaunchKernelGGL(gg_select_kernel, dim3(b->B, (nmax_sel + 3) / 4), dim3(256), 0, s, ga);
    const int dg_cap = std::max(4, std::min(GN_DEG, g_mg_deg_cap));
    hipLaunchKernelGGL(gg_fix_kernel, dim3(b->B), dim3(64), 0, s, b->node_off, b->deg, b->acnt, b->bad, b->meta, dg_cap);
    hipLaunchKernelGGL(gg_scan_kernel, dim3(1), dim3(1024), 0, s, b->deg, b->N, b->rowptr, b->meta);
    EmitArgs ea{b->node_off, b->ent, b->acnt, b->rowptr, b->meta, b->cap, g.max_images, b->E_cap, b->src, b->dst, b->code, b->ekey, b->swap, b->edge_graph};
    int nmax = 1;
    for (int v : b->num_atoms_h) nmax = std::max(nmax, v);
    hipLaunchKernelGGL(gg_emit_kernel, dim3(b->B), dim3(256), (size_t)nmax * b->cap * sizeof(int), s, ea);
    MI_KERNEL_CHECK();
    b->nosync = nosync;
    if (nosync) {
        b->E = b->E_cap;
        b->deg_max = dg_cap;   // (the triplet kernels' LDS image is sized for the capacity: no crystal left in the graph exceeds it)
        return MI_OK;
    }
    int meta[4];
    MI_HIP(hipMemcpyAsync(meta, b->meta, sizeof(meta), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    MI_CHECK(meta[2] == 0, MI_ECAPACITY, "periodic graph: capacity exceeded (flags %d: 1 = more than max_neighbors kept pairs of one atom, 2 = more than %d atoms "
             "inside the cutoff even after shrinking it, 4 = in-degree above %d)", meta[2], GN_CAND, GN_DEG);
    MI_CHECK((int64_t)meta[0] <= b->E_cap, MI_ECAPACITY, "periodic graph: %d edges exceed the capacity %lld", meta[0], (long long)b->E_cap);
    b->E = meta[0];
    b->deg_max = std::max(1, std::min(meta[1], GN_DEG));
    return MI_OK;
}

// nosync (inference only; the sampler's forwards): the graph's edge count stays on the device.  Every edge-level launch is sized by the
// capacity E_cap = 2 max_neighbors N and reads the count from meta[0] (PlanesEpilogue::m_dev, dev_rows); the program's shape is then
// the same for every evaluation, so its dry passes run once per handle, and the arena is compact (see mi_gbatch).  Needs the lean
// plane-mode program (every edge-level layer on the plane-set kernel); otherwise the call falls back to the synchronising form.
static int forward_impl(mi_gemnet* net, mi_gbatch* b, const float* pos, const float* cell, const int* types, const float* t, bool train, hipStream_t s,
                        bool nosync = false) {
    MI_CHECK(net->theta != nullptr, MI_ESTATE, "mi_gemnet_forward before mi_gemnet_set_params");
    b->tape_valid = false;
    nosync = nosync && !train && MI_PLANES_FP16 && g_mg_planes && g_gemm_mode != 0 && (g_mg_lean & 63) == 63 && b->E_cap >= MG_PLANES_MIN_ROWS && !g_optime &&
             net->sf_all_unit;   // (a ScalingFactor other than one is an elementwise pass on fp32 rows: op_scale)
    MI_TRY(graph_build(net, b, pos, cell, s, nosync));
    char* const real_base = b->fwd.base;
    size_t need = 0;
    b->compact = nosync;
    b->collect = false;
    // (the cached dry passes are valid for ONE program shape: the capacity, the network, and every knob that selects a kernel form or a tensor's format)
    uint64_t key = 0;
    if (nosync) {
        const uint64_t parts[] = {(uint64_t)b->E, (uint64_t)(uintptr_t)net, (uint64_t)g_mg_lean, (uint64_t)g_mg_f16, (uint64_t)g_gemm_mode, (uint64_t)g_planes_rt,
                                  (uint64_t)g_planes_rt_min_rows, (uint64_t)g_planes_big, (uint64_t)g_planes_big_min_rows, (uint64_t)g_planes_dma,
                                  (uint64_t)g_planes_lat_max_blocks, (uint64_t)g_planes_variant, (uint64_t)g_mg_planes, net->sf_pattern};
        key = 0xcbf29ce484222325ull;
        for (uint64_t v : parts) key = (key ^ v) * 0x100000001b3ull;   // FNV-1a over the words
        key |= 1;
    }
    if (nosync && b->dry_key == key) {
        need = b->dry_need;   // (same program as the last such forward of this handle: its dry passes are on file)
    } else {
        if (!real_base) b->fwd.base = reinterpret_cast<char*>(uintptr_t(1) << 20);   // (the dry run keys tables by tensor address: never a null one)
        Ctx dry{net, b, s, true, train};
        if (nosync) {   // first dry pass: which plane-only tensors does some consumer materialise as fp32 rows?
            b->collect = true;
            b->mat_ord.clear();
            b->ord_of.clear();
            run_program(dry, pos, cell, types, t);
            b->collect = false;
            b->ord_of.clear();
        }
        run_program(dry, pos, cell, types, t);
        b->fwd.base = real_base;
        need = b->fwd.top;
        MI_TRY(dry.rc);
        if (nosync) {
            b->dry_key = key;
            b->dry_need = need;
        }
    }
    MI_TRY(arena_ensure(b->fwd, need));
    const mi_gemnet_config& g = net->cfg;
    const size_t dz = (size_t)std::max<int64_t>(b->E, b->N) * std::max(std::max(g.emb_edge, g.emb_atom), LOGIT_LD);
    const size_t red = (size_t)1 << 24;
    if (!nosync && b->scratch_floats < dz + red) {   // (the backward's buffers: a forward without a host round trip is inference only)
        if (b->scratch) (void)hipFree(b->scratch);
        b->scratch = nullptr;
        MI_HIP(hipMalloc((void**)&b->scratch, (dz + dz / 8 + red) * sizeof(float)));
        b->scratch_floats = dz + dz / 8 + red;
        b->dz_floats = dz + dz / 8;
    }
    if (train && MI_PLANES_FP16 && g_mg_bwd_planes && g_mg_planes && g_gemm_mode != 0 && b->E >= MG_PLANES_MIN_ROWS) {
        const size_t pe = planes_elems(b->E, std::max(g.emb_edge, g.emb_atom));
        if (b->dzpl_elems < pe) {
            if (b->dzpl) (void)hipFree(b->dzpl);
            b->dzpl = nullptr;
            b->dzpl_elems = 0;
            MI_HIP(hipMalloc((void**)&b->dzpl, (pe + pe / 8) * sizeof(u16)));
            b->dzpl_elems = pe + pe / 8;
        }
        if (!b->bwd_amax) {
            MI_HIP(hipMalloc((void**)&b->bwd_amax, AMAX_W * sizeof(unsigned)));
            MI_HIP(hipMalloc((void**)&b->bwd_dsc, 2 * sizeof(float)));
        }
    }
    if (train) {  // the backward re-reads the inputs the forward saw
        MI_HIP(hipMemcpyAsync(b->types_copy, types, (size_t)b->N * sizeof(int), hipMemcpyDeviceToDevice, s));
        MI_HIP(hipMemcpyAsync(b->cell_copy, cell, (size_t)b->B * 9 * sizeof(float), hipMemcpyDeviceToDevice, s));
        types = b->types_copy;
    }
    Ctx run{net, b, s, false, train};
    b->pl_ranges.clear();
    run_program(run, pos, train ? b->cell_copy : cell, types, t);
    MI_CHECK(b->fwd.top <= need, MI_ESTATE, "activation arena: the program used %zu bytes, its dry run %zu", b->fwd.top, need);
    MI_TRY(run.rc);
    MI_KERNEL_CHECK();
    b->tape_valid = train;
    return MI_OK;
}

static int backward_impl(mi_gemnet* net, mi_gbatch* b, const float* d_pos, const float* d_cell, const float* d_logits, float* grad, hipStream_t s) {
    MI_CHECK(b->tape_valid, MI_ESTATE, "mi_gemnet_backward without a pending training forward on this batch");
    MI_CHECK(net->thetaT != nullptr, MI_ESTATE, "mi_gemnet_set_params must run before backward");
    b->tape_valid = false;
    MI_TRY(arena_ensure(b->grad, b->fwd.top));
    {   // zero the gradient arena -- except the mirror of the forward's plane sets, which no gradient kernel touches (a third of the arena)
        size_t pos = 0;
        for (const auto& r : b->pl_ranges) {
            if (r.first > pos) MI_HIP(hipMemsetAsync(b->grad.base + pos, 0, r.first - pos, s));
            pos = std::max(pos, r.first + r.second);
        }
        if (pos < b->fwd.top) MI_HIP(hipMemsetAsync(b->grad.base + pos, 0, b->fwd.top - pos, s));
    }
    auto G = [&](const float* p) -> float* {
        if (!p) return nullptr;
        const char* q = reinterpret_cast<const char*>(p);
        if (q < b->fwd.base || q >= b->fwd.base + b->fwd.top) return nullptr;
        return reinterpret_cast<float*>(b->grad.base + (q - b->fwd.base));
    };
    const mi_gemnet_config& g = net->cfg;
    const int64_t E = b->E;
    const int N = b->N, B = b->B;
    float* dz = b->scratch;
    float* red = b->scratch + b->dz_floats;
    const size_t red_floats = b->scratch_floats - b->dz_floats;
    // seeds: heads
    if (d_pos && E > 0) hipLaunchKernelGGL(force_bwd_kernel, dim3(nblk(E)), dim3(256), 0, s, d_pos, b->V, b->dst, b->node2graph, b->cell_copy, G(b->Fe), E);
    if (d_cell && E > 0)
        hipLaunchKernelGGL(stress_bwd_kernel, dim3(nblk(E)), dim3(256), 0, s, d_cell, b->V, b->edge_graph, b->rowptr, b->node_off, G(b->Se), E);
    if (d_logits)
        hipLaunchKernelGGL(copy_ld_kernel, dim3(nblk((int64_t)N * MI_MG_CLASSES)), dim3(256), 0, s, d_logits, MI_MG_CLASSES, G(b->out_logits), LOGIT_LD, (int64_t)N,
                           MI_MG_CLASSES);
    MI_KERNEL_CHECK();
    for (int k = (int)b->tape.size() - 1; k >= 0; --k) {
        const GOp& o = b->tape[k];
        float* dY = G(o.Y);
        switch (o.type) {
            case OP_DENSE: {
                if (o.M == 0) break;
                const GParam& w = net->params[o.pidx];
                const float* dZ = dY;
                int ldz = o.ldy;
                float* dX = o.x_grad ? G(o.X) : nullptr;
                const bool elementwise = o.act != ACT_NONE || o.X2 || o.s != 1.f;
                // edge-level data gradient on the plane-set kernel: dZ leaves the activation-gradient pass as a plane set as well
                const bool bwd_pl = MI_PLANES_FP16 && g_mg_bwd_planes && b->planes_mode && b->dzpl && dX && !w.derived && o.M >= MG_PLANES_MIN_ROWS &&
                                    o.N % 32 == 0 && (o.K & 7) == 0 && o.ldy == o.N && planes_elems(o.M, o.N) <= b->dzpl_elems;
                Planes dzp;
                if (bwd_pl) {
                    MI_HIP(hipMemsetAsync(b->bwd_amax, 0, AMAX_W * sizeof(unsigned), s));
                    hipLaunchKernelGGL(absmax_bits_kernel<>, dim3((unsigned)std::min<int64_t>(2048, cdiv(o.M * o.N, 1024))), dim3(256), 0, s, dY, o.M * o.N, b->bwd_amax,
                                       AMAX_W - 1);
                    hipLaunchKernelGGL(mg_scale_kernel, dim3(1), dim3(AMAX_W), 0, s, b->bwd_amax, (const float*)nullptr, (const unsigned*)nullptr, (const int*)nullptr, 1.f,
                                       (const unsigned*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr, o.act != ACT_NONE ? 1.1f * GN_ACT : 1.f, fabsf(o.s),
                                       b->bwd_dsc);
                    dzp = make_planes(b->dzpl, o.N, 1.f, b->bwd_dsc);
                    hipLaunchKernelGGL(act_bwd_pl_kernel, dim3(nblk(o.M * (o.N / 2))), dim3(256), 0, s, dY, o.ldy, o.act != ACT_NONE ? o.Z : (const float*)nullptr, o.s,
                                       G(o.X2), elementwise ? dz : (float*)nullptr, dzp, o.M, o.N);
                    if (elementwise) {
                        dZ = dz;
                        ldz = o.N;
                    }
                } else if (elementwise) {
                    hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(o.M * o.N)), dim3(256), 0, s, dY, o.ldy, o.act != ACT_NONE ? o.Z : (const float*)nullptr, o.s,
                                       G(o.X2), dz, o.M, o.N);
                    dZ = dz;
                    ldz = o.N;
                }
                if (o.bidx >= 0) MI_TRY(colsum_acc(dZ, ldz, grad + net->params[o.bidx].off, (int)o.M, o.N, red, red_floats, s));
                for (int q = 0; q < 2; ++q) {
                    const float* Gq = q == 0 ? o.G1 : o.G2;
                    const int gk = q == 0 ? o.gk1 : o.gk2;
                    if (!Gq) continue;
                    float* dG = G(Gq);
                    MI_CHECK(dG && ldz == o.N, MI_ESTATE, "gathered addend outside the arena");
                    if (gk == GK_DST) hipLaunchKernelGGL(segsum_kernel, dim3(nblk((int64_t)N * o.N)), dim3(256), 0, s, dZ, ldz, b->rowptr, (const int*)nullptr, dG, N, o.N, 1);
                    else if (gk == GK_SRC) hipLaunchKernelGGL(segsum_kernel, dim3(nblk((int64_t)N * o.N)), dim3(256), 0, s, dZ, ldz, b->rowptr, b->swap, dG, N, o.N, 1);
                    else hipLaunchKernelGGL(segsum_kernel, dim3(nblk((int64_t)B * o.N)), dim3(256), 0, s, dZ, ldz, b->node_off, (const int*)nullptr, dG, B, o.N, 1);
                }
                // dW[:, wcol0 : wcol0 + K] += dZ^T X
                // (both operands with rigorous power-of-two scales -- dZ's from this pass, X's from its forward plane set: the weight-gradient
                //  product splits them into two fp16 planes, three terms, instead of three bf16 planes, six terms)
                const float* sx = nullptr;
                if (bwd_pl) {
                    auto xi = b->pl_of.find(o.X);
                    if (xi != b->pl_of.end() && xi->second.dsc) sx = xi->second.dsc;
                }
                MI_TRY(gemm_tn_auto(dZ, ldz, o.X, o.K, grad + w.off + o.wcol0, w.cols, (int)o.M, o.N, o.K, red, red_floats, s, false, sx ? b->bwd_dsc : nullptr, sx));
                if (bwd_pl) {
                    mi_gemnet::WPl wt;
                    MI_TRY(get_wtplanes(net, s, o.pidx, o.wcol0, o.K, &wt));
                    PlanesEpilogue pd;
                    pd.C = dX;
                    pd.ldc = o.K;
                    pd.ep.residual = dX;
                    pd.ep.ld_res = o.K;
                    Planes Wtp = make_planes(wt.pl, o.N, wt.scale);
                    Wtp.frag = wt.frag;
                    MI_TRY(gemm_planes(dzp, Wtp, (int)o.M, o.K, o.N, pd, s));
                } else if (dX) {  // dX += dZ W[:, wcol0 : wcol0 + K]  = dZ (W^T rows wcol0..)^T
                    GemmEpilogue ep;
                    ep.residual = dX;
                    ep.ld_res = o.K;
                    const int kk = (ldz == o.N) ? o.N : ldz;   // logits: the padded columns are zero on both sides
                    MI_TRY(gemm_nt(dZ, ldz, net->thetaT + w.toff + (size_t)o.wcol0 * w.ldt, w.ldt, dX, o.K, (int)o.M, o.K, kk, ep, s));
                }
                break;
            }
            case OP_MUL:
                hipLaunchKernelGGL(mul_bwd_kernel, dim3(nblk(o.M * o.N)), dim3(256), 0, s, o.X, o.X2, dY, G(o.X), G(o.X2), o.M * o.N);
                break;
            case OP_AXPBY:
                hipLaunchKernelGGL(axpby_bwd_kernel, dim3(nblk(o.M * o.N)), dim3(256), 0, s, dY, o.perm ? b->swap : (const int*)nullptr, o.s, G(o.X), G(o.X2), o.M, o.N);
                break;
            case OP_SCALE:
                if (o.M > 0) hipLaunchKernelGGL(scale_bwd_kernel, dim3(nblk(o.M * o.N)), dim3(256), 0, s, dY, net->theta + net->params[o.widx].off, G(o.X), o.M * o.N);
                break;
            case OP_SEGSUM:
                if (o.M > 0) hipLaunchKernelGGL(gather_add_kernel, dim3(nblk(o.M * o.N)), dim3(256), 0, s, dY, b->dst, G(o.X), o.M, o.N);
                break;
            case OP_TRIPLET: {
                if (o.M == 0) break;
                const int DG = std::min(GN_DEG, (b->deg_max + 3) / 4 * 4);
                const size_t sh = triplet_bwd_lds_floats(g.emb_trip, g.emb_cbf, DG) * sizeof(float);
                const size_t sh_max = std::min<size_t>(160 * 1024 - 2048, triplet_bwd_lds_floats(g.emb_trip, g.emb_cbf, GN_DEG) * sizeof(float));
                MI_CHECK(sh <= 160 * 1024 && g.emb_trip <= 64 && g.emb_trip % 4 == 0, MI_EINVAL, "triplet backward: emb_trip / emb_cbf beyond the kernel's LDS budget");
#define TRIP_BWD(SS)                                                                                                                              \
    case SS:                                                                                                                                      \
        raise_lds_ceiling(reinterpret_cast<const void*>(&triplet_bwd_kernel<SS>), (int)sh_max);                                                   \
        hipLaunchKernelGGL((triplet_bwd_kernel<SS>), dim3(N), dim3(512), sh, s, o.X, b->V, o.X2, b->rowptr, dY, G(o.X), G(o.X2), g.emb_trip, g.emb_cbf, DG); \
        break;
                switch (g.num_spherical) {
                    TRIP_BWD(1) TRIP_BWD(2) TRIP_BWD(3) TRIP_BWD(4) TRIP_BWD(5) TRIP_BWD(6) TRIP_BWD(7) TRIP_BWD(8)
                }
#undef TRIP_BWD
                break;
            }
            case OP_ROWDOT: {
                if (o.M == 0) break;
                const GParam& w = net->params[o.widx];
                hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(nblk(o.M * o.K)), dim3(256), 0, s, o.X, o.X2, net->theta + w.off, dY, G(o.X), G(o.X2), o.M, o.K);
                int chunks = (int)std::min<int64_t>(512, (o.M + 255) / 256);
                while ((size_t)chunks * o.K > red_floats) --chunks;
                const int rows_per = (int)((o.M + chunks - 1) / chunks);
                chunks = (int)((o.M + rows_per - 1) / rows_per);
                hipLaunchKernelGGL(rowdot_dw_kernel, dim3(nblk(o.K), chunks), dim3(256), 0, s, o.X, o.X2, dY, red, o.M, o.K, rows_per);
                hipLaunchKernelGGL(part_reduce_kernel<>, dim3(cdiv(o.K, PART_REDUCE_COLS)), dim3(256), 0, s, red, chunks, o.K, grad + w.off, o.K);
                break;
            }
            case OP_EMBED:
                hipLaunchKernelGGL(embed_bwd_kernel, dim3(MI_MG_CLASSES), dim3(256), 0, s, dY, b->types_copy, grad + net->P("atom_emb.weight").off, N, (int)o.N);
                break;
            default: break;
        }
        MI_KERNEL_CHECK();
    }
    return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_gemnet_create(const mi_gemnet_config* cfg, mi_gemnet** out) {
    MI_CHECK(cfg && out, MI_EINVAL, "null argument");
    const mi_gemnet_config& g = *cfg;
    MI_CHECK(g.emb_atom % 4 == 0 && g.emb_edge % 4 == 0 && g.emb_trip % 4 == 0 && g.emb_rbf % 4 == 0 && g.emb_cbf >= 1 && g.emb_bil % 4 == 0 &&
                 g.num_radial % 4 == 0 && g.num_radial >= 4,
             MI_EINVAL, "embedding sizes and num_radial must be multiples of 4");
    MI_CHECK(g.emb_trip <= 64 && g.num_spherical >= 1 && g.num_spherical <= 8 && g.emb_atom % 2 == 0, MI_EINVAL, "emb_trip <= 64, 1 <= num_spherical <= 8");
    MI_CHECK((g.num_spherical * g.emb_cbf) % 4 == 0 && (g.emb_cbf * g.emb_trip) % 4 == 0, MI_EINVAL, "num_spherical * emb_cbf must be a multiple of 4");
    MI_CHECK(g.max_images >= 1 && g.max_images <= 5 && g.max_neighbors >= 1 && g.max_neighbors <= 64 && g.cutoff > 0, MI_EINVAL,
             "1 <= max_images <= 5, 1 <= max_neighbors <= 64, cutoff > 0");
    MI_CHECK(g.emb_atom == g.emb_edge, MI_EINVAL, "emb_atom must equal emb_edge (the atom update maps edge sums to atom features at equal width)");
    mi_gemnet* n = new mi_gemnet();
    n->cfg = g;
    int64_t off = 0, toff = 0;
    auto add = [&](const std::string& name, int rows, int cols) {
        GParam p;
        p.name = name;
        p.off = off;
        p.numel = (int64_t)rows * cols;
        p.rows = rows;
        p.cols = cols;
        p.ldt = (rows + 3) / 4 * 4;
        p.toff = toff;
        n->index[name] = (int)n->params.size();
        n->params.push_back(p);
        off += ((int64_t)rows * cols + 3) / 4 * 4;
        toff += (int64_t)cols * p.ldt;
    };
    const int A = g.emb_atom, Ed = g.emb_edge, Tr = g.emb_trip, Rb = g.emb_rbf, Cb = g.emb_cbf, Bl = g.emb_bil, R = g.num_radial, S = g.num_spherical;
    add("atom_emb.weight", MI_MG_CLASSES, A);
    add("atom_latent_emb.weight", A, 2 * A);
    add("atom_latent_emb.bias", 1, A);
    add("edge_emb.weight", Ed, 2 * A + R);
    add("mlp_rbf3.weight", Rb, R);
    add("mlp_cbf3.weight", S * Cb, R);
    add("mlp_rbf_h.weight", Rb, R);
    add("mlp_rbf_out.weight", Rb, R);
    auto res = [&](const std::string& prefix, int cnt, int width) {
        for (int k = 0; k < cnt; ++k) {
            add(prefix + "." + std::to_string(k) + ".0.weight", width, width);
            add(prefix + "." + std::to_string(k) + ".1.weight", width, width);
        }
    };
    auto outb = [&](int i) {
        // GemNet-T's OutputBlock with direct forces [UPSTREAM-UNVERIFIED] (oracle/mattergen_oracle.py::param_list): the ENERGY path's tensors are part
        // of the flat vector (the upstream denoiser evaluates E_t and never reads it: no output of this library depends on them, their gradient is
        // identically zero and no kernel touches them), the FORCE path is Dense + num_atom residual layers, the lattice head this restatement's own
        const std::string p = "out_blocks." + std::to_string(i);
        add(p + ".dense_rbf.weight", Ed, Rb);
        add(p + ".scale_sum.scale_factor", 1, 1);
        add(p + ".seq_energy.dense.weight", A, Ed);
        res(p + ".seq_energy.res", g.num_atom, A);
        add(p + ".out_energy.weight", 1, A);
        add(p + ".dense_F.weight", Ed, Ed);
        res(p + ".res_F", g.num_atom, Ed);
        add(p + ".rbf_F.weight", Ed, Rb);
        add(p + ".scale_rbf_F.scale_factor", 1, 1);
        add(p + ".out_F.weight", 1, Ed);
        add(p + ".dense_S.weight", Ed, Ed);
        add(p + ".rbf_S.weight", Ed, Rb);
        add(p + ".out_S.weight", 1, Ed);
    };
    outb(0);
    for (int i = 0; i < g.num_blocks; ++i) {
        const std::string p = "int_blocks." + std::to_string(i);
        add(p + ".dense_ca.weight", Ed, Ed);
        add(p + ".dense_ba.weight", Ed, Ed);
        add(p + ".mlp_rbf.weight", Ed, Rb);
        add(p + ".scale_rbf.scale_factor", 1, 1);
        add(p + ".down_projection.weight", Tr, Ed);
        add(p + ".bilinear.weight", Bl, Cb * Tr);
        add(p + ".scale_cbf_sum.scale_factor", 1, 1);
        add(p + ".up_projection_ca.weight", Ed, Bl);
        add(p + ".up_projection_ac.weight", Ed, Bl);
        res(p + ".before_skip", g.num_before_skip, Ed);
        res(p + ".after_skip", g.num_after_skip, Ed);
        add(p + ".atom_update.rbf.weight", Ed, Rb);
        add(p + ".atom_update.scale_sum.scale_factor", 1, 1);
        add(p + ".atom_update.dense.weight", A, Ed);
        res(p + ".atom_update.res", g.num_atom, A);
        add(p + ".concat.weight", Ed, 2 * A + Ed);
        res(p + ".residual_m", g.num_concat, Ed);
        outb(i + 1);
    }
    add("fc_atom.weight", MI_MG_CLASSES, A);
    add("fc_atom.bias", 1, MI_MG_CLASSES);
    n->nparams = off;
    n->ntrans = toff;
    n->n_real = (int)n->params.size();
    {
        int64_t doff = 0;
        for (int i = 0; i <= g.num_blocks; ++i)
            for (const char* h : {"F", "S"}) {
                GParam p;
                p.name = "out_blocks." + std::to_string(i) + ".q_" + h;
                p.off = doff;
                p.numel = (int64_t)Rb * Ed;
                p.rows = Rb;
                p.cols = Ed;
                p.ldt = 0;
                p.toff = 0;
                p.derived = true;
                n->index[p.name] = (int)n->params.size();
                n->params.push_back(p);
                doff += (p.numel + 3) / 4 * 4;
            }
        n->ndtheta = doff;
    }
    *out = n;
    return MI_OK;
}

void mi_gemnet_destroy(mi_gemnet* net) {
    if (!net) return;
    if (net->thetaT) (void)hipFree(net->thetaT);
    if (net->wamax) (void)hipFree(net->wamax);
    if (net->dtheta) (void)hipFree(net->dtheta);
    if (net->warena) (void)hipFree(net->warena);
    if (net->wrowsum) (void)hipFree(net->wrowsum);
    for (hipEvent_t e : net->wevents) (void)hipEventDestroy(e);
    delete net;
}
int64_t mi_gemnet_num_params(const mi_gemnet* net) { return net ? net->nparams : 0; }
int mi_gemnet_num_tensors(const mi_gemnet* net) { return net ? net->n_real : 0; }
int mi_gemnet_param_info(const mi_gemnet* net, int index, const char** name, int64_t* offset, int64_t* numel, int* rows, int* cols) {
    MI_CHECK(net && index >= 0 && index < net->n_real, MI_EINVAL, "parameter index out of range");
    const GParam& p = net->params[index];
    if (name) *name = p.name.c_str();
    if (offset) *offset = p.off;
    if (numel) *numel = p.numel;
    if (rows) *rows = p.rows;
    if (cols) *cols = p.cols;
    return MI_OK;
}
int mi_gemnet_set_params(mi_gemnet* net, const float* theta, void* stream) {
    MI_CHECK(net && theta, MI_EINVAL, "null argument");
    MI_CHECK((((uintptr_t)theta) & 15) == 0, MI_EINVAL, "theta must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    net->theta = theta;
    if (!net->thetaT) {
        MI_HIP(hipMalloc((void**)&net->thetaT, (size_t)net->ntrans * sizeof(float)));
        MI_HIP(hipMemsetAsync(net->thetaT, 0, (size_t)net->ntrans * sizeof(float), s));
    }
    if (!net->wamax) MI_HIP(hipMalloc((void**)&net->wamax, net->params.size() * sizeof(unsigned)));
    if (!net->dtheta && net->ndtheta > 0) MI_HIP(hipMalloc((void**)&net->dtheta, (size_t)net->ndtheta * sizeof(float)));
    for (int i = 0; i <= net->cfg.num_blocks; ++i)
        for (const char* h : {"F", "S"}) {
            const std::string pre = "out_blocks." + std::to_string(i);
            const GParam &q = net->P(pre + ".q_" + h), &wr = net->P(pre + ".rbf_" + h + ".weight"), &wo = net->P(pre + ".out_" + h + ".weight");
            hipLaunchKernelGGL(derive_q_kernel, dim3(nblk(q.numel)), dim3(256), 0, s, theta + wo.off, theta + wr.off, net->dtheta + q.off, q.cols, q.rows);
        }
    MI_HIP(hipMemsetAsync(net->wamax, 0, net->params.size() * sizeof(unsigned), s));
    for (size_t i = 0; i < net->params.size(); ++i) {
        const GParam& p = net->params[i];
        hipLaunchKernelGGL(absmax_bits_kernel<>, dim3((unsigned)std::min<int64_t>(256, cdiv(p.numel, 1024))), dim3(256), 0, s, net->wptr(p), p.numel, net->wamax + i);
    }
    {   // per-tensor plane scales on the host (2^floor(log2(16384 / absmax))); the lazily built weight plane sets are stale now
        std::vector<unsigned> bits(net->params.size());
        MI_HIP(hipMemcpyAsync(bits.data(), net->wamax, bits.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        std::vector<float> sf(net->params.size(), 1.f);
        for (size_t i = 0; i < (size_t)net->n_real; ++i) {
            const std::string& nm = net->params[i].name;
            if (nm.size() > 13 && nm.compare(nm.size() - 13, 13, ".scale_factor") == 0)
                MI_HIP(hipMemcpyAsync(&sf[i], theta + net->params[i].off, sizeof(float), hipMemcpyDeviceToHost, s));
        }
        MI_HIP(hipStreamSynchronize(s));
        {
            uint64_t pat = 0xcbf29ce484222325ull;
            bool unit = true;
            for (size_t i = 0; i < sf.size(); ++i)
                if (sf[i] != 1.f) {
                    unit = false;
                    pat = (pat ^ (uint64_t)(i + 1)) * 0x100000001b3ull;
                }
            net->sf_h = sf;
            net->sf_all_unit = unit;
            net->sf_pattern = unit ? 0 : pat;
        }
        net->wscale_h.assign(bits.size(), 1.f);
        for (size_t i = 0; i < bits.size(); ++i) {
            float m;
            memcpy(&m, &bits[i], 4);
            if (m > 0.f && m < 3e38f) net->wscale_h[i] = exp2f((float)std::min(30, 14 - (int)ceilf(log2f(m))));
        }
        std::lock_guard<std::mutex> guard(net->wmu);
        net->wplanes.clear();
        for (hipEvent_t e : net->wevents) (void)hipEventDestroy(e);
        net->wevents.clear();
        net->warena_top = 0;
        net->wrowsum_used = 0;
    }
    for (const GParam& p : net->params) {
        if (p.rows == 1 || p.derived) continue;  // biases and the row-dot weights are never a data-gradient operand; derived tensors are inference-only
        hipLaunchKernelGGL(gn_transpose_kernel, dim3(nblk(p.numel)), dim3(256), 0, s, theta + p.off, net->thetaT + p.toff, p.rows, p.cols, p.ldt);
    }
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_gbatch_create(const mi_gemnet* net, const int* num_atoms_host, int B, int64_t node_offset, int64_t graph_offset, mi_gbatch** out) {
    MI_CHECK(net && num_atoms_host && out && B >= 0, MI_EINVAL, "bad argument");
    mi_gbatch* b = new mi_gbatch();
    b->B = B;
    b->node_offset = node_offset;
    b->graph_offset = graph_offset;
    b->num_atoms_h.assign(num_atoms_host, num_atoms_host + B);
    b->node_off_h.assign(B + 1, 0);
    for (int i = 0; i < B; ++i) {
        if (num_atoms_host[i] < 0 || num_atoms_host[i] > GN_NMAX) {
            set_error("mi_gbatch_create: %d atoms in crystal %d (supported: 0..%d)", num_atoms_host[i], i, GN_NMAX);
            delete b;
            return MI_EINVAL;
        }
        b->node_off_h[i + 1] = b->node_off_h[i] + num_atoms_host[i];
    }
    const int N = b->N = b->node_off_h[B];
    b->cap = net->cfg.max_neighbors;
    b->R_img = net->cfg.max_images;
    b->E_cap = (int64_t)N * std::min(2 * net->cfg.max_neighbors, GN_DEG);
    std::vector<int> n2g(N);
    for (int i = 0; i < B; ++i)
        for (int k = b->node_off_h[i]; k < b->node_off_h[i + 1]; ++k) n2g[k] = i;
    int rc = MI_OK;
#define GA(ptr, n)                                  \
    if (rc == MI_OK) rc = galloc(b, &b->ptr, (size_t)(n))
    GA(num_atoms, B);
    GA(node_off, B + 1);
    GA(node2graph, N);
    GA(ent, (size_t)N * b->cap);
    GA(acnt, N);
    GA(deg, N);
    GA(mcount, B);
    GA(meta, 4);
    GA(bad, B);
    GA(rowptr, N + 1);
    GA(src, b->E_cap);
    GA(dst, b->E_cap);
    GA(code, b->E_cap);
    GA(ekey, b->E_cap);
    GA(swap, b->E_cap);
    GA(edge_graph, b->E_cap);
    GA(D, b->E_cap);
    GA(V, b->E_cap * 3);
    GA(types_copy, N);
    GA(cell_copy, (size_t)B * 9);
    GA(t_buf, B);
    GA(sp_pos, (size_t)N * 3);
    GA(sp_cell, (size_t)B * 9);
    GA(sp_logits, (size_t)N * MI_MG_CLASSES);
    GA(amax_pool, (size_t)AMAX_SLOTS * AMAX_W);
    GA(dsc_pool, 2 * AMAX_SLOTS);
#undef GA
    if (rc != MI_OK) {
        mi_gbatch_destroy(b);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (B > 0) e = hipMemcpy(b->num_atoms, num_atoms_host, B * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->node_off, b->node_off_h.data(), (B + 1) * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess && N > 0) e = hipMemcpy(b->node2graph, n2g.data(), N * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(b->rowptr, 0, (N + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMemset(b->bad, 0, (size_t)std::max(B, 1) * sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->status_h, (size_t)(B + 4) * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // (the handle will be used on non-blocking side streams: nothing of its set-up may still be in flight on the null stream)
    if (e != hipSuccess) {
        set_error("mi_gbatch_create: %s", hipGetErrorString(e));
        mi_gbatch_destroy(b);
        return MI_EHIP;
    }
    *out = b;
    return MI_OK;
}

int mi_gbatch_set_offsets(mi_gbatch* b, int64_t node_offset, int64_t graph_offset) {
    MI_CHECK(b && node_offset >= 0 && graph_offset >= 0, MI_EINVAL, "bad argument");
    b->node_offset = node_offset;
    b->graph_offset = graph_offset;
    return MI_OK;
}

void mi_gbatch_destroy(mi_gbatch* b) {
    if (!b) return;
    for (void* p : b->allocs) (void)hipFree(p);
    if (b->fwd.base) (void)hipFree(b->fwd.base);
    if (b->grad.base) (void)hipFree(b->grad.base);
    if (b->scratch) (void)hipFree(b->scratch);
    if (b->dzpl) (void)hipFree(b->dzpl);
    if (b->bwd_amax) (void)hipFree(b->bwd_amax);
    if (b->bwd_dsc) (void)hipFree(b->bwd_dsc);
    if (b->ts_dev) (void)hipFree(b->ts_dev);
    if (b->status_h) (void)hipHostFree(b->status_h);
    delete b;
}

int mi_gemnet_graph(mi_gemnet* net, mi_gbatch* b, const float* pos, const float* cell, void* stream, int64_t* num_edges) {
    MI_CHECK(net && b && pos && cell, MI_EINVAL, "null argument");
    if (b->N == 0) {
        b->E = 0;
        if (num_edges) *num_edges = 0;
        return MI_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    MI_TRY(graph_build(net, b, pos, cell, s));
    if (b->E > 0) {  // D / V without the basis: reuse the geometry kernel with a one-column basis into scratch
        float* tmp = nullptr;
        MI_HIP(hipMalloc((void**)&tmp, (size_t)b->E * 4 * sizeof(float)));
        hipLaunchKernelGGL(edge_geom_rbf_kernel, dim3(nblk(b->E * 4)), dim3(256), 0, s, pos, cell, b->src, b->dst, b->code, b->edge_graph, net->cfg.max_images,
                           net->cfg.cutoff, 4, b->E, b->D, b->V, tmp);
        MI_HIP(hipStreamSynchronize(s));
        (void)hipFree(tmp);
    }
    if (num_edges) *num_edges = b->E;
    return MI_OK;
}

__global__ void code_to_img_kernel(const int* __restrict__ code, int* __restrict__ img, int64_t E, int R) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int W = 2 * R + 1, c = code[e];
    img[e * 3] = c / (W * W) - R;
    img[e * 3 + 1] = (c / W) % W - R;
    img[e * 3 + 2] = c % W - R;
}

int mi_gemnet_graph_read(const mi_gbatch* b, int* src, int* dst, int* img, int* swap, int* rowptr, float* D, float* V, void* stream) {
    MI_CHECK(b, MI_EINVAL, "null handle");
    hipStream_t s = (hipStream_t)stream;
    const size_t E = (size_t)b->E;
    if (src) MI_HIP(hipMemcpyAsync(src, b->src, E * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (dst) MI_HIP(hipMemcpyAsync(dst, b->dst, E * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (swap) MI_HIP(hipMemcpyAsync(swap, b->swap, E * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (rowptr) MI_HIP(hipMemcpyAsync(rowptr, b->rowptr, (size_t)(b->N + 1) * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (D) MI_HIP(hipMemcpyAsync(D, b->D, E * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (V) MI_HIP(hipMemcpyAsync(V, b->V, E * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (img && E > 0) {
        hipLaunchKernelGGL(code_to_img_kernel, dim3(nblk((int64_t)E)), dim3(256), 0, s, b->code, img, (int64_t)E, b->R_img);
    }
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_gemnet_forward(mi_gemnet* net, mi_gbatch* b, const float* pos, const float* cell, const int* atomic_numbers, const float* t, float* out_pos,
                      float* out_cell, float* out_logits, int train, void* stream) {
    MI_CHECK(net && b && pos && cell && atomic_numbers && t, MI_EINVAL, "null argument");
    if (b->N == 0 || b->B == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    MI_TRY(forward_impl(net, b, pos, cell, atomic_numbers, t, (train & 1) != 0, s, (train & 2) != 0));
    // (copies by kernel, not hipMemcpyAsync: concurrent sampler chains on four non-blocking streams showed the position head -- the first
    //  copy, issued right behind the kernel that writes its source -- intermittently returning the PREVIOUS evaluation's values)
    if (out_pos) hipLaunchKernelGGL(copy_ld_kernel, dim3(nblk((int64_t)b->N * 3)), dim3(256), 0, s, b->out_pos, 3, out_pos, 3, (int64_t)b->N, 3);
    if (out_cell) hipLaunchKernelGGL(copy_ld_kernel, dim3(nblk((int64_t)b->B * 9)), dim3(256), 0, s, b->out_cell, 9, out_cell, 9, (int64_t)b->B, 9);
    if (out_logits)
        hipLaunchKernelGGL(copy_ld_kernel, dim3(nblk((int64_t)b->N * MI_MG_CLASSES)), dim3(256), 0, s, b->out_logits, LOGIT_LD, out_logits, MI_MG_CLASSES,
                           (int64_t)b->N, MI_MG_CLASSES);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_gemnet_backward(mi_gemnet* net, mi_gbatch* b, const float* d_pos, const float* d_cell, const float* d_logits, float* grad_theta, void* stream) {
    MI_CHECK(net && b && grad_theta, MI_EINVAL, "null argument");
    if (b->N == 0 || b->B == 0) return MI_OK;
    return backward_impl(net, b, d_pos, d_cell, d_logits, grad_theta, (hipStream_t)stream);
}

int mi_debug_set_mg_planes(int on) {
    mi::g_mg_planes = (on & 1) != 0;
    mi::g_mg_bwd_planes = on != 0 && (on & 2) == 0;   // +2: forward on the plane-set kernel, the backward's data gradients on the fp32-operand kernel
    return MI_OK;
}

int mi_debug_set_mg_lean(int on) {
    mi::g_mg_lean = on == 1 ? 63 : on < 0 ? -on : on;   // (0 = off, 1 = everything; other values: the bit mask of g_mg_lean, for ablations)
    return MI_OK;
}

int mi_debug_set_mg_deg_cap(int cap) {
    const int was = mi::g_mg_deg_cap;
    mi::g_mg_deg_cap = cap <= 0 ? mi::GN_DEG : std::min(cap, mi::GN_DEG);
    return was;
}

int mi_debug_set_mg_nosync(int on) {
    const int was = mi::g_mg_nosync;
    mi::g_mg_nosync = on != 0;
    return was;
}

// Graph-capacity flags of the crystals of `b` after mi_mg_sampler_run (or any forward): bad_host[i] != 0 -- crystal i exceeded a
// capacity of the periodic graph at some evaluation (1: more than max_neighbors kept pairs of one atom, 2: more than GN_CAND atoms
// inside the cutoff even after shrinking it, 4: an in-degree above GN_DEG) and ran WITHOUT edges from then on: its sample is invalid
// and must be dropped, the other crystals are unaffected (the reference's invalid_filter, pipeline/filters/opt_filter.py:49-61, drops
// collapsed crystals one by one too).  Waits for the work queued on `stream`.  n_bad: number of flagged crystals.
int mi_gbatch_graph_status(mi_gbatch* b, int* bad_host, int* n_bad, void* stream) {
    MI_CHECK(b, MI_EINVAL, "null handle");
    hipStream_t s = (hipStream_t)stream;
    int nb = 0;
    if (b->B > 0) {
        MI_HIP(hipMemcpyAsync(b->status_h, b->bad, (size_t)b->B * sizeof(int), hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < b->B; ++i) {
            if (bad_host) bad_host[i] = b->status_h[i];
            nb += b->status_h[i] != 0;
        }
    }
    if (n_bad) *n_bad = nb;
    return MI_OK;
}

int mi_debug_set_mg_f16(int on) {
    mi::g_mg_f16 = on != 0 && MI_PLANES_FP16;
    return MI_OK;
}

int mi_gemnet_tap(mi_gbatch* b, const char* name, float* out, int64_t capacity, int64_t* numel, void* stream) {
    MI_CHECK(b && name, MI_EINVAL, "null argument");
    auto it = b->taps.find(name);
    MI_CHECK(it != b->taps.end(), MI_EINVAL, "unknown tap %s", name);
    MI_CHECK(!b->nosync, MI_ESTATE, "taps exist after a synchronising forward only (the sampler's forwards keep the edge count on the device)");
    if (numel) *numel = it->second.second;
    if (out) {
        MI_CHECK(capacity >= it->second.second, MI_EINVAL, "tap %s needs %lld floats", name, (long long)it->second.second);
        MI_TRY(mi::materialize_f32(b, it->second.first, (hipStream_t)stream));   // (a lean inference forward kept this tensor as a plane set only)
        MI_HIP(hipMemcpyAsync(out, it->second.first, (size_t)it->second.second * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return MI_OK;
}

}  // extern "C"

// ================================================================================================================================
// corruptions and the predictor-corrector sampler (oracle/mattergen_oracle.py: sample_marginal, pc_sample)
// ================================================================================================================================
namespace mi {

__device__ __forceinline__ float mg_pos_std(const mi_mg_corruption& c, float t, int n) {
    return powf(c.sigma_min, 1.0f - t) * powf(c.sigma_max, t) * powf((float)n, -1.0f / 3.0f);
}
__device__ __forceinline__ float mg_alpha(const mi_mg_corruption& c, float t) {
    return expf(-0.5f * (t * c.beta_min + 0.5f * t * t * (c.beta_max - c.beta_min)));
}
__device__ __forceinline__ float mg_tau(const mi_mg_corruption& c, float t) {
    return fminf(fmaxf(ceilf(t * (float)c.d3pm_steps - 1e-6f), 1.0f), (float)c.d3pm_steps);
}
// entry q = 3 i + j of the symmetric noise made of the 9 normals G: diagonal kept, off-diagonal (G_ij + G_ji) / sqrt(2)
__device__ __forceinline__ float sym_entry(const float* G, int q) {
    const int i = q / 3, j = q % 3;
    return i == j ? G[q] : (G[i * 3 + j] + G[j * 3 + i]) * GN_ISQ2;
}

struct MargArgs {
    const float *pos0, *cell0;
    const int* types0;
    const float* t;
    const int* node_off;
    mi_mg_corruption c;
    uint64_t seed;
    uint32_t step;
    int64_t node_offset, graph_offset;
    const float *npos, *ncell, *ntypes;
    float *pos, *cell;
    int* types;
    float *delta, *eps;
    int* masked;
};
__global__ __launch_bounds__(64) void mg_marginal_kernel(MargArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x, n0 = a.node_off[b], n1 = a.node_off[b + 1], n = n1 - n0;
    const float t = a.t[b];
    const float std = mg_pos_std(a.c, t, n > 0 ? n : 1);
    for (int idx = n0 * 3 + lane; idx < n1 * 3; idx += 64) {
        const float z = a.npos ? a.npos[idx] : philox_normal1(a.seed, a.step, DRAW_MG_POS, (uint64_t)a.node_offset * 3 + idx);
        const float d = std * z;
        a.delta[idx] = d;
        a.pos[idx] = pymod1(a.pos0[idx] + d);
    }
    __shared__ float G[9];
    if (lane < 9) G[lane] = a.ncell ? a.ncell[b * 9 + lane] : philox_normal1(a.seed, a.step, DRAW_MG_CELL, (uint64_t)a.graph_offset * 9 + b * 9 + lane);
    __syncthreads();
    if (lane < 9) {
        const float alpha = mg_alpha(a.c, t), nf = (float)(n > 0 ? n : 1);
        const float mu = powf(nf / a.c.limit_density, 1.0f / 3.0f), kap = sqrtf(a.c.limit_var_scale) * powf(nf, 1.0f / 3.0f);
        const float e = sym_entry(G, lane);
        a.eps[b * 9 + lane] = e;
        a.cell[b * 9 + lane] = alpha * a.cell0[b * 9 + lane] + (1.0f - alpha) * mu * (lane % 4 == 0 ? 1.f : 0.f) + sqrtf(1.0f - alpha * alpha) * kap * e;
    }
    const float frac = mg_tau(a.c, t) / (float)a.c.d3pm_steps;
    for (int i = n0 + lane; i < n1; i += 64) {
        const float u = a.ntypes ? a.ntypes[i] : philox_uniform1(a.seed, a.step, DRAW_MG_TYPES, (uint64_t)a.node_offset + i);
        const int mk = u < frac;
        a.masked[i] = mk;
        a.types[i] = mk ? MI_MG_MASK : a.types0[i];
    }
}

struct InitArgs {
    const int* node_off;
    mi_mg_corruption c;
    uint64_t seed;
    int64_t node_offset, graph_offset;
    const float *ipos, *icell;
    float *pos, *cell;
    int* types;
};
__global__ __launch_bounds__(64) void mg_init_kernel(InitArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x, n0 = a.node_off[b], n1 = a.node_off[b + 1], n = n1 - n0;
    for (int idx = n0 * 3 + lane; idx < n1 * 3; idx += 64)
        a.pos[idx] = pymod1(a.ipos ? a.ipos[idx] : philox_uniform1(a.seed, 0u, DRAW_MG_INIT_POS, (uint64_t)a.node_offset * 3 + idx));
    __shared__ float G[9];
    if (lane < 9) G[lane] = a.icell ? a.icell[b * 9 + lane] : philox_normal1(a.seed, 0u, DRAW_MG_INIT_CELL, (uint64_t)a.graph_offset * 9 + b * 9 + lane);
    __syncthreads();
    if (lane < 9) {
        const float nf = (float)(n > 0 ? n : 1);
        const float mu = powf(nf / a.c.limit_density, 1.0f / 3.0f), kap = sqrtf(a.c.limit_var_scale) * powf(nf, 1.0f / 3.0f);
        a.cell[b * 9 + lane] = mu * (lane % 4 == 0 ? 1.f : 0.f) + kap * sym_entry(G, lane);
    }
    for (int i = n0 + lane; i < n1; i += 64) a.types[i] = MI_MG_MASK;
}

struct StepArgs {
    const int* node_off;
    mi_mg_corruption c;
    uint64_t seed;
    uint32_t step;         // Philox counter step = i + 1
    int64_t node_offset, graph_offset;
    float t, dt;
    int last;              // 1 on the final grid point: the next std is zero
    const float *out_pos, *out_cell, *out_logits;   // denoiser outputs (logits row stride MI_MG_CLASSES)
    const float *n_pos, *n_cell, *n_u1, *n_u2;      // injected noise slices or NULL
    float *pos, *cell;
    int* types;
    float *mean_pos, *mean_cell;
};
// Langevin corrector with a per-crystal signal-to-noise step size: step = 2 (snr |z| / |score|)^2, x += step score + sqrt(2 step) z
__global__ __launch_bounds__(64) void mg_corrector_kernel(StepArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x, n0 = a.node_off[b], n1 = a.node_off[b + 1], n = n1 - n0;
    const float stdp = mg_pos_std(a.c, a.t, n > 0 ? n : 1);
    float zz = 0.f, ss = 0.f, zv[3], sv[3];
    int cnt = 0;
    for (int idx = n0 * 3 + lane; idx < n1 * 3; idx += 64, ++cnt) {
        const float z = a.n_pos ? a.n_pos[idx] : philox_normal1(a.seed, a.step, DRAW_MG_CORR_POS, (uint64_t)a.node_offset * 3 + idx);
        const float sc = a.out_pos[idx] / stdp;
        zv[cnt] = z;
        sv[cnt] = sc;
        zz += z * z;
        ss += sc * sc;
    }
    zz = wave_sum(zz);
    ss = wave_sum(ss);
    float r = 0.4f * sqrtf(zz) / fmaxf(sqrtf(ss), 1e-12f);
    float step = fminf(2.0f * r * r, 1e6f);
    cnt = 0;
    for (int idx = n0 * 3 + lane; idx < n1 * 3; idx += 64, ++cnt) a.pos[idx] = pymod1(a.pos[idx] + step * sv[cnt] + sqrtf(2.0f * step) * zv[cnt]);
    __shared__ float G[9];
    if (lane < 9) G[lane] = a.n_cell ? a.n_cell[b * 9 + lane] : philox_normal1(a.seed, a.step, DRAW_MG_CORR_CELL, (uint64_t)a.graph_offset * 9 + b * 9 + lane);
    __syncthreads();
    const float alpha = mg_alpha(a.c, a.t), nf = (float)(n > 0 ? n : 1);
    const float kap = sqrtf(a.c.limit_var_scale) * powf(nf, 1.0f / 3.0f), stdc = sqrtf(1.0f - alpha * alpha) * kap;
    float zc = 0.f, sc = 0.f;
    if (lane < 9) {
        zc = sym_entry(G, lane);
        sc = a.out_cell[b * 9 + lane] / stdc;
    }
    const float zn = wave_sum(zc * zc), sn = wave_sum(sc * sc);
    r = 0.2f * sqrtf(zn) / fmaxf(sqrtf(sn), 1e-12f);
    step = fminf(2.0f * r * r, 1e6f);
    if (lane < 9) a.cell[b * 9 + lane] = a.cell[b * 9 + lane] + step * sc + sqrtf(2.0f * step) * zc;
}
// ancestral predictor: positions (wrapped VE), cell (VP towards the limit mean), types (D3PM absorbing: a masked atom is revealed
// with probability 1 / tau, drawing its element from the softmax over the 100 element logits)
__global__ __launch_bounds__(64) void mg_predictor_kernel(StepArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x, n0 = a.node_off[b], n1 = a.node_off[b + 1], n = n1 - n0;
    const int nn = n > 0 ? n : 1;
    const float stdp = mg_pos_std(a.c, a.t, nn), tn = fmaxf(a.t - a.dt, 0.f), stdn = a.last ? 0.f : mg_pos_std(a.c, tn, nn);
    const float var_d = stdp * stdp - stdn * stdn, nstd = sqrtf(stdn * stdn * var_d / (stdp * stdp));
    for (int idx = n0 * 3 + lane; idx < n1 * 3; idx += 64) {
        const float z = a.n_pos ? a.n_pos[idx] : philox_normal1(a.seed, a.step, DRAW_MG_PRED_POS, (uint64_t)a.node_offset * 3 + idx);
        const float mean = a.pos[idx] + var_d * (a.out_pos[idx] / stdp);
        a.mean_pos[idx] = pymod1(mean);
        a.pos[idx] = pymod1(mean + nstd * z);
    }
    __shared__ float G[9];
    if (lane < 9) G[lane] = a.n_cell ? a.n_cell[b * 9 + lane] : philox_normal1(a.seed, a.step, DRAW_MG_PRED_CELL, (uint64_t)a.graph_offset * 9 + b * 9 + lane);
    __syncthreads();
    if (lane < 9) {
        const float alpha = mg_alpha(a.c, a.t), nf = (float)nn;
        const float mu = powf(nf / a.c.limit_density, 1.0f / 3.0f), kap = sqrtf(a.c.limit_var_scale) * powf(nf, 1.0f / 3.0f);
        const float stdc = sqrtf(1.0f - alpha * alpha) * kap;
        const float bd = (a.c.beta_min + a.t * (a.c.beta_max - a.c.beta_min)) * a.dt;
        const float L = a.cell[b * 9 + lane], sc = a.out_cell[b * 9 + lane] / stdc;
        const float mean = L + 0.5f * bd * (L - mu * (lane % 4 == 0 ? 1.f : 0.f)) + bd * (kap * kap) * sc;
        a.mean_cell[b * 9 + lane] = mean;
        a.cell[b * 9 + lane] = mean + sqrtf(bd) * kap * sym_entry(G, lane);
    }
    const float inv_tau = 1.0f / mg_tau(a.c, a.t);
    for (int i = n0 + lane; i < n1; i += 64) {
        if (a.types[i] != MI_MG_MASK) continue;
        const float u1 = a.n_u1 ? a.n_u1[i] : philox_uniform1(a.seed, a.step, DRAW_MG_PRED_U1, (uint64_t)a.node_offset + i);
        if (!(u1 < inv_tau)) continue;
        const float u2 = a.n_u2 ? a.n_u2[i] : philox_uniform1(a.seed, a.step, DRAW_MG_PRED_U2, (uint64_t)a.node_offset + i);
        const float* lg = a.out_logits + (size_t)i * MI_MG_CLASSES;
        float mx = lg[0];
        for (int k = 1; k < MI_MG_CLASSES - 1; ++k) mx = fmaxf(mx, lg[k]);
        float den = 0.f;
        for (int k = 0; k < MI_MG_CLASSES - 1; ++k) den += expf(lg[k] - mx);
        float cdf = 0.f;
        int draw = 0;
        for (int k = 0; k < MI_MG_CLASSES - 1; ++k) {
            cdf += expf(lg[k] - mx) / den;
            draw += u2 >= cdf;
        }
        a.types[i] = (draw > MI_MG_CLASSES - 2 ? MI_MG_CLASSES - 2 : draw) + 1;
    }
}

}  // namespace mi

extern "C" {

int mi_mg_sample_marginal(mi_gbatch* b, const mi_mg_corruption* c, const float* pos0, const float* cell0, const int* types0, const float* t, uint64_t seed,
                          uint32_t step, const float* noise_pos, const float* noise_cell, const float* noise_types, float* pos, float* cell, int* types,
                          float* delta, float* eps, int* masked, void* stream) {
    MI_CHECK(b && c && pos0 && cell0 && types0 && t && pos && cell && types && delta && eps && masked, MI_EINVAL, "null argument");
    if (b->B == 0) return MI_OK;
    MargArgs a{pos0, cell0, types0, t, b->node_off, *c, seed, step, b->node_offset, b->graph_offset, noise_pos, noise_cell, noise_types, pos, cell, types,
               delta, eps, masked};
    hipLaunchKernelGGL(mg_marginal_kernel, dim3(b->B), dim3(64), 0, (hipStream_t)stream, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_mg_sampler_init(mi_gbatch* b, const mi_mg_corruption* c, uint64_t seed, const float* init_pos, const float* init_cell, float* pos, float* cell,
                       int* types, void* stream) {
    MI_CHECK(b && c && pos && cell && types, MI_EINVAL, "null argument");
    if (b->B == 0) return MI_OK;
    InitArgs a{b->node_off, *c, seed, b->node_offset, b->graph_offset, init_pos, init_cell, pos, cell, types};
    hipLaunchKernelGGL(mg_init_kernel, dim3(b->B), dim3(64), 0, (hipStream_t)stream, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_mg_sampler_run(mi_gemnet* net, mi_gbatch* b, const mi_mg_corruption* c, int n_steps, int i_start, int i_stop, const float* ts_host, uint64_t seed,
                      const mi_mg_sampler_noise* noise, float* pos, float* cell, int* types, float* mean_pos, float* mean_cell, void* stream) {
    MI_CHECK(net && b && c && ts_host && pos && cell && types && mean_pos && mean_cell, MI_EINVAL, "null argument");
    MI_CHECK(n_steps >= 1 && i_start >= 0 && i_stop <= n_steps && i_start <= i_stop, MI_EINVAL, "bad step range");
    if (b->B == 0 || b->N == 0) return MI_OK;
    hipStream_t s = (hipStream_t)stream;
    const int N = b->N, B = b->B;
    const float dt = n_steps > 1 ? ts_host[0] - ts_host[1] : ts_host[0];
    // The chain's forwards keep the graph's edge count on the device (no host round trip per evaluation); a crystal over a graph
    // capacity runs without edges from that evaluation on and keeps its flag: mi_gbatch_graph_status after the chain.
    const int fwd_flags = g_mg_nosync ? 2 : 0;
    MI_HIP(hipMemsetAsync(b->bad, 0, (size_t)B * sizeof(int), s));
    hipLaunchKernelGGL(wrap_pos_kernel, dim3(nblk((int64_t)N * 3)), dim3(256), 0, s, pos, (int64_t)N * 3);   // (a resumed state may be unwrapped; in-library, not a torch op on a chain's stream)
    for (int i = i_start; i < i_stop; ++i) {
        TraceRange range("mi_mg_sampler_step");
        const float t = ts_host[i];
        hipLaunchKernelGGL(fill_kernel, dim3(nblk(B)), dim3(256), 0, s, b->t_buf, t, (int64_t)B);
        StepArgs a;
        a.node_off = b->node_off;
        a.c = *c;
        a.seed = seed;
        a.step = (uint32_t)(i + 1);
        a.node_offset = b->node_offset;
        a.graph_offset = b->graph_offset;
        a.t = t;
        a.dt = dt;
        a.last = i + 1 >= n_steps;
        a.out_pos = b->sp_pos;
        a.out_cell = b->sp_cell;
        a.out_logits = b->sp_logits;
        a.pos = pos;
        a.cell = cell;
        a.types = types;
        a.mean_pos = mean_pos;
        a.mean_cell = mean_cell;
        // corrector
        MI_TRY(mi_gemnet_forward(net, b, pos, cell, types, b->t_buf, b->sp_pos, b->sp_cell, nullptr, fwd_flags, stream));
        a.n_pos = noise && noise->corr_pos ? noise->corr_pos + (size_t)i * N * 3 : nullptr;
        a.n_cell = noise && noise->corr_cell ? noise->corr_cell + (size_t)i * B * 9 : nullptr;
        a.n_u1 = a.n_u2 = nullptr;
        hipLaunchKernelGGL(mg_corrector_kernel, dim3(B), dim3(64), 0, s, a);
        // predictor
        MI_TRY(mi_gemnet_forward(net, b, pos, cell, types, b->t_buf, b->sp_pos, b->sp_cell, b->sp_logits, fwd_flags, stream));
        a.n_pos = noise && noise->pred_pos ? noise->pred_pos + (size_t)i * N * 3 : nullptr;
        a.n_cell = noise && noise->pred_cell ? noise->pred_cell + (size_t)i * B * 9 : nullptr;
        a.n_u1 = noise && noise->pred_u1 ? noise->pred_u1 + (size_t)i * N : nullptr;
        a.n_u2 = noise && noise->pred_u2 ? noise->pred_u2 + (size_t)i * N : nullptr;
        hipLaunchKernelGGL(mg_predictor_kernel, dim3(B), dim3(64), 0, s, a);
        MI_KERNEL_CHECK();
    }
    return MI_OK;
}

}  // extern "C"

// CSPNet score network on gfx950: weight packing, node-level kernels, forward orchestration
// and the mi_net / mi_batch C entry points.  Reference: models/diffcsp/cspnet.py.
#include <dlfcn.h>
#include <stdarg.h>

#include <algorithm>
#include <atomic>

#define MI_GEMM_OWNER   // this unit defines the GEMM launchers of gemm.h / gemm_split.h (and therefore carries their kernels); the others see prototypes
#include "edge_mlp.h"
#include "gemm_split.h"
#include "net.h"

namespace mi {

int g_gemm_mode = 1;  // MI_GEMM_SPLIT
int g_planes_variant = 0;  // 0: 128-row kernel everywhere (default: with three workgroups per CU it beats the 256-row double-buffered
                           // kernel in situ: 26.8 vs 25.2 structures/s on one stream); 1: 256-row kernel from g_planes_db_min_tiles up
int g_planes_db_min_tiles = 512;
int g_pair_kernel = 0;
int g_fold_pair_extras = 1;  // pair mode: activation scales and self edges ride in the launch of the Fourier-block GEMM (0: separate launches)
int g_planes_big = 1;             // large-M plain products on the 256 x 256 LDS-DMA kernel (gemm_split.h); the pinned path's launches are below the row limit
int g_planes_big_min_rows = 65536;
int g_pair_wide_force = 0;         // tests: the pair-mode epilogue's 64-bit addressing whatever the operand sizes (it is otherwise taken beyond 4 GB only)
int g_node_train = 1;             // the training forward's node-level work on the one-launch chain too (node_chain.hip writes the tape on the way); 0: seven launches per layer
int g_edge2_train = 1;            // the training forward's second edge GEMM on the register-tile kernel too, its pre-activation written row-major through LDS patches
int g_planes_rt = 2;              // register-tile kernel for every qualifying product (gemm_split.h / edge_stage.hip); 1 = only those with epilogue extensions
int g_planes_rt_min_rows = 16384;
int g_planes_big_seg_min_rows = 0;
int g_planes_dma = 1;
int g_heads_rows16_min_nodes = 1 << 30;   // atoms from which the fused heads run on 16 rows per workgroup instead of 4 (a quarter of the weight streams; mi_debug_set_heads_rows16).
                                          // OFF: measured neutral on the headline (53.9 / 54.0 / 54.4 against 53.7 / 53.8 / 54.2) and -2 % on the reference's default batch, profiles/r5_heads_rows16_ab.log
int g_eval_reuse = 7;   // bits 0, 1: what the sampler's predictor evaluation keeps from the corrector evaluation in front of it (the lattice term G; layer 0's projections);
                        // bit 2: the corrector evaluation computes the coordinate head alone (nothing else of it is read); mi_debug_set_eval_reuse
int g_fused_heads = 1;  // inference: coordinate and type heads in one launch (0: two fp32-operand GEMM launches; mi_debug_set_node_priority(2))
int g_node_hi = 0;  // 1: the node-level kernels of an inference forward on a helper stream of the highest priority (joined by events)
int g_planes_lat_max_blocks = 256;  // plane GEMMs of at most one workgroup per CU: the deep-prefetch latency form (gemm_split.h)
int g_planes_small_tiles = 0;  // off: with concurrent chains the 128-row tiles win (31.3 vs 30.8 structures/s); one chain alone gains 2.7 % from 64-row tiles
extern int g_bwd_pairs_fused, g_tn_xsilu, g_bwd_dz2_planes, g_bwd_wgrad_f16, g_bwd_pairs_tile, g_bwd_wgrad_planes, g_bwd_head_window;
int g_tn128 = 1;
int g_knn_nosync = 1;             // knn edge style inside the sampler's chain: the per-evaluation graph build without a host round trip (0: synchronising, as rounds 1-5)
int g_tn_target_tiles = 768;      // three workgroups per CU for a contraction that has the chip to itself
int g_concurrent_groups = 1;      // crystal groups the caller fine-tunes concurrently (mi_set_concurrent_groups): a contraction's share of the chip is 1 / this
int g_tn_split = 1;
int g_tn_split_min_rows = 4096;  // (at 5120 rows: 43 -> 39 us for 512 x 512, 71 -> 56 us for 512 x 1024; no gain below)
int g_edge_pairs = 1;  // first edge GEMM over unordered pairs (fc edge style, plane-GEMM edge stage)
int g_node_planes_min_rows = 512;  // node-level products: plane-set kernel from this many nodes up, fp32-operand split-K kernel below

// readers of the per-translation-unit saturation counters (gemm_split.h); a function-local static, because the registrars run
// during static initialisation of several units in unspecified order
std::vector<int (*)(unsigned*, bool)>& sat_readers() {
    static std::vector<int (*)(unsigned*, bool)> v;
    return v;
}
void sat_register(int (*fetch)(unsigned*, bool)) { sat_readers().push_back(fetch); }

int edge_gemm1(mi_net* net, const Planes& A, int layer, int M, PlanesEpilogue pe, hipStream_t s);   // edge_stage.hip

static bool cfg_ln_and_wide(const mi_net* n) { return n->cfg.ln && (n->H == 128 || n->H == 256 || n->H == 512); }

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("MI_ROCTX");
        if (!e || !*e || *e == '0') return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
Roctx& roctx() {
    static Roctx r;
    return r;
}
}  // namespace
void trace_push(const char* name) {
    if (roctx().push) roctx().push(name);
}
void trace_pop() {
    if (roctx().pop) roctx().pop();
}

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// issued matrix-pipe work of every product launched by this process (common.h): host-side counters, several enqueueing threads
static std::atomic<uint64_t> g_mfma16_mflop{0}, g_mfma32_mflop{0};   // in units of 1e6 flops (2^64 of them outlast any run)
void count_mfma(int64_t M, int64_t N, int64_t K, int terms16) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    const double f = 2.0 * (double)M * (double)N * (double)K * (terms16 > 0 ? terms16 : 1) * 1e-6;
    (terms16 > 0 ? g_mfma16_mflop : g_mfma32_mflop).fetch_add((uint64_t)(f + 0.5), std::memory_order_relaxed);
}

// ------------------------------------------------------------------------------------------
// Packing kernels: theta (nn.Linear [out,in]) -> MFMA-ready layouts
// ------------------------------------------------------------------------------------------
// Whh[2H][H]: rows [0,H) = W1[f][0:H], rows [H,2H) = W1[f][H:2H]   (W1 row stride = edge_in)
__global__ void pack_whh_kernel(const float* __restrict__ W1, int edge_in, int H, float* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * H * H) return;
    int r = idx / H, k = idx % H;
    out[idx] = r < H ? W1[(size_t)r * edge_in + k] : W1[(size_t)(r - H) * edge_in + H + k];
}

// Wff_p[m][t][lane][q]: feature f = 32t + (lane&31); pair s = 4m+q = c*FP + k (FP = KP/3, k >= F is
// zero padding); hi = lane>>5 selects the sin column (2H+9 + c*F + k) or the cos column
// (2H+9 + 3F + c*F + k) of W1.
__global__ void pack_wff_kernel(const float* __restrict__ W1, int edge_in, int H, int F, int KP, float* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int NT = H / 32;
    int total = (KP / 4) * NT * 256;
    if (idx >= total) return;
    int q = idx & 3, lane = (idx >> 2) & 63, t = (idx >> 8) % NT, m = (idx >> 8) / NT;
    int s = 4 * m + q, f = 32 * t + (lane & 31), hi = lane >> 5;
    int FP = KP / 3, c = s / FP, k = s % FP;
    float v = 0.f;
    if (k < F) v = W1[(size_t)f * edge_in + 2 * H + 9 + (hi ? 3 * F : 0) + c * F + k];
    out[idx] = v;
}

// Fourier operand of the edge kernel, once per network evaluation, in B-fragment order:
// FFp[tile][m][lane][q] = hi ? cos(arg) : sin(arg), arg = ((x_j - x_i) % 1)_c * (2*pi*k), pair 4m+q = c*FP + k,
// lane = (edge-in-tile, hi).  SinusoidsEmbedding (cspnet.py:12-24); frequency table 2*pi*k in fp32 (:16).
// frac_diff of edge e, coordinate c: explicit (knn branch) or (x_dst - x_src) % 1 (fc branch, cspnet.py:242)
__device__ __forceinline__ float edge_diff(const float* __restrict__ frac, const float* __restrict__ fd, int64_t e, int i, int j, int c) {
    return fd ? fd[e * 3 + c] : pymod1(frac[j * 3 + c] - frac[i * 3 + c]);
}

__global__ void fourier_pack_kernel(const float* __restrict__ frac, const float* __restrict__ fd, const int* __restrict__ src, const int* __restrict__ dst,
                                    float* __restrict__ FFp, int64_t E, int F, int KP) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per thread
    const int nm = KP / 4, FP = KP / 3;
    const int64_t total = ((E + 31) / 32) * nm * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), m = (int)((idx >> 6) % nm);
    const int64_t tile = (idx >> 6) / nm;
    int64_t e = tile * 32 + (lane & 31);
    if (e >= E) e = E - 1;
    const int hi = lane >> 5, i = src[e], j = dst[e];
    f32x4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = 4 * m + q, c = s / FP, k = s % FP;
        float v = 0.f;
        if (k < F) {
            const float d = edge_diff(frac, fd, e, i, j, c);
            float sn, cs;
            sincos_bounded(d * ((float)k * 6.28318530717958647692f), &sn, &cs);
            v = hi ? cs : sn;
        }
        out[q] = v;
    }
    reinterpret_cast<f32x4*>(FFp)[idx] = out;
}

// FF[e][c*F+k] = sin(d_c * 2*pi*k), FF[e][3F + c*F+k] = cos(...)   (cspnet.py:20-24)
__global__ void fourier_kernel(const float* __restrict__ frac, const float* __restrict__ fd, const int* __restrict__ src,
                               const int* __restrict__ dst, float* __restrict__ FF, int64_t E, int F) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * 3 * F) return;
    int64_t e = idx / (3 * F);
    int ck = (int)(idx % (3 * F)), c = ck / F, k = ck % F;
    float d = edge_diff(frac, fd, e, src[e], dst[e], c);
    float sn, cs;
    sincos_bounded(d * ((float)k * 6.28318530717958647692f), &sn, &cs);
    FF[e * (6 * F) + ck] = sn;
    FF[e * (6 * F) + 3 * F + ck] = cs;
}

// Fourier features written directly as a tile-blocked bf16 plane set (the A operand of the first edge GEMM).
// Columns = [sin(3F) | cos(3F)] in the reference order (cspnet.py:20-24).  One thread per (edge, pair of sin columns): the
// sine and cosine of an argument are produced together and stored to column ck and column 3F + ck.  Pad columns (>= 6F)
// and pad rows (>= E) are written as zero.  Needs 3F even (F even); the per-column kernel below covers odd F.
__global__ void fourier_planes_kernel(const float* __restrict__ frac, const float* __restrict__ fd, const int* __restrict__ src,
                                      const int* __restrict__ dst, Planes FF, int64_t E, int F, const int* __restrict__ e_dev = nullptr) {
    // (e_dev: the list's length lives on the device, knn_build nosync -- E is then its CAPACITY, the launch a fixed number of workgroups that stride
    //  over the rows that exist, instead of one thread per element of the capacity: 94 M mostly idle threads at 256 crystals)
    if (e_dev) E = min(E, (int64_t)*e_dev);
    const int F3 = 3 * F, np = F3 / 2, npad = (FF.KT * 32 - 2 * F3) / 2, per_row = np + npad;
    const int64_t rows_pad = (E + 127) / 128 * 128, total = rows_pad * per_row, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t e = idx / per_row;
        const int m = (int)(idx % per_row);
        if (m >= np) {  // zero padding of the K direction
            const int col = 2 * F3 + 2 * (m - np);
#pragma unroll
            for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(FF.base + FF.elem((int)e, col, k)) = 0u;
            continue;
        }
        float sn[2] = {0.f, 0.f}, cs[2] = {0.f, 0.f};
        if (e < E) {
            const int i = src[e], j = dst[e];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ck = 2 * m + u, c = ck / F, k = ck - c * F;
                const float d = edge_diff(frac, fd, e, i, j, c);
                sincos_bounded(d * ((float)k * 6.28318530717958647692f), &sn[u], &cs[u]);
            }
        }
        unsigned p[3];
        pl_split_pair(sn[0], sn[1], FF.s(), p);
#pragma unroll
        for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(FF.base + FF.elem((int)e, 2 * m, k)) = p[k];
        pl_split_pair(cs[0], cs[1], FF.s(), p);
#pragma unroll
        for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(FF.base + FF.elem((int)e, F3 + 2 * m, k)) = p[k];
    }
}

// per-column-pair form (any F)
__global__ void fourier_planes_cols_kernel(const float* __restrict__ frac, const float* __restrict__ fd, const int* __restrict__ src,
                                           const int* __restrict__ dst, Planes FF, int64_t E, int F, const int* __restrict__ e_dev = nullptr) {
    if (e_dev) E = min(E, (int64_t)*e_dev);
    const int cp = FF.KT * 16;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows_pad = (E + 127) / 128 * 128;
    if (idx >= rows_pad * cp) return;
    const int64_t e = idx / cp;
    const int c0 = (int)(idx % cp) * 2;
    float v[2] = {0.f, 0.f};
    if (e < E) {
        const int i = src[e], j = dst[e];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = c0 + u;
            if (col < 6 * F) {
                const int ck = col < 3 * F ? col : col - 3 * F, c = ck / F, k = ck % F;
                const float d = edge_diff(frac, fd, e, i, j, c);
                float sn, cs;
                sincos_bounded(d * ((float)k * 6.28318530717958647692f), &sn, &cs);
                v[u] = col < 3 * F ? sn : cs;
            }
        }
    }
    unsigned p[3];
    pl_split_pair(v[0], v[1], FF.s(), p);
#pragma unroll
    for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(FF.base + FF.elem((int)e, c0, k)) = p[k];
}

// Pair-mode Fourier operand: rows = unordered pairs (i, j), columns = [sin(3F) | 0 .. Kh) | cos(3F) | 0 .. 2Kh) of
// d = (x_j - x_i) % 1; pad rows / pad columns are written as zero.  A wave owns 16 consecutive rows of one 32-column k-tile
// (lane = row * 4 + 16-byte chunk): in the tile-blocked layout that is 1 KiB of contiguous bytes per plane, so every store
// instruction writes whole lines (four-byte stores of one row per wave ran at 2.2 TB/s).  Each lane evaluates eight sine /
// cosine pairs and stores them into the sine tile and the matching cosine tile.
template <bool F8>   // F8: F % 8 == 0 (the lane's eight columns share one coordinate)
__global__ __launch_bounds__(256) void fourier_pair_planes_kernel(const float* __restrict__ frac, const int* __restrict__ pi,
                                                                  const int* __restrict__ pj, Planes FF, int64_t Np, int F, int Kh,
                                                                  unsigned* __restrict__ zero = nullptr, int nzero = 0, int zero_keep = 0) {
    // (`zero`: the per-evaluation absmax slots, cleared here instead of by a memset launch of their own in front of the kernels that raise them;
    //  zero_keep bit 0: the odd slots stay -- they belong to the lattice term, which the caller keeps from the previous evaluation; bit 1: slot 0 stays --
    //  layer 0's projections, kept likewise)
    if (blockIdx.x == 0 && (int)threadIdx.x < nzero && !((zero_keep & 1) && (threadIdx.x & 1)) && !((zero_keep & 2) && threadIdx.x == 0)) zero[threadIdx.x] = 0u;
    const int F3 = 3 * F, kts = Kh / 32, lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t rows_pad = (Np + 127) / 128 * 128;
    const int64_t rb = w / kts;
    if (rb * 16 >= rows_pad) return;
    const int kt = (int)(w - rb * kts);
    const int64_t e = rb * 16 + (lane >> 2);
    const int col0 = kt * 32 + (lane & 3) * 8;
    float sn[8], cs[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sn[u] = cs[u] = 0.f;
    if (e < Np && col0 < F3) {
        const int i = pi[e], j = pj[e];
        if constexpr (F8) {
            // the lane's eight columns belong to ONE coordinate and are all in range (3F and col0 are multiples of eight): one division and one
            // difference per lane instead of eight of each -- the general loop below compiled to 1 134 VALU instructions per lane, this to a third
            // (same expression per element: bit-identical)
            const int c = col0 / F, k0 = col0 - c * F;
            const float d = pymod1(frac[j * 3 + c] - frac[i * 3 + c]);
#pragma unroll
            for (int u = 0; u < 8; ++u) sincos_bounded(d * ((float)(k0 + u) * 6.28318530717958647692f), &sn[u], &cs[u]);
        } else {
            int cprev = -1;
            float d = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ck = col0 + u;
                if (ck < F3) {
                    const int c = ck / F, k = ck - c * F;
                    if (c != cprev) {
                        d = pymod1(frac[j * 3 + c] - frac[i * 3 + c]);
                        cprev = c;
                    }
                    sincos_bounded(d * ((float)k * 6.28318530717958647692f), &sn[u], &cs[u]);
                }
            }
        }
    }
    u32x4 os[3], oc[3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned p[3];
        pl_split_pair(sn[2 * u], sn[2 * u + 1], FF.s(), p);
        os[0][u] = p[0];
        os[1][u] = p[1];
        os[2][u] = p[2];
        pl_split_pair(cs[2 * u], cs[2 * u + 1], FF.s(), p);
        oc[0][u] = p[0];
        oc[1][u] = p[1];
        oc[2][u] = p[2];
    }
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        *reinterpret_cast<u32x4*>(FF.base + FF.elem((int)e, col0, k)) = os[k];
        *reinterpret_cast<u32x4*>(FF.base + FF.elem((int)e, Kh + col0, k)) = oc[k];
    }
}

// plane set of the Fourier block of edge_mlp.0 in the pair-mode column layout; C0[f] = sum of its cosine block
__global__ void pack_wff_pair_planes_kernel(const float* __restrict__ W1, int edge_in, int H, int F, int Kh, Planes dst) {
    const int F3 = 3 * F, per_row = Kh;  // column pairs per row over 2*Kh columns
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows_pad = (int64_t)(H + 127) / 128 * 128;
    if (idx >= rows_pad * per_row) return;
    const int f = (int)(idx / per_row), col = (int)(idx % per_row) * 2;
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = col + u, blk = c >= Kh, cc = blk ? c - Kh : c;
        if (f < H && cc < F3) v[u] = W1[(size_t)f * edge_in + 2 * H + 9 + blk * F3 + cc];
    }
    unsigned p[3];
    pl_split_pair(v[0], v[1], dst.s(), p);
#pragma unroll
    for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(dst.base + dst.elem(f, col, k)) = p[k];
}
// C0[f] = sum of the cosine block of row f of edge_mlp.0's Fourier columns: one wave per row, lanes over the 3F columns, fixed-order
// shuffle reduction (a thread per row walking 3F floats a row stride apart took 100 us per layer at every parameter update).
__global__ __launch_bounds__(256) void wff_cos_rowsum_kernel(const float* __restrict__ W1, int edge_in, int H, int F, float* __restrict__ C0) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= H) return;
    const float* w = W1 + (size_t)f * edge_in + 2 * H + 9 + 3 * F;
    float s = 0.f;
    for (int k = lane; k < 3 * F; k += 64) s += w[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) C0[f] = s;
}

// Self edges (i, i) of the fc list in pair mode: d = 0, so the Fourier term is the constant C0:
//   Z1 = P_i[i] + P_j[i] + G[graph] + C0;  M1 = SiLU(Z1) written as planes at row e_diag[i]  (cspnet.py:59-79)
__global__ void edge_diag_kernel(const float* __restrict__ PQ, const float* __restrict__ G, const float* __restrict__ C0,
                                 const int* __restrict__ node2graph, const int* __restrict__ e_diag, float* __restrict__ pre_act, Planes M1,
                                 int N, int H, int ldpq) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * (H / 2)) return;
    const int i = (int)(idx / (H / 2)), f = (int)(idx % (H / 2)) * 2, e = e_diag[i], g = node2graph[i];
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
        v[u] = C0[f + u] + ((PQ[(size_t)i * ldpq + f + u] + PQ[(size_t)i * ldpq + H + f + u]) + G[(size_t)g * H + f + u]);
    if (pre_act) {
        pre_act[(size_t)e * H + f] = v[0];
        pre_act[(size_t)e * H + f + 1] = v[1];
    }
    unsigned p[3];
    pl_split_pair(silu_fast(v[0]), silu_fast(v[1]), M1.s(), p);
#pragma unroll
    for (int k = 0; k < NPL; ++k) *reinterpret_cast<unsigned*>(M1.base + M1.elem(e, f, k)) = p[k];
}

// Wff[f][0:6F] = W1[f][2H+9 : 2H+9+6F]  (contiguous, 16-byte aligned rows for the GEMM path)
__global__ void pack_wff_plain_kernel(const float* __restrict__ W1, int edge_in, int H, int F6, float* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * F6) return;
    int f = idx / F6, k = idx % F6;
    out[idx] = W1[(size_t)f * edge_in + 2 * H + 9 + k];
}

// agg[i] = mean over the edge run of node i of M2[e]  -> cat[i][H:2H]     (scatter mean, cspnet.py:79)
__global__ void segment_mean_kernel(const float* __restrict__ M2, const int* __restrict__ rowptr, float* __restrict__ cat, int N, int H) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * H) return;
    int i = (int)(idx / H), f = (int)(idx % H);
    int e0 = rowptr[i], e1 = rowptr[i + 1];
    float s = 0.f;
    for (int e = e0; e < e1; ++e) s += M2[(size_t)e * H + f];
    cat[(size_t)i * (2 * H) + H + f] = e1 > e0 ? s / (float)(e1 - e0) : 0.f;
}

// W2_p[u][t][q][lane][c] = W2[32u + (lane&31)][32t + 8q + 4(lane>>5) + c]
__global__ void pack_w2_kernel(const float* __restrict__ W2, int H, float* __restrict__ out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int NT = H / 32;
    if (idx >= H * H) return;
    int c = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3, t = (idx >> 10) % NT, u = (idx >> 10) / NT;
    out[idx] = W2[(size_t)(32 * u + (lane & 31)) * H + 32 * t + 8 * q + 4 * (lane >> 5) + c];
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim (one wave per row).  y has row stride ldy (writes into cat[:, :H]).
// ------------------------------------------------------------------------------------------
// ypl (optional): the same values as a bf16 plane set (the A operand of the node-level GEMMs).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bsh, float* __restrict__ y, int ldy,
                                                        float* __restrict__ stats, int N, int H, Planes ypl = Planes()) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = x + (size_t)row * H;
    // H = 8 x 64 or less, 16-byte aligned operands: a lane owns eight CONSECUTIVE columns -- two 16-byte loads, two 16-byte stores and one
    // 16-byte store per plane instead of eight 4-byte loads / stores and sixteen 2-byte plane stores (the kernel sits on every chain's
    // serial path 14 times per evaluation, beside the other chains' GEMM workgroups)
    if (H <= 512 && (H & 7) == 0 && (ldy & 3) == 0 &&
        (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bsh | (uintptr_t)y) & 15) == 0) {
        const int c0 = lane * 8;
        const bool act = c0 < H;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            a = *reinterpret_cast<const f32x4*>(xr + c0);
            b = *reinterpret_cast<const f32x4*>(xr + c0 + 4);
        }
        const float mean = wave_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) / (float)H;
        float q = 0.f;
        if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float da = a[k] - mean, db = b[k] - mean;
                q += da * da + db * db;
            }
        }
        const float var = wave_sum(q) / (float)H;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (act) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c0), w1 = *reinterpret_cast<const f32x4*>(w + c0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bsh + c0), b1 = *reinterpret_cast<const f32x4*>(bsh + c0 + 4);
            f32x4 o0, o1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o0[k] = (a[k] - mean) * rstd * w0[k] + b0[k];
                o1[k] = (b[k] - mean) * rstd * w1[k] + b1[k];
            }
            float* yr = y + (size_t)row * ldy + c0;
            *reinterpret_cast<f32x4*>(yr) = o0;
            *reinterpret_cast<f32x4*>(yr + 4) = o1;
            if (ypl.base) {
                const float ps = ypl.s();
                u32x4 pk[3];
                unsigned pr[3];
                pl_split_pair(o0[0], o0[1], ps, pr); pk[0][0] = pr[0]; pk[1][0] = pr[1]; pk[2][0] = pr[2];
                pl_split_pair(o0[2], o0[3], ps, pr); pk[0][1] = pr[0]; pk[1][1] = pr[1]; pk[2][1] = pr[2];
                pl_split_pair(o1[0], o1[1], ps, pr); pk[0][2] = pr[0]; pk[1][2] = pr[1]; pk[2][2] = pr[2];
                pl_split_pair(o1[2], o1[3], ps, pr); pk[0][3] = pr[0]; pk[1][3] = pr[1]; pk[2][3] = pr[2];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(ypl.base + ypl.elem(row, c0, pl)) = pk[pl];
            }
        }
        if (stats && lane == 0) {
            stats[2 * row] = mean;
            stats[2 * row + 1] = rstd;
        }
        return;
    }
    float v[8];
    float s = 0.f;
    int cnt = 0;
    for (int c = lane; c < H; c += 64) {
        v[cnt] = xr[c];
        s += v[cnt++];
    }
    float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int k = 0; k < cnt; ++k) {
        float d = v[k] - mean;
        q += d * d;
    }
    float var = wave_sum(q) / (float)H;
    float rstd = 1.0f / sqrtf(var + 1e-5f);
    float* yr = y + (size_t)row * ldy;
    cnt = 0;
    for (int c = lane; c < H; c += 64) {
        const float o = (v[cnt++] - mean) * rstd * w[c] + bsh[c];
        yr[c] = o;
        if (ypl.base) {
            u16 p0, p1, p2;
            pl_split(o, ypl.s(), p0, p1, p2);
            ypl.base[ypl.elem(row, c, 0)] = p0;
            ypl.base[ypl.elem(row, c, 1)] = p1;
            ypl.base[ypl.elem(row, c, 2)] = p2;
        }
    }
    if (stats && lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

__global__ void copy_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int ldy, int N, int H, Planes ypl = Planes()) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * H) return;
    int r = (int)(idx / H), c = (int)(idx % H);
    y[(size_t)r * ldy + c] = x[idx];
    if (ypl.base) {
        u16 p0, p1, p2;
        pl_split(x[idx], ypl.s(), p0, p1, p2);
        ypl.base[ypl.elem(r, c, 0)] = p0;
        ypl.base[ypl.elem(r, c, 1)] = p1;
        ypl.base[ypl.elem(r, c, 2)] = p2;
    }
}

// G[l][b][f] = b1_l[f] + sum_m (L L^T)[b].flat[m] * W1_l[f][2H + m]      (cspnet.py:68-72), every layer in one launch (the lattices
// do not change inside an evaluation); layer l's edge_mlp.0 weight / bias sit `layer_stride` floats after layer l-1's in the flat
// parameter vector.
constexpr int GRAM_GB = 8;  // crystals per workgroup of gram_term_all_kernel
// grid (ceil(B / GRAM_GB), L), 256 threads over the features: a thread reads its nine weights once (rows of edge_mlp.0 are
// edge_in floats apart, so this is a gather) and reuses them for GRAM_GB crystals; one atomicMax per workgroup.
__global__ __launch_bounds__(256) void gram_term_all_kernel(const float* __restrict__ lattices, const float* __restrict__ W1_0, int64_t layer_stride,
                                                            int edge_in, const float* __restrict__ b1_0, float* __restrict__ G, int H, int B,
                                                            unsigned* __restrict__ gmax) {
    const int b0 = blockIdx.x * GRAM_GB, l = blockIdx.y, nb = min(GRAM_GB, B - b0);
    __shared__ float gram[GRAM_GB][9];
    __shared__ float wmax[4];
    if (threadIdx.x < 9 * nb) {
        const int g = threadIdx.x / 9, m = threadIdx.x % 9, r = m / 3, c = m % 3;
        const float* Lm = lattices + (size_t)(b0 + g) * 9;
        gram[g][m] = Lm[r * 3] * Lm[c * 3] + Lm[r * 3 + 1] * Lm[c * 3 + 1] + Lm[r * 3 + 2] * Lm[c * 3 + 2];
    }
    __syncthreads();
    const float* W1 = W1_0 + l * layer_stride;
    const float* b1 = b1_0 + l * layer_stride;
    float gm = 0.f;
    for (int f = threadIdx.x; f < H; f += blockDim.x) {
        const float* w = W1 + (size_t)f * edge_in + 2 * H;
        float wv[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) wv[m] = w[m];
        const float bf = b1[f];
        for (int g = 0; g < nb; ++g) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 9; ++m) s += gram[g][m] * wv[m];
            G[((size_t)l * B + b0 + g) * H + f] = s + bf;
            gm = fmaxf(gm, fabsf(s + bf));
        }
    }
    if (gmax) {  // max |G[l]| over the batch (order-independent): part of the bound that scales the M1 plane set
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = gm;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(gmax + 2 * l, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    }
}

// max |x| over n floats -> atomicMax of the bit pattern (non-negative floats order like unsigned integers)
__global__ void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// wb[l][0..4] = max_f sum_k |Wff[f][k]| (Fourier block of edge_mlp.0), max_f sum_k |W2[f][k]|, max |b2|,
//               max_f sum_k |node_mlp.0.weight[f][H + k]|, max |node_mlp.0.bias|      (at parameter updates)
// grid (L, ceil(H / 64)): a wave per output row group, four rows at a time; maxima merged with atomicMax on the bit patterns
// (non-negative floats order like unsigned integers; wb is zeroed first).
__global__ __launch_bounds__(256) void weight_bounds_kernel(const float* __restrict__ W1_0, const float* __restrict__ W2_0,
                                                            const float* __restrict__ b2_0, const float* __restrict__ Wn0_0,
                                                            const float* __restrict__ bn0_0, int64_t layer_stride, int edge_in, int H, int F6,
                                                            unsigned* __restrict__ wb) {
    const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *W1 = W1_0 + l * layer_stride, *W2 = W2_0 + l * layer_stride, *b2 = b2_0 + l * layer_stride, *Wn0 = Wn0_0 + l * layer_stride,
                *bn0 = bn0_0 + l * layer_stride;
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int f = blockIdx.y * 64 + wave; f < min(H, (int)blockIdx.y * 64 + 64); f += 4) {  // one row per wave, lanes over its columns
        float a = 0.f, c = 0.f, d = 0.f;
        for (int k = lane; k < F6; k += 64) a += fabsf(W1[(size_t)f * edge_in + 2 * H + 9 + k]);
        for (int k = lane; k < H; k += 64) {
            c += fabsf(W2[(size_t)f * H + k]);
            d += fabsf(Wn0[(size_t)f * 2 * H + H + k]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            c += __shfl_xor(c, o, 64);
            d += __shfl_xor(d, o, 64);
        }
        m[0] = fmaxf(m[0], a);
        m[1] = fmaxf(m[1], c);
        m[2] = fmaxf(m[2], fabsf(b2[f]));
        m[3] = fmaxf(m[3], d);
        m[4] = fmaxf(m[4], fabsf(bn0[f]));
    }
    if (lane < 5) {
        float v = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : lane == 3 ? m[3] : m[4];
        atomicMax(wb + l * 8 + lane, __float_as_uint(v * 1.0001f));  // (the row sums are rounded: a hair of slack keeps the bound a bound)
    }
}

// Per layer, after the LayerNorm(h) product: the scales of act_scales_eval (gemm_split.h) as a launch of its own -- the
// paths that do not fold them into the pair-mode edge GEMM.  pq / gmax: this layer's slots of the absmax array.
__global__ void act_scales_kernel(const unsigned* __restrict__ pq, const unsigned* __restrict__ gmax, const float* __restrict__ wb, float* __restrict__ dsc) {
    float d[6];
    act_scales_eval(__uint_as_float(*pq), __uint_as_float(*gmax), wb, d);
    for (int c = 0; c < 6; ++c) dsc[c] = d[c];
}


// agg[i] = (sum of this node's slots) / degree  -> cat[i][H:2H]       (scatter mean, cspnet.py:79)
// aggpl (optional): the same values as a bf16 plane set (N x H).
__global__ void finalize_agg_kernel(const float* __restrict__ part, const int* __restrict__ rowptr,
                                    float* __restrict__ cat, int N, int H, Planes aggpl = Planes(), int seg_shift = 5) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((H & 7) == 0 && (((uintptr_t)part | (uintptr_t)cat) & 15) == 0) {   // eight consecutive columns per thread: 16-byte accesses throughout
        const int h8 = H >> 3;
        if (idx >= (int64_t)N * h8) return;
        const int i = (int)(idx / h8), f = (int)(idx % h8) * 8;
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (e1 > e0) {
            const int t0 = e0 >> seg_shift, t1 = (e1 - 1) >> seg_shift;   // (row blocks behind the partial sums: 32 rows, or 128 from edge_stage.hip)
            for (int sl = 0; sl <= t1 - t0; ++sl) {
                const float* p = part + ((size_t)sl * N + i) * H + f;
                a += *reinterpret_cast<const f32x4*>(p);
                b += *reinterpret_cast<const f32x4*>(p + 4);
            }
            const float d = (float)(e1 - e0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = a[k] / d;
                b[k] = b[k] / d;
            }
        }
        float* o = cat + (size_t)i * (2 * H) + H + f;
        *reinterpret_cast<f32x4*>(o) = a;
        *reinterpret_cast<f32x4*>(o + 4) = b;
        if (aggpl.base) {
            const float ps = aggpl.s();
            u32x4 pk[3];
            unsigned pr[3];
            pl_split_pair(a[0], a[1], ps, pr); pk[0][0] = pr[0]; pk[1][0] = pr[1]; pk[2][0] = pr[2];
            pl_split_pair(a[2], a[3], ps, pr); pk[0][1] = pr[0]; pk[1][1] = pr[1]; pk[2][1] = pr[2];
            pl_split_pair(b[0], b[1], ps, pr); pk[0][2] = pr[0]; pk[1][2] = pr[1]; pk[2][2] = pr[2];
            pl_split_pair(b[2], b[3], ps, pr); pk[0][3] = pr[0]; pk[1][3] = pr[1]; pk[2][3] = pr[2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(aggpl.base + aggpl.elem(i, f, pl)) = pk[pl];
        }
        return;
    }
    if (idx >= (int64_t)N * H) return;
    int i = (int)(idx / H), f = (int)(idx % H);
    int e0 = rowptr[i], e1 = rowptr[i + 1];
    float s = 0.f;
    if (e1 > e0) {
        int t0 = e0 >> seg_shift, t1 = (e1 - 1) >> seg_shift;
        for (int sl = 0; sl <= t1 - t0; ++sl) s += part[((size_t)sl * N + i) * H + f];
        s = s / (float)(e1 - e0);
    }
    cat[(size_t)i * (2 * H) + H + f] = s;
    if (aggpl.base) {
        u16 p0, p1, p2;
        pl_split(s, aggpl.s(), p0, p1, p2);
        aggpl.base[aggpl.elem(i, f, 0)] = p0;
        aggpl.base[aggpl.elem(i, f, 1)] = p1;
        aggpl.base[aggpl.elem(i, f, 2)] = p2;
    }
}

// Coordinate and type heads of an inference forward in ONE launch (cspnet.py:276-279: coord_out (no bias) and type_out on the final
// LayerNorm's rows): the two [N, 3] / [N, 100] products were two fp32-operand GEMM launches plus their split-K reductions on every chain's
// serial path (~60 us per evaluation for 67 MFLOP).  Four rows per block in LDS, a thread per output column, the 103 weight rows read
// transposed ([H][104]: coalesced), plain fp32 FMA chains over k.
constexpr int HEADS_LD = 104;
__global__ void pack_heads_kernel(const float* __restrict__ Wc, const float* __restrict__ Wt, float* __restrict__ WT, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * HEADS_LD) return;
    const int k = idx / HEADS_LD, c = idx % HEADS_LD;
    WT[idx] = c < 3 ? Wc[(size_t)c * H + k] : c < 3 + MI_NUM_TYPES ? Wt[(size_t)(c - 3) * H + k] : 0.f;
}
// HEADS_ROWS rows per workgroup: every workgroup streams the whole [H][104] weight block (213 KB at H = 512) from L2, so 4 rows per workgroup made the
// launch 320 such streams at 1 280 atoms (47 us beside the other chains' GEMMs, on every chain's serial path twice per step); 16 rows make it 80.  The
// per-output fp32 FMA chains are unchanged (same k order, same four partial sums): the result does not depend on HEADS_ROWS.
template <int HEADS_ROWS>
__global__ __launch_bounds__(512) void heads_kernel(const float* __restrict__ hf, const float* __restrict__ WT, const float* __restrict__ type_bias,
                                                    float* __restrict__ coord_out, float* __restrict__ type_out, int N, int H, int ncols = 3 + MI_NUM_TYPES) {
    // (ncols = 3: the coordinate head alone -- the sampler's corrector evaluation reads nothing else, diffusion.py:310-322; same FMA chains for those columns)
    // 512 threads = 4 k-groups x 128 output columns (103 used): a k-group walks a quarter of H with eight weight loads in flight per
    // step (a single group with four was a chain of 128 load latencies: slower than the two GEMM launches it replaced); the four partial
    // sums of an output are added in a fixed order through LDS.
    extern __shared__ __attribute__((aligned(16))) float hs[];   // [HEADS_ROWS][H] | partial sums [4][HEADS_ROWS][128]
    float* red = hs + HEADS_ROWS * H;
    const int r0 = blockIdx.x * HEADS_ROWS, tid = threadIdx.x, t = tid & 127, kgp = tid >> 7;
    for (int i = tid; i < HEADS_ROWS * H; i += 512) {
        const int r = r0 + i / H;
        hs[i] = r < N ? hf[(size_t)r * H + i % H] : 0.f;
    }
    __syncthreads();
    float acc[HEADS_ROWS];
#pragma unroll
    for (int r = 0; r < HEADS_ROWS; ++r) acc[r] = 0.f;
    const int kq = H >> 2, k0 = kgp * kq, k1 = k0 + kq;
    if (t < ncols) {
        for (int k = k0; k < k1; k += 16) {   // (H % 64 == 0: a k-group's quarter is a multiple of 16)
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = WT[(size_t)(k + u) * HEADS_LD + t];
#pragma unroll
            for (int r = 0; r < HEADS_ROWS; ++r) {
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const f32x4 h = *reinterpret_cast<const f32x4*>(hs + r * H + k + 4 * v4);
                    acc[r] = fmaf(w[4 * v4], h[0], acc[r]);
                    acc[r] = fmaf(w[4 * v4 + 1], h[1], acc[r]);
                    acc[r] = fmaf(w[4 * v4 + 2], h[2], acc[r]);
                    acc[r] = fmaf(w[4 * v4 + 3], h[3], acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HEADS_ROWS; ++r) red[(kgp * HEADS_ROWS + r) * 128 + t] = acc[r];
    __syncthreads();
    if (kgp != 0 || t >= ncols) return;
#pragma unroll
    for (int r = 0; r < HEADS_ROWS; ++r) {
        const int row = r0 + r;
        if (row >= N) break;
        const float v = ((red[(0 * HEADS_ROWS + r) * 128 + t] + red[(1 * HEADS_ROWS + r) * 128 + t]) + red[(2 * HEADS_ROWS + r) * 128 + t]) +
                        red[(3 * HEADS_ROWS + r) * 128 + t];
        if (t < 3) coord_out[(size_t)row * 3 + t] = v;
        else type_out[(size_t)row * MI_NUM_TYPES + (t - 3)] = v + type_bias[t - 3];
    }
}

// graph mean-pool + lattice head:  out[b] = reshape(Wl * mean_i hf_i, 3, 3) @ L_b  (cspnet.py:281-289)
__global__ __launch_bounds__(256) void lattice_head_kernel(const float* __restrict__ hf, const int* __restrict__ node_off,
                                                           const float* __restrict__ Wl, const float* __restrict__ lattices,
                                                           float* __restrict__ out, float* __restrict__ gf_out, int H) {
    // One block per crystal, on every chain's serial path twice per step.  Everything that does not depend on the pooled features is
    // requested first (the head's weight rows of this wave, the lattice), the pooling runs as two independent half-sums of 16-byte
    // loads per thread instead of one serial walk over the atoms, and the nine dot products follow from registers.
    int b = blockIdx.x;
    extern __shared__ float sm[];  // gf [H] | lo [9] | second half-sums [H]
    float* gf = sm;
    float* lo = sm + H;
    float* half2 = sm + H + 12;
    const int n0 = node_off[b], n1 = node_off[b + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool fast = H <= 512 && (H & 7) == 0 && ((uintptr_t)hf & 15) == 0;
    float wreg[3][8];
    if (fast) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int m = wave + 4 * q;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = lane + 64 * j;
                wreg[q][j] = (m < 9 && f < H) ? Wl[(size_t)m * H + f] : 0.f;
            }
        }
        float lm[3] = {0.f, 0.f, 0.f};
        if (threadIdx.x < 9) {
            const float* Lm = lattices + (size_t)b * 9;
            const int c = threadIdx.x % 3;
            lm[0] = Lm[c];
            lm[1] = Lm[3 + c];
            lm[2] = Lm[6 + c];
        }
        const int h4 = H >> 2, grp = threadIdx.x >= 128 ? 1 : 0, f4 = threadIdx.x & 127;
        const int mid = n0 + (n1 - n0 + 1) / 2, ia = grp ? mid : n0, ib = grp ? n1 : mid;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        if (f4 < h4)
            for (int i = ia; i < ib; ++i) s4 += *reinterpret_cast<const f32x4*>(hf + (size_t)i * H + 4 * f4);
        if (grp == 1 && f4 < h4) *reinterpret_cast<f32x4*>(half2 + 4 * f4) = s4;
        __syncthreads();
        if (grp == 0 && f4 < h4) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(half2 + 4 * f4);
            const float cnt = (float)(n1 - n0), d = cnt < 1.f ? 1.f : cnt;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = (s4[k] + o[k]) / d;
                gf[4 * f4 + k] = v;
                if (gf_out) gf_out[(size_t)b * H + 4 * f4 + k] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int m = wave + 4 * q;
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = lane + 64 * j;
                if (f < H) sacc += wreg[q][j] * gf[f];
            }
            sacc = wave_sum(sacc);
            if (lane == 0 && m < 9) lo[m] = sacc;
        }
        __syncthreads();
        if (threadIdx.x < 9) {
            const int r = threadIdx.x / 3;
            out[(size_t)b * 9 + threadIdx.x] = lo[r * 3] * lm[0] + lo[r * 3 + 1] * lm[1] + lo[r * 3 + 2] * lm[2];
        }
        return;
    }
    for (int f = threadIdx.x; f < H; f += blockDim.x) {
        float s = 0.f;
        for (int i = n0; i < n1; ++i) s += hf[(size_t)i * H + f];
        float cnt = (float)(n1 - n0);
        gf[f] = s / (cnt < 1.f ? 1.f : cnt);
        if (gf_out) gf_out[(size_t)b * H + f] = gf[f];
    }
    __syncthreads();
    for (int m = wave; m < 9; m += 4) {
        float s = 0.f;
        for (int f = lane; f < H; f += 64) s += Wl[(size_t)m * H + f] * gf[f];
        s = wave_sum(s);
        if (lane == 0) lo[m] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        int r = threadIdx.x / 3, c = threadIdx.x % 3;
        const float* Lm = lattices + (size_t)b * 9;
        out[(size_t)b * 9 + threadIdx.x] = lo[r * 3] * Lm[c] + lo[r * 3 + 1] * Lm[3 + c] + lo[r * 3 + 2] * Lm[6 + c];
    }
}

template <typename T>
int dev_alloc(mi_batch* b, T** p, size_t n) {
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
template int dev_alloc<float>(mi_batch*, float**, size_t);
template int dev_alloc<int>(mi_batch*, int**, size_t);
template int dev_alloc<unsigned short>(mi_batch*, unsigned short**, size_t);
template int dev_alloc<unsigned>(mi_batch*, unsigned**, size_t);

// bench.py's roofline hook: bracket the dominant stage (the per-edge MLP of one layer) with events
struct ProfSlot {
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
static int prof_begin(mi_net* net, hipStream_t s, ProfSlot* slot) {
    *slot = ProfSlot();
    if (!net->prof) return MI_OK;
    {
        std::lock_guard<std::mutex> g(net->prof_mu);
        if (net->ev_used + 2 > net->ev.size())
            for (int k = 0; k < 2; ++k) {
                hipEvent_t ev;
                MI_HIP(hipEventCreate(&ev));
                net->ev.push_back(ev);
            }
        slot->e0 = net->ev[net->ev_used];
        slot->e1 = net->ev[net->ev_used + 1];
        net->ev_used += 2;
    }
    MI_HIP(hipEventRecord(slot->e0, s));
    return MI_OK;
}
static int prof_end(mi_net* net, hipStream_t s, const ProfSlot& slot) {
    if (slot.e1) MI_HIP(hipEventRecord(slot.e1, s));
    return MI_OK;
}

static int launch_edge(mi_net* net, mi_batch* b, int layer, const float* frac, float* Z1, float* Z2, hipStream_t s) {
    const std::string p = "csp_layer_" + std::to_string(layer) + ".";
    EdgeFwdArgs a;
    a.PQ = b->PQ;
    a.G = b->G + (size_t)layer * b->B * b->H;
    a.FFp = b->FFp;
    a.src = b->src;
    a.dst = b->dst;
    a.node2graph = b->node2graph;
    a.rowptr = b->rowptr;
    a.Wff_p = net->Wff_p + layer * net->wff_stride();
    a.W2_p = net->W2_p + layer * net->w2_stride();
    a.b2 = net->p(p + "edge_mlp.2.bias");
    a.part = b->part;
    a.Z1 = Z1;
    a.Z2 = Z2;
    a.dbg = nullptr;
#ifdef MI_TIMING
    static unsigned long long* g_dbg = nullptr;
    static int g_count = 0;
    const size_t ntile = (size_t)cdiv(b->E, 32);
    if (!g_dbg) MI_HIP(hipMalloc((void**)&g_dbg, (size_t)1 << 24));
    a.dbg = g_dbg;
#endif
    a.E = b->E;
    a.N = b->N;
    a.F = net->F;
    a.KP = net->KP;
    if (b->E == 0) return MI_OK;
    dim3 grid((unsigned)cdiv(b->E, 32)), block(64);
    ProfSlot ps;
    MI_TRY(prof_begin(net, s, &ps));
    const bool save = Z1 != nullptr;
    MI_CHECK((Z1 == nullptr) == (Z2 == nullptr), MI_EINVAL, "Z1 and Z2 must be given together");
    count_mfma(b->E, net->H, 2 * net->KP + net->H, 0);   // (the register-chained f32-MFMA edge stage: Fourier block over KP (sin, cos) pairs + second linear)
#define MI_EDGE_LAUNCH(HH)                                                                   \
    case HH:                                                                                 \
        if (save) hipLaunchKernelGGL((edge_mlp_fwd_kernel<HH, true>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((edge_mlp_fwd_kernel<HH, false>), grid, block, 0, s, a);     \
        break;
    switch (net->H) {
        MI_EDGE_LAUNCH(64)
        MI_EDGE_LAUNCH(128)
        MI_EDGE_LAUNCH(256)
        MI_EDGE_LAUNCH(512)
        default: set_error("unsupported hidden_dim %d", net->H); return MI_EINVAL;
    }
#undef MI_EDGE_LAUNCH
    MI_KERNEL_CHECK();
    MI_TRY(prof_end(net, s, ps));
#ifdef MI_TIMING
    if (++g_count == 40 && ntile * 128 <= ((size_t)1 << 24)) {
        std::vector<unsigned long long> h(ntile * 16);
        MI_HIP(hipStreamSynchronize(s));
        MI_HIP(hipMemcpy(h.data(), g_dbg, h.size() * 8, hipMemcpyDeviceToHost));
        const char* nm[8] = {"gemm1_A", "finish_A", "gemm1_B", "pass0_mfma", "pass0_epi", "pass1_mfma", "pass1_epi", "passes2.."};
        const int lo[8] = {0, 1, 2, 3, 4, 5, 6, 7}, hiX[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        double tot = 0;
        for (int k = 0; k < 8; ++k) {
            double sum = 0;
            for (size_t t = 0; t < ntile; ++t) sum += (double)(h[t * 16 + hiX[k]] - h[t * 16 + lo[k]]);
            fprintf(stderr, "[MI_TIMING] %-12s %10.0f clk/tile\n", nm[k], sum / ntile);
        }
        double whole = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t t = 0; t < ntile; ++t) { whole += (double)(h[t * 16 + 8] - h[t * 16]); tmin = std::min(tmin, h[t * 16]); tmax = std::max(tmax, h[t * 16 + 8]); }
        fprintf(stderr, "[MI_TIMING] whole tile %10.0f clk; kernel span %llu clk; tiles %zu\n", whole / ntile, tmax - tmin, ntile);
    }
#endif
    return MI_OK;
}

int net_forward(mi_net* net, mi_batch* b, const float* t_emb, const float* atom_types, const float* frac,
                const float* lattices, float* lattice_out, float* coord_out, float* type_out, hipStream_t s, bool train, bool reuse_embedding, bool coords_only) {
    MI_CHECK(net->theta != nullptr, MI_ESTATE, "mi_cspnet_forward before mi_net_set_params");
    const int H = net->H, L = net->L, N = b->N, B = b->B, TD = net->TD;
    if (N == 0 || B == 0) return MI_OK;
    const size_t NH = (size_t)N * H;
    Tape& tp = b->tape;
    tp.valid = false;  // this forward overwrites h / hf / x1, which a pending backward would read
    // cspnet.py:243-257: the edge list follows the coordinates.  Inside the sampler's chain (mi_batch::knn_nosync) on the plane-GEMM path the build does not
    // synchronise: b->E is then the capacity, b->e_dev the device-side edge count every consumer below takes its row count from
    if (b->knn) MI_TRY(knn_build(b, frac, lattices, s, b->knn_nosync && !train && g_knn_nosync && MI_PLANES_FP16 && g_gemm_mode == MI_GEMM_SPLIT && net->edge_mode != 0));
    // Head / embedding weight gradients deferred over a window of micro-steps (Tape::w_hf ...): this forward's x1 and hf ARE the window slot's rows
    const bool hw = train && tp.allocated && tp.head_window();
    const size_t hslot = hw ? (size_t)tp.wcur : 0;
    b->x1 = hw ? tp.w_x1 + hslot * N * H : b->x1_base;
    b->hf = hw ? tp.w_hf + hslot * N * H : b->hf_base;
    if (train) {
        MI_CHECK(tp.allocated, MI_ESTATE, "training forward without tape");
        if (tp.borrow_inputs) {   // (the fused micro-step's own arrays: see Tape::in_types; with an open window it hands over the slot's rows)
            tp.in_types = atom_types, tp.in_temb = t_emb, tp.in_lat = lattices, tp.in_frac = frac;
        } else {
            float* const types_dst = hw ? tp.w_types + hslot * N * MI_NUM_TYPES : tp.atom_types;   // (what the deferred contraction reads must outlive this micro-step)
            float* const temb_dst = hw ? tp.w_temb + hslot * B * TD : tp.t_emb;
            MI_HIP(hipMemcpyAsync(types_dst, atom_types, (size_t)N * MI_NUM_TYPES * 4, hipMemcpyDeviceToDevice, s));
            MI_HIP(hipMemcpyAsync(temb_dst, t_emb, (size_t)B * TD * 4, hipMemcpyDeviceToDevice, s));
            MI_HIP(hipMemcpyAsync(tp.lattices, lattices, (size_t)B * 9 * 4, hipMemcpyDeviceToDevice, s));
            MI_HIP(hipMemcpyAsync(tp.frac, frac, (size_t)N * 3 * 4, hipMemcpyDeviceToDevice, s));
            tp.in_types = types_dst, tp.in_temb = temb_dst, tp.in_lat = tp.lattices, tp.in_frac = tp.frac;
        }
    }
    // ---- embedding (cspnet.py:265-271) ----
    // (reuse_embedding: same t_emb and atom_types as this batch's previous evaluation -- the sampler's predictor evaluation
    // differs from its corrector evaluation only in the coordinates -- so b->h[0] is still valid)
    if (!reuse_embedding) {
        GemmEpilogue ep;
        ep.bias = net->p("node_embedding.bias");
        MI_TRY(gemm_nt(atom_types, MI_NUM_TYPES, net->p("node_embedding.weight"), MI_NUM_TYPES, b->x1, H, N, H, MI_NUM_TYPES, ep, s, &b->sk));
        GemmEpilogue et;
        et.bias = net->p("atom_latent_emb.bias");
        MI_TRY(gemm_nt(t_emb, TD, net->p("atom_latent_emb.weight") + H, H + TD, b->tproj, H, B, H, TD, et, s, &b->sk));
        GemmEpilogue eh;
        eh.row_bias = b->tproj;
        eh.row_group = b->node2graph;
        eh.ld_row_bias = H;
        MI_TRY(gemm_nt(b->x1, H, net->p("atom_latent_emb.weight"), H + TD, b->h, H, N, H, H, eh, s, &b->sk));
    }
    // what the previous evaluation of this handle left behind is valid for THIS call only if it says so (reuse_embedding); every forward re-earns the flags
    // (a forward that takes another path -- a knob changed, a training forward -- leaves nothing a later call could mistake for its own)
    const bool same_net = b->reuse_net == (const void*)net && b->reuse_epoch == net->param_epoch;   // (same network AND same parameter version)
    const bool had_gram = b->gram_valid && same_net, had_pq0 = b->pq0_valid && same_net;
    b->gram_valid = b->pq0_valid = false;
    b->reuse_net = net;
    b->reuse_epoch = net->param_epoch;
    bool absmax_cleared = false;   // the pair-mode Fourier launch cleared b->absmax on the way (one launch fewer per evaluation)
    bool gram_kept = false;        // ... and left the lattice term's slots alone: G is the previous evaluation's
    bool pq0_kept = false;         // ... and layer 0's slot: its [P_i | P_j | X_part] (b->PQ0) is the previous evaluation's as well
    const bool fused_early = (!train || g_node_train) && !(g_node_hi && !train) && MI_PLANES_FP16 && g_gemm_mode == MI_GEMM_SPLIT && net->edge_mode != 0 && b->E > 0 &&
                             node_chain_supported(net);   // (= `fused` below: the node chain runs as node_chain.hip's launches)
    if (fused_early && !train && !b->PQ0) MI_TRY(dev_alloc(b, &b->PQ0, (size_t)3 * N * H));   // (first inference forward on the fused path: once per handle)
    // ---- Fourier operand: identical in every layer (cspnet.py:65-66), built once per evaluation ----
    if (b->E > 0 && net->edge_mode == 0) {
        const int64_t nf4 = (int64_t)cdiv(b->E, 32) * (net->KP / 4) * 64;
        hipLaunchKernelGGL(fourier_pack_kernel, dim3((unsigned)cdiv(nf4, 256)), dim3(256), 0, s, frac, b->fd, b->src, b->dst, b->FFp, b->E, net->F,
                           net->KP);
        MI_KERNEL_CHECK();
    } else if (b->E > 0 && g_gemm_mode == MI_GEMM_SPLIT && g_edge_pairs && !b->knn && H % 8 == 0) {
        if (b->Np > 0 && !((g_ablate_skip & 16) && b->ff_built_once)) {  // pair mode: one operand row per unordered pair  (bit 4 of the timing ablations: built once, then stale)
            b->ff_built_once = true;
            Planes ffp = make_planes(b->FFpl, 2 * net->Kh, PL_S_UNIT);
            const int64_t nthr = (b->Np + 127) / 128 * 128 * (int64_t)(net->Kh / 8);  // a lane per row and 8-column chunk
            // (the launch also clears the evaluation's 2 L absmax slots: see below)
            absmax_cleared = 2 * L <= 256 && B > 0 && L > 0;
            // (reuse_embedding = the sampler's predictor evaluation: it differs from the corrector evaluation in front of it in the coordinates only,
            //  diffusion.py:320-322 -- the lattice term G of every layer and its absmax slots are still valid: one more launch off the chain's serial path)
            gram_kept = absmax_cleared && reuse_embedding && had_gram && (g_eval_reuse & 1);
            pq0_kept = gram_kept && !train && fused_early && had_pq0 && b->PQ0 != nullptr && (g_eval_reuse & 2);
            unsigned* const zp = absmax_cleared ? b->absmax : nullptr;
            const int zn = absmax_cleared ? 2 * L : 0, ze = (gram_kept ? 1 : 0) | (pq0_kept ? 2 : 0);
            if (net->F % 8 == 0) hipLaunchKernelGGL(fourier_pair_planes_kernel<true>, dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, s, frac, b->pair_i, b->pair_j, ffp, b->Np, net->F, net->Kh, zp, zn, ze);
            else hipLaunchKernelGGL(fourier_pair_planes_kernel<false>, dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, s, frac, b->pair_i, b->pair_j, ffp, b->Np, net->F, net->Kh, zp, zn, ze);
            MI_KERNEL_CHECK();
        }
    } else if (b->E > 0 && g_gemm_mode == MI_GEMM_SPLIT) {
        Planes ffp = make_planes(b->FFpl, 6 * net->F, PL_S_UNIT);
        const int64_t rows_pad = (b->E + 127) / 128 * 128;
        if (net->F % 2 == 0) {
            const int64_t nthr = rows_pad * (int64_t)(ffp.KT * 16 - 3 * net->F / 2);
            const unsigned nblk = (unsigned)(b->e_dev ? std::min<int64_t>(cdiv(nthr, 256), 16384) : cdiv(nthr, 256));   // (a device-side length: a fixed, striding launch)
            hipLaunchKernelGGL(fourier_planes_kernel, dim3(nblk), dim3(256), 0, s, frac, b->fd, b->src, b->dst, ffp, b->E, net->F, b->e_dev);
        } else {
            const int64_t nthr = rows_pad * (int64_t)ffp.KT * 16;
            hipLaunchKernelGGL(fourier_planes_cols_kernel, dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, s, frac, b->fd, b->src, b->dst, ffp, b->E, net->F, b->e_dev);
        }
        MI_KERNEL_CHECK();
    } else if (b->E > 0) {
        hipLaunchKernelGGL(fourier_kernel, dim3((unsigned)cdiv(b->E * 3 * net->F, 256)), dim3(256), 0, s, frac, b->fd, b->src, b->dst, b->FF, b->E, net->F);
        MI_KERNEL_CHECK();
    }
    // gram-matrix term of every layer's first edge linear (cspnet.py:68-72), one launch
    if (B > 0 && L > 0) {
        const float* w0 = net->p("csp_layer_0.edge_mlp.0.weight");
        const float* b0 = net->p("csp_layer_0.edge_mlp.0.bias");
        const int64_t lstride = L > 1 ? net->p("csp_layer_1.edge_mlp.0.weight") - w0 : 0;
        MI_CHECK(L == 1 || net->p("csp_layer_1.edge_mlp.0.bias") - b0 == lstride, MI_ESTATE, "layer parameters are not uniformly strided");
        if (!absmax_cleared) MI_HIP(hipMemsetAsync(b->absmax, 0, 2 * L * sizeof(unsigned), s));
        if (g_node_cols == 2 && b->nc_flags) MI_HIP(hipMemsetAsync(b->nc_flags, 0, (size_t)(L + 1) * 2 * cdiv(N, 32) * sizeof(unsigned), s));
        if (!gram_kept) hipLaunchKernelGGL(gram_term_all_kernel, dim3(cdiv(B, GRAM_GB), L), dim3(256), 0, s, lattices, w0, lstride, net->edge_in, b0, b->G, H, B, b->absmax + 1);
        b->gram_valid = !train;   // (valid for a following reuse_embedding evaluation of this batch handle; a training forward's G is consumed by its own tape)
        MI_KERNEL_CHECK();
    }
    // Node-level kernels on a helper stream of the highest priority (experiment, `g_node_hi`): with four chains in flight a chain's short
    // node-level launches queue behind the other chains' edge GEMM workgroups (18 us of work take ~55 us), and a denoising step is the
    // SUM of one chain's kernel durations -- the chains run side by side, each at the pace of its own serial sequence.
    const bool use_hi = g_node_hi && !train;
    hipStream_t ns = s, cur = s;
    int evi = 0;
    if (use_hi) {
        if (!b->hi_stream) {
            int lo = 0, hi = 0;
            MI_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            MI_HIP(hipStreamCreateWithPriority(&b->hi_stream, hipStreamNonBlocking, hi));
            for (hipEvent_t& e : b->hi_ev) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        ns = b->hi_stream;
    }
    auto to = [&](hipStream_t x) -> int {   // the work that follows goes to stream x: join it behind the stream used so far
        if (cur == x) return MI_OK;
        MI_HIP(hipEventRecord(b->hi_ev[evi], cur));
        MI_HIP(hipStreamWaitEvent(x, b->hi_ev[evi], 0));
        evi ^= 1;
        cur = x;
        return MI_OK;
    };
    // ---- message-passing layers (cspnet.py:84-91) ----
    const bool node_planes = g_gemm_mode == MI_GEMM_SPLIT && net->edge_mode != 0 && H % 32 == 0 && N >= g_node_planes_min_rows;
    // inference: everything between two edge stages -- aggregation, node MLP, residual, LayerNorm, the projections LayerNorm(h) feeds --
    // is ONE launch per layer boundary (node_chain.hip), for any batch size
    // training: the same launch, which then also writes what the backward pass reads (cat = [LayerNorm(h) | agg], the two pre-activations,
    // the LayerNorm statistics) -- five launches per layer fewer than layernorm + PQ product + finalize_agg + the two node-MLP products
    const bool fused = fused_early;   // (use_hi = g_node_hi && !train is part of it)
    for (int l = 0; l < L; ++l) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        const float* h_in = b->h + l * NH;
        float* h_out = b->h + (l + 1) * NH;
        // (training: kept for the backward pass -- in this micro-step's slot of the weight-gradient window when one is set, see Tape)
        float* cat = !train ? b->cat : tp.wslots > 0 ? tp.w_cat + ((size_t)l * tp.wslots * N + (size_t)tp.wcur * N) * 2 * H : tp.cat + (size_t)l * N * 2 * H;
        // node-level products on the plane-set kernel too (weights pre-split once per parameter update, the activations' planes
        // written by their producers; the fp32-operand kernel re-splits every operand tile in every workgroup).  Grouped by
        // operand: everything LayerNorm(h) feeds -- P_i, P_j and its half of the node MLP's first product -- is ONE product
        // before the edge stage, so only agg x W[:, H:] (K = H instead of 2H) is left on the path after it.
        const Planes lnp = node_planes ? make_planes(b->lnpl, H, PL_S_LN) : Planes();
        const Planes aggp = node_planes ? make_planes(b->aggpl, H, PL_S_ACT, b->dsc + 2) : Planes();
        const int ldpq = (node_planes || fused) ? 3 * H : 2 * H;
        MI_TRY(to(ns));
        // layer l's [P_i | P_j | X_part]: layer 0's lives in a buffer of its own on the inference node-chain path (mi_batch::PQ0)
        float* const PQl = (fused && !train && l == 0 && b->PQ0) ? b->PQ0 : b->PQ;
        if (fused) {
            if (!(l == 0 && pq0_kept)) MI_TRY(node_chain(net, b, l, s, train));
            if (l == 0) b->pq0_valid = !train && b->PQ0 != nullptr;
        } else {
        if (net->cfg.ln) {
            hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(N, 4)), dim3(256), 0, ns, h_in, net->p(p + "layer_norm.weight"),
                               net->p(p + "layer_norm.bias"), cat, 2 * H, train ? tp.lnstat + (size_t)l * N * 2 : (float*)nullptr, N, H, lnp);
        } else {
            hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(NH, 256)), dim3(256), 0, ns, h_in, cat, 2 * H, N, H, lnp);
        }
        MI_KERNEL_CHECK();
        if (node_planes) {
            PlanesEpilogue pq;
            pq.C = PQl;
            pq.ldc = ldpq;
            pq.absmax = MI_PLANES_FP16 ? b->absmax + 2 * l : nullptr;
            MI_TRY(gemm_planes(lnp, make_planes(net->Wlnpl + (size_t)l * planes_elems(3 * H, H), H), N, 3 * H, H, pq, ns));
        } else {
            MI_TRY(gemm_nt(cat, 2 * H, net->Whh + l * net->whh_stride(), H, PQl, 2 * H, N, 2 * H, H, GemmEpilogue(), ns, &b->sk));
            if (MI_PLANES_FP16 && net->edge_mode != 0 && g_gemm_mode == MI_GEMM_SPLIT && b->E > 0) {
                hipLaunchKernelGGL(absmax_kernel, dim3(std::min<int64_t>(256, cdiv((int64_t)N * 2 * H, 256))), dim3(256), 0, ns, PQl, (int64_t)N * 2 * H,
                                   b->absmax + 2 * l);
                MI_KERNEL_CHECK();
            }
        }
        }
        MI_TRY(to(s));
        // pair mode folds the activation scales and the self edges into the launch of its Fourier-block GEMM
        const bool pair_path = net->edge_mode != 0 && b->E > 0 && g_gemm_mode == MI_GEMM_SPLIT && g_edge_pairs && !b->knn && H % 8 == 0;
        const bool fold = pair_path && g_fold_pair_extras && b->Np > 0;
        if (train) tp.dsc_layers_valid = fold && MI_PLANES_FP16;
        if (MI_PLANES_FP16 && net->edge_mode != 0 && g_gemm_mode == MI_GEMM_SPLIT && !fold) {  // scales of this layer's M1 / agg / X plane sets
            hipLaunchKernelGGL(act_scales_kernel, dim3(1), dim3(1), 0, s, b->absmax + 2 * l, b->absmax + 2 * l + 1, net->wbounds + (size_t)l * 8, b->dsc);
            MI_KERNEL_CHECK();
        }
        if (net->edge_mode == 0) {  // fused register-chained f32-MFMA kernel
            MI_TRY(launch_edge(net, b, l, frac, train ? tp.Z1 + (size_t)l * b->E * H : nullptr, train ? tp.Z2 + (size_t)l * b->E * H : nullptr, s));
            MI_TRY(to(ns));
            if (!fused) hipLaunchKernelGGL(finalize_agg_kernel, dim3(cdiv(NH, 256)), dim3(256), 0, ns, b->part, b->rowptr, cat, N, H, aggp);
            MI_KERNEL_CHECK();
        } else if (b->E > 0) {      // two tiled GEMMs over the edge list with gather / SiLU epilogues
            const int E = (int)b->E, F6 = 6 * net->F;
            ProfSlot ps;
            MI_TRY(prof_begin(net, s, &ps));
            GemmEpilogue g1e;       // Z1 = FF Wff^T + P_i[src] + P_j[dst] + G[graph];  M1 = SiLU(Z1)
            g1e.row_bias = PQl;
            g1e.row_group = b->src;
            g1e.ld_row_bias = ldpq;
            g1e.row_bias2 = PQl + H;
            g1e.row_group2 = b->dst;
            g1e.ld_row_bias2 = ldpq;
            g1e.row_bias3 = b->G + (size_t)l * B * H;
            g1e.row_group3 = b->edge_graph;
            g1e.ld_row_bias3 = H;
            g1e.act = ACT_SILU;
            if (train) {
                g1e.pre_act = tp.Z1 + (size_t)l * b->E * H;
                g1e.ld_pre = H;
            }
            GemmEpilogue g2e;       // M2 = SiLU(M1 W2^T + b2)
            g2e.bias = net->p(p + "edge_mlp.2.bias");
            g2e.act = ACT_SILU;
            if (train) {
                g2e.pre_act = tp.Z2 + (size_t)l * b->E * H;
                g2e.ld_pre = H;
            }
            if (g_gemm_mode == MI_GEMM_SPLIT) {  // operands pre-split into bf16 planes: pure bf16 GEMMs
                Planes ffp = make_planes(b->FFpl, F6, PL_S_UNIT);
                Planes wffp = make_planes(net->Wffpl + (size_t)l * planes_elems(H, F6), F6);
                b->m1_cur = (train && tp.M1pl_l) ? tp.M1pl_l + (size_t)l * tp.m1pl_stride : b->M1pl;
                Planes m1p = make_planes(b->m1_cur, H, PL_S_ACT, b->dsc);
                Planes w2p = make_planes(net->W2pl + (size_t)l * planes_elems(H, H), H);
                PlanesEpilogue pe1;
                pe1.ep = g1e;
                pe1.Cp = m1p;
                pe1.m_dev = b->e_dev;   // (knn lists built without a host round trip: E is the capacity)
                pe1.m_hint = (int)std::min<int64_t>(b->e_hint, INT32_MAX);
                if (g_edge_pairs && !b->knn && H % 8 == 0) {  // (the pair epilogue moves 8 columns per lane)
                    // symmetric edge list: sin(2 pi k (1 - d)) = -sin(2 pi k d), cos unchanged, so one operand row per unordered
                    // pair yields both directed edges (half the MFMA work of this GEMM); self edges (d = 0) are a constant
                    const int Kp = 2 * net->Kh;
                    pe1.pair_i = b->pair_i;
                    pe1.pair_j = b->pair_j;
                    pe1.pair_e1 = b->pair_e1;
                    pe1.pair_e2 = b->pair_e2;
                    pe1.pair_graph = b->pair_graph;
                    pe1.pair_wide = g_pair_wide_force || !pairs_fit_32bit(N, ldpq, B, H, b->E, H);   // (sizes allowing, the epilogue addresses in 32 bits)
                    if (fold) {
                        if (MI_PLANES_FP16) {
                            pe1.sc_pq = b->absmax + 2 * l;
                            pe1.sc_gmax = b->absmax + 2 * l + 1;
                            pe1.sc_wb = net->wbounds + (size_t)l * 8;
                            pe1.sc_dsc = b->dsc;
                            if (train) pe1.sc_dsc2 = tp.dsc_layers + (size_t)l * 8;
                        }
                        pe1.diag_C0 = net->C0 + (size_t)l * H;
                        pe1.diag_node2graph = b->node2graph;
                        pe1.diag_e = b->e_diag;
                        pe1.diag_nodes = N;
                    }
                    const bool efused = !train && fused && fold && MI_PLANES_FP16 && edge_fused_supported(net, b);
                    if (efused) {   // both edge products and the edge -> node sums in one launch, M1 stays in LDS (edge_fused.hip)
                        MI_TRY(edge_fused(net, b, l, s));
                        b->seg_shift = -1;
                        MI_TRY(prof_end(net, s, ps));
                        continue;   // (fused: the rest of the layer runs in node_chain(l + 1))
                    }
                    if (b->Np > 0 && fold && edge_gemm1_supported(net, b->Np))   // 128 x 128 tiles, weights straight from L2 in fragment order (edge_stage.hip)
                        MI_TRY(edge_gemm1(net, make_planes(b->FFpl, Kp, PL_S_UNIT), l, (int)b->Np, pe1, s));
                    else if (b->Np > 0)
                        MI_TRY(gemm_planes(make_planes(b->FFpl, Kp, PL_S_UNIT), make_planes(net->Wffpl_pair + (size_t)l * planes_elems(H, Kp), Kp), (int)b->Np, H, Kp,
                                           pe1, s));
                    if (!fold) {
                        hipLaunchKernelGGL(edge_diag_kernel, dim3(cdiv((int64_t)N * (H / 2), 256)), dim3(256), 0, s, PQl, b->G + (size_t)l * B * H,
                                           net->C0 + (size_t)l * H, b->node2graph, b->e_diag, g1e.pre_act, m1p, N, H, ldpq);
                        MI_KERNEL_CHECK();
                    }
                } else {
                    MI_TRY(gemm_planes(ffp, wffp, E, H, F6, pe1, s));
                }
                // 128-row register tiles, segmented sum on the matrix pipe (edge_stage.hip): inference forwards next to the node-chain launch, and
                // the training forward with the pre-activation kept for the backward pass (written row-major through per-wave LDS patches: as
                // 4-byte stores from the result layout the instantiation spilled 228 registers and LOST 6-8 %; this form gains 1.9 % on the
                // fine-tune line).  The partial sums then round M2 to the plane format's 22 bits, as in inference.
                // (fully connected lists only: the kernel sizes its per-tile node tables for at most 128 consecutive node ids per 128-row tile, which
                //  holds when every node has an edge -- a knn list may hold zero-degree atoms inside a tile's span; those batches keep the plane GEMM)
                const bool eg2_ok = edge_gemm2_supported(net) && !b->knn;
                const bool eg2_train = train && g_edge2_train && MI_PLANES_FP16 && g2e.pre_act && g2e.ld_pre == H && !use_hi && eg2_ok;
                if ((fused && eg2_ok) || eg2_train) {
                    MI_TRY(edge_gemm2(net, b, l, s, eg2_train ? g2e.pre_act : nullptr));
                    b->seg_shift = 7;
                } else {
                PlanesEpilogue pe2;   // M2 never reaches HBM: the edge -> node sum happens in the epilogue
                pe2.ep = g2e;
                pe2.seg_part = b->part;
                pe2.seg_src = b->src;
                pe2.seg_rowptr = b->rowptr;
                pe2.seg_nodes = N;
                pe2.m_dev = b->e_dev;
                pe2.m_hint = (int)std::min<int64_t>(b->e_hint, INT32_MAX);
                MI_TRY(gemm_planes(m1p, w2p, E, H, H, pe2, s));
                b->seg_shift = 5;
                }
                MI_TRY(prof_end(net, s, ps));
                MI_TRY(to(ns));
                if (!fused) hipLaunchKernelGGL(finalize_agg_kernel, dim3(cdiv(NH, 256)), dim3(256), 0, ns, b->part, b->rowptr, cat, N, H, aggp, b->seg_shift);
                MI_KERNEL_CHECK();
            } else {
                MI_TRY(gemm_nt(b->FF, F6, net->Wff + (size_t)l * H * F6, F6, b->M1, H, E, H, F6, g1e, s));
                MI_TRY(gemm_nt(b->M1, H, net->p(p + "edge_mlp.2.weight"), H, b->M2, H, E, H, H, g2e, s));
                MI_TRY(prof_end(net, s, ps));
                MI_TRY(to(ns));
                hipLaunchKernelGGL(segment_mean_kernel, dim3(cdiv(NH, 256)), dim3(256), 0, ns, b->M2, b->rowptr, cat, N, H);
                MI_KERNEL_CHECK();
            }
        } else {
            MI_TRY(to(ns));
            hipLaunchKernelGGL(finalize_agg_kernel, dim3(cdiv(NH, 256)), dim3(256), 0, ns, b->part, b->rowptr, cat, N, H, aggp);
            MI_KERNEL_CHECK();
        }
        if (fused) continue;  // the rest of this layer runs in front of the next layer's edge stage (node_chain(l + 1))
        GemmEpilogue e1;
        e1.bias = net->p(p + "node_mlp.0.bias");
        e1.act = ACT_SILU;
        if (train) {
            e1.pre_act = tp.Xpre + (size_t)l * NH;
            e1.ld_pre = H;
        }
        if (node_planes) {
            PlanesEpilogue p1;
            p1.ep = e1;
            p1.ep.pre_add = PQl + 2 * H;  // LayerNorm(h) x W[:, :H], computed with P_i / P_j
            p1.ep.ld_pre_add = ldpq;
            p1.Cp = make_planes(b->Xpl, H, PL_S_ACT, b->dsc + 4);
            MI_TRY(gemm_planes(aggp, make_planes(net->Waggpl + (size_t)l * planes_elems(H, H), H), N, H, H, p1, ns));
        } else {
            MI_TRY(gemm_nt(cat, 2 * H, net->p(p + "node_mlp.0.weight"), 2 * H, b->X, H, N, H, 2 * H, e1, ns, &b->sk));
        }
        GemmEpilogue e2;
        e2.bias = net->p(p + "node_mlp.2.bias");
        e2.act = ACT_SILU;
        e2.residual = h_in;
        e2.ld_res = H;
        if (train) {
            e2.pre_act = tp.Ypre + (size_t)l * NH;
            e2.ld_pre = H;
        }
        if (node_planes) {
            PlanesEpilogue p2;
            p2.ep = e2;
            p2.C = h_out;
            p2.ldc = H;
            MI_TRY(gemm_planes(make_planes(b->Xpl, H, PL_S_ACT, b->dsc + 4), make_planes(net->Wn2pl + (size_t)l * planes_elems(H, H), H), N, H, H, p2, ns));
        } else {
            MI_TRY(gemm_nt(b->X, H, net->p(p + "node_mlp.2.weight"), H, h_out, H, N, H, H, e2, ns, &b->sk));
        }
    }
    MI_TRY(to(s));
    // ---- heads (cspnet.py:276-291) ----
    const float* h_last = b->h + (size_t)L * NH;
    if (fused) {
        MI_TRY(node_chain(net, b, L, s, train));   // the last layer's node MLP + residual and the final LayerNorm -> b->hf
    } else if (net->cfg.ln) {
        hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, h_last, net->p("final_layer_norm.weight"),
                           net->p("final_layer_norm.bias"), b->hf, H, train ? tp.lnstat + (size_t)L * N * 2 : (float*)nullptr, N, H);
    } else {
        hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(NH, 256)), dim3(256), 0, s, h_last, b->hf, H, N, H);
    }
    MI_KERNEL_CHECK();
    MI_CHECK(!(coords_only && train), MI_EINVAL, "a training forward evaluates every head");
    if (!train && net->WheadT && g_fused_heads && H % 64 == 0) {
        const int ncols = (coords_only && (g_eval_reuse & 4)) ? 3 : 3 + MI_NUM_TYPES;
        auto lds = [&](int rows) { return (size_t)(rows * H + 4 * rows * 128) * sizeof(float); };
        if (N >= g_heads_rows16_min_nodes) {   // (16 rows per workgroup need 64.5 KB of LDS at H = 512: above the default ceiling)
            static std::once_flag once;
            static hipError_t attr_err = hipSuccess;
            std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)heads_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); });
            MI_HIP(attr_err);
            hipLaunchKernelGGL(heads_kernel<16>, dim3(cdiv(N, 16)), dim3(512), lds(16), s, b->hf, net->WheadT, net->p("type_out.bias"), coord_out, type_out, N, H, ncols);
        } else {
            hipLaunchKernelGGL(heads_kernel<4>, dim3(cdiv(N, 4)), dim3(512), lds(4), s, b->hf, net->WheadT, net->p("type_out.bias"), coord_out, type_out, N, H, ncols);
        }
        MI_KERNEL_CHECK();
    } else {
        MI_TRY(gemm_nt(b->hf, H, net->p("coord_out.weight"), H, coord_out, 3, N, 3, H, GemmEpilogue(), s, &b->sk));
        if (!(coords_only && (g_eval_reuse & 4))) {
            GemmEpilogue et;
            et.bias = net->p("type_out.bias");
            MI_TRY(gemm_nt(b->hf, H, net->p("type_out.weight"), H, type_out, MI_NUM_TYPES, N, MI_NUM_TYPES, H, et, s, &b->sk));
        }
    }
    // (coords_only -- the sampler's corrector evaluation, which reads the coordinate score alone, diffusion.py:310-322: no type columns above, no lattice head)
    if (!(coords_only && (g_eval_reuse & 4)))
        hipLaunchKernelGGL(lattice_head_kernel, dim3(B), dim3(256), (2 * H + 12) * sizeof(float), s, b->hf, b->node_off,
                           net->p("lattice_out.weight"), lattices, lattice_out, train ? (hw ? tp.w_gf + hslot * B * H : tp.gf) : (float*)nullptr, H);
    MI_KERNEL_CHECK();
    tp.valid = train;
    return MI_OK;
}

}  // namespace mi

using namespace mi;

int64_t mi_net::off(const std::string& name) const {
    for (const auto& p : params)
        if (p.name == name) return p.off;
    fprintf(stderr, "matinvent_hip: unknown parameter %s\n", name.c_str());
    abort();
}

extern "C" {

const char* mi_last_error(void) { return mi::g_err; }
int mi_version(void) { return 1; }

int mi_trace_push(const char* name) {
    mi::trace_push(name ? name : "mi");
    return MI_OK;
}
int mi_trace_pop(void) {
    mi::trace_pop();
    return MI_OK;
}

int mi_net_create(const mi_net_config* cfg, mi_net** out) {
    MI_CHECK(cfg && out, MI_EINVAL, "null argument");
    const int H = cfg->hidden_dim;
    // any multiple of 64 up to 512 on the GEMM forms (the reference's code default is 128, cspnet.py:98-113; its real hparams.yaml is unreachable,
    // models/suite/diffcsp.py:52-55, so a checkpoint of another width must not be refused outright); the register-chained f32 edge kernel, the
    // node-chain launch and the 128 x 512 register-tile GEMM exist for their own widths only and the forward falls back to the plane GEMMs
    MI_CHECK(H >= 64 && H <= 512 && H % 64 == 0, MI_EINVAL, "hidden_dim must be a multiple of 64 in 64..512, got %d", H);
    MI_CHECK(cfg->num_layers >= 1 && cfg->num_freqs >= 1, MI_EINVAL, "num_layers/num_freqs must be >= 1");
    MI_CHECK(cfg->time_dim > 0 && cfg->time_dim % 4 == 0, MI_EINVAL, "time_dim must be a positive multiple of 4");
    mi_net* n = new mi_net();
    n->cfg = *cfg;
    n->H = H;
    n->L = cfg->num_layers;
    n->F = cfg->num_freqs;
    n->TD = cfg->time_dim;
    n->NT = H / 32;
    n->KP = 3 * ((n->F + 7) / 8 * 8);
    n->edge_in = 2 * H + 9 + 6 * n->F;
    int64_t off = 0;
    auto add = [&](const std::string& name, int rows, int cols) {
        n->params.push_back(ParamInfo{name, off, (int64_t)rows * cols, rows, cols});
        off += (int64_t)rows * cols;
    };
    add("node_embedding.weight", H, MI_NUM_TYPES);
    add("node_embedding.bias", 1, H);
    add("atom_latent_emb.weight", H, H + n->TD);
    add("atom_latent_emb.bias", 1, H);
    for (int l = 0; l < n->L; ++l) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        add(p + "edge_mlp.0.weight", H, n->edge_in);
        add(p + "edge_mlp.0.bias", 1, H);
        add(p + "edge_mlp.2.weight", H, H);
        add(p + "edge_mlp.2.bias", 1, H);
        add(p + "node_mlp.0.weight", H, 2 * H);
        add(p + "node_mlp.0.bias", 1, H);
        add(p + "node_mlp.2.weight", H, H);
        add(p + "node_mlp.2.bias", 1, H);
        if (cfg->ln) {
            add(p + "layer_norm.weight", 1, H);
            add(p + "layer_norm.bias", 1, H);
        }
    }
    add("coord_out.weight", 3, H);
    add("lattice_out.weight", 9, H);
    if (cfg->ln) {
        add("final_layer_norm.weight", 1, H);
        add("final_layer_norm.bias", 1, H);
    }
    add("type_out.weight", MI_NUM_TYPES, H);
    add("type_out.bias", 1, MI_NUM_TYPES);
    n->nparams = off;
    *out = n;
    return MI_OK;
}

void mi_net_destroy(mi_net* n) {
    if (!n) return;
    for (float* p : {n->freqs, n->Whh, n->Wff_p, n->W2_p, n->Wff, n->W2T, n->Wn2T, n->Wn1T, n->WhhT, n->WaT, n->WheadT})
        if (p) (void)hipFree(p);
    if (n->Wffpl) (void)hipFree(n->Wffpl);
    if (n->Wffpl_pair) (void)hipFree(n->Wffpl_pair);
    if (n->C0) (void)hipFree(n->C0);
    if (n->W2pl) (void)hipFree(n->W2pl);
    if (n->W2Tpl) (void)hipFree(n->W2Tpl);
    if (n->W2Tf) (void)hipFree(n->W2Tf);
    if (n->Wbw) (void)hipFree(n->Wbw);
    if (n->Wlnpl) (void)hipFree(n->Wlnpl);
    if (n->Waggpl) (void)hipFree(n->Waggpl);
    if (n->Wn2pl) (void)hipFree(n->Wn2pl);
    if (n->Wnc) (void)hipFree(n->Wnc);
    if (n->Wffc) (void)hipFree(n->Wffc);
    if (n->Wffc2) (void)hipFree(n->Wffc2);
    if (n->wbounds) (void)hipFree(n->wbounds);
    for (auto e : n->ev) (void)hipEventDestroy(e);
    delete n;
}

int64_t mi_net_num_params(const mi_net* n) { return n ? n->nparams : 0; }
int mi_net_num_tensors(const mi_net* n) { return n ? (int)n->params.size() : 0; }

int mi_net_param_info(const mi_net* n, int index, const char** name, int64_t* offset, int64_t* numel, int* rows, int* cols) {
    MI_CHECK(n && index >= 0 && index < (int)n->params.size(), MI_EINVAL, "parameter index %d out of range", index);
    const ParamInfo& p = n->params[index];
    if (name) *name = p.name.c_str();
    if (offset) *offset = p.off;
    if (numel) *numel = p.numel;
    if (rows) *rows = p.rows;
    if (cols) *cols = p.cols;
    return MI_OK;
}

int mi_net_set_params(mi_net* n, const float* theta, const float* freqs_host, void* stream) {
    MI_CHECK(n && theta, MI_EINVAL, "null argument");
    MI_CHECK((((uintptr_t)theta) & 15) == 0, MI_EINVAL, "theta must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int H = n->H;
    ++n->param_epoch;   // (what earlier evaluations left on their batch handles for reuse belongs to the previous parameter version)
    if (!n->Whh) {
        MI_HIP(hipMalloc((void**)&n->Whh, n->L * n->whh_stride() * sizeof(float)));
        MI_HIP(hipMalloc((void**)&n->Wff_p, n->L * n->wff_stride() * sizeof(float)));
        MI_HIP(hipMalloc((void**)&n->W2_p, n->L * n->w2_stride() * sizeof(float)));
        MI_HIP(hipMalloc((void**)&n->freqs, n->F * sizeof(float)));
        MI_HIP(hipMalloc((void**)&n->Wff, (size_t)n->L * H * 6 * n->F * sizeof(float)));
        MI_HIP(hipMalloc((void**)&n->Wffpl, (size_t)n->L * planes_elems(H, 6 * n->F) * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&n->W2pl, (size_t)n->L * planes_elems(H, H) * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&n->Wlnpl, (size_t)n->L * planes_elems(3 * H, H) * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&n->Waggpl, (size_t)n->L * planes_elems(H, H) * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&n->Wn2pl, (size_t)n->L * planes_elems(H, H) * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&n->wbounds, (size_t)n->L * 8 * sizeof(float)));
        if (node_chain_pack_elems(H) && cfg_ln_and_wide(n)) MI_HIP(hipMalloc((void**)&n->Wnc, (size_t)n->L * node_chain_pack_elems(H) * sizeof(u16)));
        if (MI_PLANES_FP16 && H % 128 == 0) MI_HIP(hipMalloc((void**)&n->Wffc, (size_t)n->L * H * 2 * ((3 * n->F + 31) / 32 * 32) * 2 * sizeof(u16)));
        if (MI_PLANES_FP16 && MI_HAVE_ABLATION_KERNELS && H % 256 == 0) MI_HIP(hipMalloc((void**)&n->Wffc2, (size_t)n->L * H * ((3 * n->F + 31) / 32 * 32) * 2 * sizeof(u16)));   // -2 x the sine block (edge_gemm1e_kernel)
        n->Kh = (3 * n->F + 31) / 32 * 32;
        MI_HIP(hipMalloc((void**)&n->Wffpl_pair, (size_t)n->L * planes_elems(H, 2 * n->Kh) * sizeof(u16)));
        MI_HIP(hipMalloc((void**)&n->C0, (size_t)n->L * H * sizeof(float)));
    }
    if (freqs_host) {
        MI_HIP(hipMemcpyAsync(n->freqs, freqs_host, n->F * sizeof(float), hipMemcpyHostToDevice, s));
        MI_HIP(hipStreamSynchronize(s));  // the host buffer may be transient
        n->have_freqs = true;
    }
    MI_CHECK(n->have_freqs, MI_ESTATE, "fourier_freqs_host must be given on the first mi_net_set_params");
    n->theta = theta;
    for (int l = 0; l < n->L; ++l) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        const float* W1 = n->p(p + "edge_mlp.0.weight");
        const float* W2 = n->p(p + "edge_mlp.2.weight");
        hipLaunchKernelGGL(pack_whh_kernel, dim3(cdiv(2 * H * H, 256)), dim3(256), 0, s, W1, n->edge_in, H,
                           n->Whh + l * n->whh_stride());
        hipLaunchKernelGGL(pack_wff_kernel, dim3(cdiv(n->wff_stride(), 256)), dim3(256), 0, s, W1, n->edge_in, H, n->F, n->KP,
                           n->Wff_p + l * n->wff_stride());
        hipLaunchKernelGGL(pack_w2_kernel, dim3(cdiv(H * H, 256)), dim3(256), 0, s, W2, H, n->W2_p + l * n->w2_stride());
        hipLaunchKernelGGL(pack_wff_plain_kernel, dim3(cdiv(H * 6 * n->F, 256)), dim3(256), 0, s, W1, n->edge_in, H, 6 * n->F,
                           n->Wff + (size_t)l * H * 6 * n->F);
        {
            const int F6 = 6 * n->F, Hp = (H + 127) / 128 * 128;
            Planes wffp = make_planes(n->Wffpl + (size_t)l * planes_elems(H, F6), F6);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)Hp * wffp.KT * 16, 256)), dim3(256), 0, s, W1 + 2 * H + 9, n->edge_in, H, F6, wffp);
            Planes w2p = make_planes(n->W2pl + (size_t)l * planes_elems(H, H), H);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)Hp * w2p.KT * 16, 256)), dim3(256), 0, s, W2, H, H, H, w2p);
            Planes wpp = make_planes(n->Wffpl_pair + (size_t)l * planes_elems(H, 2 * n->Kh), 2 * n->Kh);
            hipLaunchKernelGGL(pack_wff_pair_planes_kernel, dim3(cdiv((int64_t)Hp * n->Kh, 256)), dim3(256), 0, s, W1, n->edge_in, H, n->F, n->Kh, wpp);
            hipLaunchKernelGGL(wff_cos_rowsum_kernel, dim3(cdiv(H, 4)), dim3(256), 0, s, W1, n->edge_in, H, n->F, n->C0 + (size_t)l * H);
            // node-level weights, grouped by operand: [P_i; P_j; node_mlp.0[:, :H]] multiply LayerNorm(h), node_mlp.0[:, H:] the
            // aggregated messages (that product then sits alone on the path after the edge stage)
            const float* Wn0 = n->p(p + "node_mlp.0.weight");
            const int H3p = (3 * H + 127) / 128 * 128;
            Planes wlnp = make_planes(n->Wlnpl + (size_t)l * planes_elems(3 * H, H), H);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)H3p * wlnp.KT * 16, 256)), dim3(256), 0, s, n->Whh + l * n->whh_stride(), H, 2 * H, H, wlnp, 0);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)H3p * wlnp.KT * 16, 256)), dim3(256), 0, s, Wn0, 2 * H, H, H, wlnp, 2 * H);
            Planes waggp = make_planes(n->Waggpl + (size_t)l * planes_elems(H, H), H);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)Hp * waggp.KT * 16, 256)), dim3(256), 0, s, Wn0 + H, 2 * H, H, H, waggp, 0);
            Planes wn2p = make_planes(n->Wn2pl + (size_t)l * planes_elems(H, H), H);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)Hp * wn2p.KT * 16, 256)), dim3(256), 0, s, n->p(p + "node_mlp.2.weight"), H, H, H, wn2p);
            if (n->Wnc) MI_TRY(node_chain_pack(n, l, W1, Wn0, n->p(p + "node_mlp.2.weight"), W2, s));
            if (n->Wffc) MI_TRY(edge_gemm1_pack(n, l, W1, s));  // the same weights in fragment order (node_chain.hip)
        }
    }
    if (n->L > 0) {  // weight bounds behind the activation scales of the fp16 plane format (see act_scales_kernel)
        const float* w0 = n->p("csp_layer_0.edge_mlp.0.weight");
        const int64_t lstride = n->L > 1 ? n->p("csp_layer_1.edge_mlp.0.weight") - w0 : 0;
        MI_HIP(hipMemsetAsync(n->wbounds, 0, (size_t)n->L * 8 * sizeof(float), s));
        hipLaunchKernelGGL(weight_bounds_kernel, dim3(n->L, cdiv(H, 64)), dim3(256), 0, s, w0, n->p("csp_layer_0.edge_mlp.2.weight"),
                           n->p("csp_layer_0.edge_mlp.2.bias"), n->p("csp_layer_0.node_mlp.0.weight"), n->p("csp_layer_0.node_mlp.0.bias"), lstride,
                           n->edge_in, H, 6 * n->F, reinterpret_cast<unsigned*>(n->wbounds));
    }
    if (!n->WheadT) MI_HIP(hipMalloc((void**)&n->WheadT, (size_t)H * HEADS_LD * sizeof(float)));
    hipLaunchKernelGGL(pack_heads_kernel, dim3(cdiv(H * HEADS_LD, 256)), dim3(256), 0, s, n->p("coord_out.weight"), n->p("type_out.weight"), n->WheadT, H);
    MI_KERNEL_CHECK();
    MI_TRY(net_pack_transposes(n, s));
    return MI_OK;
}

static int batch_create_impl(const mi_net* net, const int* num_atoms_host, int B, int64_t node_offset, int64_t graph_offset, bool knn,
                             int max_neighbors, int cap_per_node, mi_batch** out) {
    MI_CHECK(net && out && (num_atoms_host || B == 0) && B >= 0, MI_EINVAL, "bad argument");
    mi_batch* b = new mi_batch();
    b->B = B;
    b->H = net->H;
    b->L = net->L;
    b->node_offset = node_offset;
    b->graph_offset = graph_offset;
    b->num_atoms_h.assign(num_atoms_host, num_atoms_host + B);
    b->node_off_h.assign(B + 1, 0);
    int64_t E = 0;
    for (int g = 0; g < B; ++g) {
        if (num_atoms_host[g] < 0) {
            delete b;
            set_error("num_atoms[%d] = %d is negative", g, num_atoms_host[g]);
            return MI_EINVAL;
        }
        b->node_off_h[g + 1] = b->node_off_h[g] + num_atoms_host[g];
        E += (int64_t)num_atoms_host[g] * num_atoms_host[g];
    }
    const int N = b->node_off_h[B];
    b->N = N;
    if (knn) {  // edge list built on the device by every forward; buffers sized for the capacity
        int rck = knn_alloc(b, max_neighbors, cap_per_node);
        if (rck != MI_OK) {
            mi_batch_destroy(b);
            return rck;
        }
        E = b->E_cap;
    }
    b->E = knn ? 0 : E;
    b->E_cap = E;
    if (E >= (int64_t)1 << 31) {
        delete b;
        set_error("edge count %lld exceeds int32", (long long)E);
        return MI_EINVAL;
    }
    // fully connected edges, row-major incl. self loops (cspnet.py:239-241)
    const size_t Efc = knn ? 0 : (size_t)E;
    std::vector<int> n2g(N), src(Efc), dst(Efc), rowptr(N + 1, 0), egraph(Efc), pr_i, pr_j, pr_e1, pr_e2, pr_g, ediag(knn ? 0 : N), pr_off(B + 1, 0);
    size_t e = 0;
    int nslots = knn ? b->deg_cap / 32 + 2 : 1;
    for (int g = 0; g < B; ++g) {
        int n = num_atoms_host[g], o = b->node_off_h[g];
        for (int i = 0; i < n; ++i) {
            n2g[o + i] = g;
            rowptr[o + i] = (int)e;
            for (int j = 0; j < n && !knn; ++j) {
                src[e] = o + i;
                dst[e] = o + j;
                egraph[e] = g;
                ++e;
            }
            if (!knn) nslots = std::max(nslots, (int)((e - 1) >> 5) - (rowptr[o + i] >> 5) + 1);
        }
        if (!knn) {  // unordered pairs i < j and self edges of this crystal; edge (a -> b) sits at rowptr[a] + b_local
            pr_off[g] = (int)pr_i.size();
            b->nmax_fc = std::max(b->nmax_fc, n);
            for (int i = 0; i < n; ++i) {
                ediag[o + i] = rowptr[o + i] + i;
                for (int j = i + 1; j < n; ++j) {
                    pr_i.push_back(o + i);
                    pr_j.push_back(o + j);
                    pr_e1.push_back(rowptr[o + i] + j);
                    pr_e2.push_back(rowptr[o + j] + i);
                    pr_g.push_back(g);
                }
            }
        }
    }
    b->Np = (int64_t)pr_i.size();
    pr_off[B] = (int)pr_i.size();
    rowptr[N] = (int)e;
    // edge_fused.hip's tables: 64-pair tiles, slots of a node's partial sums (see mi_batch)
    std::vector<int> ef_tile0(std::max(N, 1), 1 << 30);   // (nodes of crystals without pairs: no pair tile)
    std::vector<unsigned> ef_mask(std::max(N, 1), 0u);
    if (!knn && b->Np > 0) {
        int span = 0;
        for (int g = 0; g < B; ++g) {
            const int n = num_atoms_host[g], o = b->node_off_h[g];
            if (pr_off[g + 1] > pr_off[g]) {
                const int t0 = pr_off[g] >> 6, t1 = (pr_off[g + 1] - 1) >> 6;
                span = std::max(span, t1 - t0 + 1);
                for (int i = 0; i < n; ++i) ef_tile0[o + i] = t0;
            }
        }
        b->ef_nslots = span + 1;
        b->ef_ok = b->ef_nslots <= 32;
        if (b->ef_ok) {
            for (size_t q = 0; q < pr_i.size(); ++q) {
                const int t = (int)(q >> 6);
                ef_mask[pr_i[q]] |= 1u << (t - ef_tile0[pr_i[q]]);
                ef_mask[pr_j[q]] |= 1u << (t - ef_tile0[pr_j[q]]);
            }
            for (int v = 0; v < N; ++v) ef_mask[v] |= 1u << (b->ef_nslots - 1);
            for (size_t q0 = 0; q0 < pr_i.size() && b->ef_ok; q0 += 64) {   // every tile's node range must fit the 128 local nodes of its S matrix
                int lo = pr_i[q0], hi = pr_j[q0];
                for (size_t q = q0; q < std::min(pr_i.size(), q0 + 64); ++q) {
                    lo = std::min(lo, pr_i[q]);
                    hi = std::max(hi, pr_j[q]);
                }
                if (hi - lo + 1 > 128) b->ef_ok = false;
            }
        }
        if (b->ef_ok && MI_HAVE_ABLATION_KERNELS) nslots = std::max(nslots, b->ef_nslots);   // (edge_fused.hip exists in ablation builds only)
    }
    b->nslots = nslots;
    const int H = net->H, L = net->L;
    const size_t NH = (size_t)N * H;
    int rc = MI_OK;
#define A_(p, n)                                   \
    if (rc == MI_OK) rc = dev_alloc(b, &b->p, (n))
    A_(num_atoms, B);
    A_(node_off, B + 1);
    A_(node2graph, N);
    A_(src, (size_t)E);
    A_(dst, (size_t)E);
    A_(edge_graph, (size_t)E);
    A_(rowptr, N + 1);
    A_(h, (L + 1) * NH);
    A_(cat, 2 * NH);
    A_(PQ, 3 * NH);  // [N][2H] P_i | P_j, or [N][3H] with the LayerNorm(h) part of the node MLP's first product appended
    // (PQ0 -- layer 0's projections of the INFERENCE node chain, 3 N H floats -- is allocated by the first forward that takes that path: training-only, knn and
    //  non-fp16 handles never pay for it)
    A_(G, (size_t)L * B * H);
    A_(part, nslots * NH);
    A_(FFp, (size_t)cdiv(E, 32) * (net->KP / 4) * 256);
    A_(FF, (size_t)E * 6 * net->F);
    A_(M1, (size_t)E * H);
    A_(M2, (size_t)E * H);
    A_(FFpl, std::max(planes_elems(E, 6 * net->F), planes_elems(b->Np, 2 * ((3 * net->F + 31) / 32 * 32))));
    A_(pair_i, (size_t)b->Np);
    A_(pair_j, (size_t)b->Np);
    A_(pair_e1, (size_t)b->Np);
    A_(pair_e2, (size_t)b->Np);
    A_(pair_graph, (size_t)b->Np);
    A_(pair_off, B + 1);
    A_(ef_tile0, (size_t)std::max(N, 1));
    if (rc == MI_OK) rc = dev_alloc(b, &b->ef_mask, (size_t)std::max(N, 1));
    A_(e_diag, knn ? 0 : N);
    A_(M1pl, planes_elems(E, H));
    A_(lnpl, planes_elems(N, H));
    A_(aggpl, planes_elems(N, H));
    A_(Xpl, planes_elems(N, H));
    A_(dsc, 16);
    A_(absmax, 4 * L + 2);   // [0, 2L): the forward's per-layer slots; [2L, 2L + 2): the backward's (seven-launch form); [2L + 2, 4L + 2): one pair per layer (fused backward chain)
    A_(nc_flags, (size_t)(L + 1) * 2 * cdiv(N, 32));
    A_(X, NH);
    A_(x1, NH);
    A_(tproj, (size_t)B * H);
    A_(hf, NH);
    b->x1_base = b->x1;   // (x1 / hf point into the weight-gradient window's slot while a training forward with an open window runs: net_forward)
    b->hf_base = b->hf;
    A_(temb, (size_t)B * net->TD);
    A_(times, B);
    A_(pred_l, (size_t)B * 9);
    A_(pred_x, (size_t)N * 3);
    A_(pred_t, (size_t)N * MI_NUM_TYPES);
    A_(x_mid, (size_t)N * 3);
    A_(lp_corr, B);
    if (N <= 2048) {  // small batches: node-level products are latency-bound, split-K partial sums live here
        b->sk.floats = (size_t)N * 2 * H * 8;
        A_(sk.buf, b->sk.floats);
    }
#undef A_
    if (rc != MI_OK) {
        mi_batch_destroy(b);
        return rc;
    }
    // M1 planes: the GEMM epilogue only writes rows < E; the row padding of the last tile must be finite
    if (hipMemset(b->M1pl, 0, planes_elems(E, H) * sizeof(unsigned short)) != hipSuccess ||
        hipMemset(b->lnpl, 0, planes_elems(N, H) * sizeof(unsigned short)) != hipSuccess ||
        hipMemset(b->aggpl, 0, planes_elems(N, H) * sizeof(unsigned short)) != hipSuccess ||
        hipMemset(b->Xpl, 0, planes_elems(N, H) * sizeof(unsigned short)) != hipSuccess ||
        hipMemset(b->absmax, 0, (4 * L + 2) * sizeof(unsigned)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {   // (the handle will be used on non-blocking side streams: its set-up on the null stream must have finished)
        mi_batch_destroy(b);
        set_error("hipMemset failed");
        return MI_EHIP;
    }
    auto up = [&](int* d, const std::vector<int>& h) {
        return h.empty() ? hipSuccess : hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    };
    hipError_t he = up(b->num_atoms, b->num_atoms_h);
    if (he == hipSuccess) he = up(b->node_off, b->node_off_h);
    if (he == hipSuccess) he = up(b->node2graph, n2g);
    if (he == hipSuccess) he = up(b->src, src);
    if (he == hipSuccess) he = up(b->dst, dst);
    if (he == hipSuccess) he = up(b->edge_graph, egraph);
    if (he == hipSuccess) he = up(b->rowptr, rowptr);
    if (he == hipSuccess) he = up(b->pair_i, pr_i);
    if (he == hipSuccess) he = up(b->pair_j, pr_j);
    if (he == hipSuccess) he = up(b->pair_e1, pr_e1);
    if (he == hipSuccess) he = up(b->pair_e2, pr_e2);
    if (he == hipSuccess) he = up(b->pair_graph, pr_g);
    if (he == hipSuccess) he = up(b->pair_off, pr_off);
    if (he == hipSuccess) he = up(b->ef_tile0, ef_tile0);
    if (he == hipSuccess) he = hipMemcpy(b->ef_mask, ef_mask.data(), ef_mask.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = up(b->e_diag, ediag);
    if (he != hipSuccess) {
        set_error("index table upload failed: %s", hipGetErrorString(he));
        mi_batch_destroy(b);
        return MI_EHIP;
    }
    *out = b;
    return MI_OK;
}

void mi_batch_destroy(mi_batch* b) {
    if (!b) return;
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    if (b->hi_stream) (void)hipStreamDestroy(b->hi_stream);
    for (hipEvent_t e : b->hi_ev)
        if (e) (void)hipEventDestroy(e);
    for (void* p : b->allocs) (void)hipFree(p);
    delete b;
}

int mi_batch_create(const mi_net* net, const int* num_atoms_host, int B, int64_t node_offset, int64_t graph_offset, mi_batch** out) {
    return batch_create_impl(net, num_atoms_host, B, node_offset, graph_offset, false, 0, 0, out);
}

int mi_batch_create_knn(const mi_net* net, const int* num_atoms_host, int B, int64_t node_offset, int64_t graph_offset, int max_neighbors,
                        int edge_cap_per_node, mi_batch** out) {
    return batch_create_impl(net, num_atoms_host, B, node_offset, graph_offset, true, max_neighbors, edge_cap_per_node, out);
}

int mi_batch_num_nodes(const mi_batch* b) { return b ? b->N : 0; }
int64_t mi_batch_num_edges(const mi_batch* b) { return b ? b->E : 0; }
const int* mi_batch_node2graph(const mi_batch* b) { return b ? b->node2graph : nullptr; }

int mi_cspnet_forward(mi_net* net, mi_batch* b, const float* t_emb, const float* atom_types, const float* frac,
                      const float* lattices, float* lattice_out, float* coord_out, float* type_out, void* stream) {
    MI_CHECK(net && b, MI_EINVAL, "null handle");
    MI_CHECK(b->H == net->H && b->L == net->L, MI_EINVAL, "batch was created for a different network");
    return net_forward(net, b, t_emb, atom_types, frac, lattices, lattice_out, coord_out, type_out, (hipStream_t)stream);
}

int mi_cspnet_tap(mi_net* net, mi_batch* b, int layer, float* out, void* stream) {
    MI_CHECK(net && b && out, MI_EINVAL, "null argument");
    MI_CHECK(layer >= 0 && layer <= net->L + 1, MI_EINVAL, "layer %d out of range", layer);
    const size_t NH = (size_t)b->N * net->H;
    const float* src = layer <= net->L ? b->h + (size_t)layer * NH : b->hf;
    MI_HIP(hipMemcpyAsync(out, src, NH * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MI_OK;
}

int mi_set_gemm_mode(int mode) {
    MI_CHECK(mode == MI_GEMM_F32 || mode == MI_GEMM_SPLIT, MI_EINVAL, "unknown gemm mode %d", mode);
    mi::g_gemm_mode = mode;
    return MI_OK;
}

// The pair-mode Fourier operand of `Np` atom pairs, read back as fp32: out[p][0:3F] = sin(2 pi k d_c), out[p][3F:6F] = cos(...) in
// SinusoidsEmbedding's column order (c * F + k), each value the exact sum of its planes divided by the plane scale.
__global__ void debug_fourier_read_kernel(mi::Planes FF, int64_t Np, int F, int Kh, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int F3 = 3 * F;
    if (idx >= Np * 2 * F3) return;
    const int64_t p = idx / (2 * F3);
    const int c = (int)(idx % (2 * F3)), col = c < F3 ? c : Kh + (c - F3);
    float v = 0.f;
    for (int pl = mi::NPL - 1; pl >= 0; --pl) {
        const mi::u16 w = FF.base[FF.elem((int)p, col, pl)];
#if MI_PLANES_FP16
        v += (float)__builtin_bit_cast(_Float16, w);
#else
        v += __uint_as_float((unsigned)w << 16);
#endif
    }
    out[idx] = v / FF.s();
}
int mi_debug_fourier_pairs(const float* frac, const int* pair_i, const int* pair_j, int64_t Np, int F, float* out, void* stream) {
    MI_CHECK(frac && pair_i && pair_j && out && Np > 0 && F > 0, MI_EINVAL, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int Kh = (3 * F + 31) / 32 * 32;
    mi::u16* buf = nullptr;
    MI_HIP(hipMalloc((void**)&buf, mi::planes_elems(Np, 2 * Kh) * sizeof(mi::u16)));
    mi::Planes ffp = mi::make_planes(buf, 2 * Kh, mi::PL_S_UNIT);
    const int64_t nthr = (Np + 127) / 128 * 128 * (int64_t)(Kh / 8);
    if (F % 8 == 0) hipLaunchKernelGGL(mi::fourier_pair_planes_kernel<true>, dim3((unsigned)mi::cdiv(nthr, 256)), dim3(256), 0, s, frac, pair_i, pair_j, ffp, Np, F, Kh);
    else hipLaunchKernelGGL(mi::fourier_pair_planes_kernel<false>, dim3((unsigned)mi::cdiv(nthr, 256)), dim3(256), 0, s, frac, pair_i, pair_j, ffp, Np, F, Kh);
    hipLaunchKernelGGL(debug_fourier_read_kernel, dim3((unsigned)mi::cdiv(Np * 6 * F, 256)), dim3(256), 0, s, ffp, Np, F, Kh, out);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(buf);
    MI_HIP(e);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_debug_set_db_min_tiles(int n) {
    if (n < 0) g_pair_kernel = 1;  // negative: also let pair-mode GEMMs pick the kernel by size
    g_planes_db_min_tiles = n < 0 ? -n : n;
    return MI_OK;
}

int mi_debug_set_planes_small_tiles(int n) {
    g_planes_small_tiles = n;
    return MI_OK;
}

int mi_debug_set_planes_big(int on, int min_rows) {
    g_planes_big = on;   // 0 = off, 1 = products without epilogue extensions, 2 = every qualifying product
    if (min_rows > 0) g_planes_big_min_rows = min_rows;
    return MI_OK;
}

int mi_debug_set_planes_rt(int mode, int min_rows) {
    g_planes_rt = mode;   // 0 = off, 1 = products with epilogue extensions, 2 = every qualifying product
    if (min_rows > 0) g_planes_rt_min_rows = min_rows;
    return MI_OK;
}

int mi_debug_set_planes_latency(int max_blocks) {
    g_planes_lat_max_blocks = max_blocks;   // 0 = the latency form is never used
    return MI_OK;
}

int mi_debug_set_node_priority(int on) {
    g_fused_heads = (on & 2) == 0;   // +2: the two-GEMM heads (ablation)
    g_node_hi = on & 1;
    return MI_OK;
}

int mi_debug_set_planes_big_seg(int min_rows) {
    MI_CHECK(min_rows <= 0 || MI_HAVE_ABLATION_KERNELS, MI_EINVAL, "the segmented-sum epilogue on the 256 x 256 kernel is an ablation instantiation: rebuild with "
             "MI_EXTRA_FLAGS=-DMI_ABLATION_KERNELS");
    g_planes_big_seg_min_rows = min_rows;
    return MI_OK;
}

int mi_debug_set_planes_dma(int mode) {
    MI_CHECK(mode < 4 || MI_HAVE_ABLATION_KERNELS, MI_EINVAL, "mode 4 (the four-waves-per-SIMD build of the 128-row loop) is an ablation instantiation: rebuild with "
             "MI_EXTRA_FLAGS=-DMI_ABLATION_KERNELS");
    g_planes_dma = mode;
    return MI_OK;
}

int mi_plane_format(void) { return NPL; }
int mi_terms_per_product(void) { return MI_TF32_CLASS ? 1 : 3; }
int mi_debug_mfma_flops(double* flops16, double* flops32, int reset) {
    if (flops16) *flops16 = 1e6 * (double)mi::g_mfma16_mflop.load();
    if (flops32) *flops32 = 1e6 * (double)mi::g_mfma32_mflop.load();
    if (reset) {
        mi::g_mfma16_mflop.store(0);
        mi::g_mfma32_mflop.store(0);
    }
    return MI_OK;
}

int mi_debug_set_tn128(int on) {
    g_tn128 = (on & 1) != 0;
    g_tn_split = (on & 2) != 0;
    g_tn_xsilu = (on & 16) == 0;         // +16: separate silu(Z1) pass instead of forming M1 inside the weight-gradient product
    g_bwd_pairs_fused = (on & 8) == 0;  // +8: the separate dZ1 consumers instead of the fused pair-mode backward pass  // 0: 64x64 f32, 1: 128x128 f32, 3 (default): bf16 three-plane split on the split path
    g_bwd_dz2_planes = (on & 32) == 0;  // +32: dM1 data gradient on the on-the-fly three-plane bf16 split instead of the fp16 plane GEMM
    g_bwd_pairs_tile = (on & 128) == 0;  // +128: the thread-per-column form of the fused pair-mode backward pass instead of the LDS-tile form
    g_bwd_wgrad_planes = (on & 256) == 0;   // +256: edge_mlp.2's weight gradient from fp32 rows (split, SiLU and transposition on the way into LDS) instead of from the M1 / dZ2 plane sets
    g_bwd_head_window = (on & 512) == 0;   // +512: the head / embedding weight gradients in every backward instead of in the deferred window (applies to windows sized afterwards)
    g_bwd_wgrad_f16 = (on & 64) == 0;   // +64: edge-level weight gradients on three bf16 planes / six terms instead of two fp16 planes / three
    return MI_OK;
}

int mi_debug_set_eval_reuse(int mask) {
    const int was = g_eval_reuse;
    g_eval_reuse = mask;
    return was;
}

int mi_debug_set_heads_rows16(int min_nodes) {
    const int was = g_heads_rows16_min_nodes;
    g_heads_rows16_min_nodes = min_nodes;
    return was;
}

int mi_debug_set_pair_wide(int on) {
    const int was = g_pair_wide_force;
    g_pair_wide_force = on != 0;
    return was;
}

int mi_debug_set_knn_nosync(int on) {
    const int was = g_knn_nosync;
    g_knn_nosync = on < 0 ? 0 : (on > 2 ? 2 : on);
    return was;
}

int mi_set_concurrent_groups(int n) {
    const int was = g_concurrent_groups;
    g_concurrent_groups = n < 1 ? 1 : (n > 16 ? 16 : n);
    return was;
}

int mi_debug_set_tn_target_tiles(int n) {
    const int was = g_tn_target_tiles;
    if (n > 0) g_tn_target_tiles = n;
    return was;
}

int mi_debug_set_tn_split_min_rows(int n) {
    g_tn_split_min_rows = n;
    return MI_OK;
}

int mi_debug_set_node_planes_min_rows(int n) {
    g_node_planes_min_rows = n;
    return MI_OK;
}

int mi_set_edge_pairs(int on) {
    g_edge_pairs = on != 0;
    return MI_OK;
}

int mi_net_set_edge_mode(mi_net* net, int mode) {
    MI_CHECK(net && (mode == MI_EDGE_FUSED_F32 || mode == MI_EDGE_GEMM), MI_EINVAL, "unknown edge mode %d", mode);
    MI_CHECK(mode != MI_EDGE_FUSED_F32 || net->H == 64 || net->H == 128 || net->H == 256 || net->H == 512, MI_EINVAL,
             "the register-chained f32 edge kernel exists for hidden_dim 64/128/256/512, not %d", net->H);
    net->edge_mode = mode;
    return MI_OK;
}

int mi_saturation_events(int64_t* count, int reset) {
    MI_CHECK(count, MI_EINVAL, "null argument");
    MI_HIP(hipDeviceSynchronize());
    int64_t total = 0;
    for (auto fetch : mi::sat_readers()) {   // one reader per translation unit that converts to the plane format
        unsigned v = 0;
        MI_TRY(fetch(&v, reset != 0));
        total += v;
    }
    *count = total;
    return MI_OK;
}

int mi_profile_enable(mi_net* net, int on) {
    MI_CHECK(net, MI_EINVAL, "null handle");
    std::lock_guard<std::mutex> g(net->prof_mu);
    net->prof = on != 0;
    net->ev_used = 0;
    if (on) {
        if (!net->ev_origin) MI_HIP(hipEventCreate(&net->ev_origin));
        MI_HIP(hipEventRecord(net->ev_origin, nullptr));
    }
    return MI_OK;
}

int mi_profile_read(mi_net* net, int64_t* launches, double* total_ms, double* union_ms) {
    MI_CHECK(net && launches && total_ms, MI_EINVAL, "null argument");
    std::lock_guard<std::mutex> g(net->prof_mu);
    double tot = 0;
    int64_t n = 0;
    std::vector<std::pair<float, float>> iv;
    for (size_t k = 0; k + 1 < net->ev_used; k += 2) {
        MI_HIP(hipEventSynchronize(net->ev[k + 1]));
        float ms = 0;
        MI_HIP(hipEventElapsedTime(&ms, net->ev[k], net->ev[k + 1]));
        tot += ms;
        ++n;
        if (union_ms && net->ev_origin) {
            float t0 = 0;
            MI_HIP(hipEventElapsedTime(&t0, net->ev_origin, net->ev[k]));
            iv.emplace_back(t0, t0 + ms);
        }
    }
    if (union_ms) {  // time during which at least one bracketed stage was executing (launches on concurrent streams overlap)
        std::sort(iv.begin(), iv.end());
        double u = 0, lo = 0, hi = -1;
        for (auto& x : iv) {
            if (hi < 0 || x.first > hi) {
                if (hi >= 0) u += hi - lo;
                lo = x.first;
                hi = x.second;
            } else if (x.second > hi) {
                hi = x.second;
            }
        }
        if (hi >= 0) u += hi - lo;
        *union_ms = u;
    }
    *launches = n;
    *total_ms = tot;
    net->ev_used = 0;
    return MI_OK;
}

}  // extern "C"

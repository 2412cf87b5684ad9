// Backward of the CSPNet score network (parameter gradients only: the fine-tune step never needs
// gradients w.r.t. the noised inputs), fused Adam on the flat parameter vector, and forward
// noising.  Reference: autograd through models/diffcsp/cspnet.py:260-294 as driven by
// pipeline/mat_invent.py:150-177; torch.optim.Adam (:136); DiffCSPModule.add_noise
// (models/diffcsp/diffusion.py:81-119).
//
// v1 structure: the edge stage is unfused -- dZ2, M1, dZ1 and the Fourier features are materialised
// as [E,H] / [E,6F] scratch and pushed through the generic MFMA GEMMs (dgrad = gemm_nt against
// pre-transposed weights, wgrad = gemm_tn with a fixed-order split reduction).  Deterministic: no
// float atomics anywhere.
#include "gemm_split.h"
#include "net.h"

namespace mi {
extern int g_edge_pairs;
int g_tn_xsilu = 1;          // M1 = silu(Z1) inside the weight-gradient product's operand load (0: separate pass, ablation)
int g_bwd_pairs_tile = 1;   // crystals of at most 24 atoms: the LDS-tile form of the fused pass (0: the thread-per-column form, ablation)
int g_bwd_pairs_fused = 1;  // fc pair mode: one fused pass for every consumer of dZ1 (0: the separate kernels, ablation)
int g_bwd_dz2_planes = 1;   // fp16 plane format: dZ2 also as a plane set, its data gradient on the pre-split plane GEMM (0: on-the-fly bf16 split)
int g_bwd_wgrad_planes = 1; // fp16 plane format: edge_mlp.2's weight gradient from the plane sets of M1 (kept per layer by the training forward) and dZ2 (0: fp32 rows re-split on the way into LDS)
int g_bwd_head_window = 1;  // the head / embedding weight gradients join the deferred window of the node-level ones (0: contracted in every backward; read when a window is sized)
int g_bwd_wgrad_f16 = 1;    // fp16 plane format: edge-level weight gradients on two fp16 planes / three terms (0: three bf16 planes / six)
}

namespace mi {

__global__ void transpose_kernel(const float* __restrict__ src, int ld_src, int rows, int cols, float* __restrict__ dst) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    int c = idx / rows, r = idx % rows;  // dst[c][r]
    dst[idx] = src[(size_t)r * ld_src + c];
}

// y = dy * silu'(z)   (in place allowed)
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dy[i] * silu_grad(z[i]);
}
__global__ void fill_int_kernel2(int* p, int v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
__global__ void silu_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = silu(z[i]);
}

// dZ2[e][f] = dagg[src(e)][f] / deg(src) * silu'(Z2[e][f])   (dagg = dcat[:, H:2H]); in place over Z2
__global__ void edge_dz2_kernel(const float* __restrict__ dcat, const int* __restrict__ src, const int* __restrict__ rowptr,
                                float* __restrict__ Z2, int64_t E, int H) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * H) return;
    int64_t e = idx / H;
    int f = (int)(idx % H);
    int i = src[e];
    float deg = (float)(rowptr[i + 1] - rowptr[i]);
    Z2[idx] = (dcat[(size_t)i * (2 * H) + H + f] / deg) * silu_grad(Z2[idx]);
}

// max |x| over n floats (16-byte loads) -> atomicMax of the bit pattern (non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void absmax_bwd_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// The same with the bias gradient's column sums folded in (one pass over [E, H] less): a block owns a chunk of `rows` rows, writes dZ2
// in place and the chunk's column sums to part[chunk][H] (reduced by part_reduce_kernel).  A thread owns FOUR consecutive columns
// (16-byte accesses: a wave moves 1 KiB of a row per instruction; one column per thread ran at 2.1 TB/s) and every second row of
// the chunk -- the two row phases of a 256-thread block over H <= 512 columns are combined through LDS in a fixed order.
// Needs H % 4 == 0.
__global__ __launch_bounds__(256) void edge_dz2_colsum_kernel(const float* __restrict__ dcat, const int* __restrict__ src,
                                                              const int* __restrict__ rowptr, float* __restrict__ Z2, float* __restrict__ part,
                                                              int64_t E, int H, int rows, Planes dzp = Planes(),
                                                              const unsigned* __restrict__ amax = nullptr, float* __restrict__ dsc_out = nullptr,
                                                              int store_f32 = 1) {   // (0: every consumer of dZ2 reads the plane set -- its fp32 rows are not written)
    const int q = H / 4;                                     // column quads per row
    // optional: dZ2 also as an fp16 plane set (the A operand of the dM1 data gradient on the pre-split plane GEMM).  Its
    // power-of-two scale follows from a rigorous bound, |dZ2| <= max|d cat| * max|silu'| (degrees are >= 1, |silu'| < 1.1):
    // every thread derives it, block 0 publishes {scale, 1 / scale} for the GEMM.
    float ps = 1.f;
    if (dzp.base) {
        const float bnd = 1.1f * __uint_as_float(amax[0]);
        int ex = 14 - (int)ceilf(log2f(fmaxf(bnd, 1e-30f)));
        if (!(bnd == bnd) || bnd > 3e38f) ex = -100;
        ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
        ps = exp2f((float)ex);
        if (blockIdx.y == 0 && threadIdx.x == 0) {
            dsc_out[0] = ps;
            dsc_out[1] = exp2f(-(float)ex);
        }
    }
    const int64_t e0 = (int64_t)blockIdx.y * rows, e1 = e0 + rows < E ? e0 + rows : E;
    __shared__ f32x4 comb[256];
    for (int cq0 = 0; cq0 < q; cq0 += 256) {                 // (H > 1024: several column passes)
        const int qq = q - cq0 < 256 ? q - cq0 : 256;        // quads handled in this pass
        const int ph = threadIdx.x / qq, cq = cq0 + threadIdx.x % qq;
        const int nph = 256 / qq;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (ph < nph) {
#pragma unroll 4
            for (int64_t e = e0 + ph; e < e1; e += nph) {
                const int i = src[e];
                const float deg = (float)(rowptr[i + 1] - rowptr[i]);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dcat + (size_t)i * (2 * H) + H + 4 * cq);
                float* zp = Z2 + e * H + 4 * cq;
                const f32x4 z = *reinterpret_cast<const f32x4*>(zp);
                f32x4 v;
                // (the mean's 1 / deg stays an IEEE division -- it is exact for the powers of two and shared by a row; silu' on the hardware transcendentals)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (d[k] / deg) * silu_grad_fast(z[k]);
                if (store_f32) *reinterpret_cast<f32x4*>(zp) = v;
                if (dzp.base) {
                    unsigned lo[3], hi[3];
                    pl_split_pair(v[0], v[1], ps, lo);
                    pl_split_pair(v[2], v[3], ps, hi);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        uint2 pk = {lo[pl], hi[pl]};
                        *reinterpret_cast<uint2*>(dzp.base + dzp.elem((int)e, 4 * cq, pl)) = pk;
                    }
                }
                sum += v;
            }
        }
        comb[threadIdx.x] = sum;
        __syncthreads();
        if (ph == 0) {
            for (int p2 = 1; p2 < nph; ++p2) sum += comb[p2 * qq + threadIdx.x];
            *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * H + 4 * cq) = sum;
        }
        __syncthreads();
    }
}

__global__ void fourier_kernel(const float* __restrict__ frac, const float* __restrict__ fd, const int* __restrict__ src,
                               const int* __restrict__ dst, float* __restrict__ FF, int64_t E, int F);

// dPQ[i][0:H]  = sum_j dZ1[(i,j)]      (row run of node i)
// dPQ[j][H:2H] = sum_i dZ1[(i,j)]      (column of node j inside its fully connected crystal)
__global__ void edge_dpq_kernel(const float* __restrict__ dZ1, const int* __restrict__ rowptr, const int* __restrict__ node2graph,
                                const int* __restrict__ node_off, float* __restrict__ dPQ, int N, int H) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * H) return;
    int i = (int)(idx / H), f = (int)(idx % H);
    float s = 0.f;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) s += dZ1[(size_t)e * H + f];
    dPQ[(size_t)i * (2 * H) + f] = s;
    int g = node2graph[i], n0 = node_off[g], n1 = node_off[g + 1], jl = i - n0;
    float t = 0.f;
    for (int ii = n0; ii < n1; ++ii) t += dZ1[(size_t)(rowptr[ii] + jl) * H + f];
    dPQ[(size_t)i * (2 * H) + H + f] = t;
}

// pair-mode weight gradient of the Fourier block (fc edge list): with S = sin, C = cos of the pair's arguments,
//   dW_sin += sum_p (dZ1[i->j] - dZ1[j->i])^T S_p,   dW_cos += sum_p (dZ1[i->j] + dZ1[j->i])^T C_p  (+ the self edges, cos = 1)
__global__ void pair_combine_kernel(const float* __restrict__ dZ1, const int* __restrict__ e1, const int* __restrict__ e2,
                                    float* __restrict__ Dm, float* __restrict__ Dp, int64_t Np, int H) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Np * (H / 4)) return;
    const int64_t p = idx / (H / 4);
    const int f = (int)(idx % (H / 4)) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(dZ1 + (size_t)e1[p] * H + f), b = *reinterpret_cast<const f32x4*>(dZ1 + (size_t)e2[p] * H + f);
    *reinterpret_cast<f32x4*>(Dm + (size_t)p * H + f) = a - b;
    *reinterpret_cast<f32x4*>(Dp + (size_t)p * H + f) = a + b;
}
// Everything of the fc pair-mode backward that consumes dZ1 = dM1 * silu'(Z1), in ONE pass over a crystal's edge block (instead of
// the silu' pass, pair_combine, the two-read dPQ sums, the self-edge column sum and the per-graph sum: 7 -> 3 passes over [E, H]):
//   Dm[p] = dZ1[i->j] - dZ1[j->i],  Dp[p] = dZ1[i->j] + dZ1[j->i]                 (pair-mode weight gradient operands)
//   dPQ[i][0:H] = sum_j dZ1[(i,j)],  dPQ[j][H:2H] = sum_i dZ1[(i,j)],  dG[g] = sum_i dPQ[i][0:H]
//   dsum_part[g] = sum_i dZ1[(i,i)]                                               (self edges: cosine block's constant column)
// One block per (crystal, 128-column slice), a thread per column; the per-node row / column sums live in LDS ([2][n][128] floats,
// each thread only touches its own column: no atomics, fixed order).  Edge (a -> b) of the crystal sits at e0 + a*n + b.
__global__ __launch_bounds__(128) void edge_bwd_pairs_kernel(const float* __restrict__ dM1, const float* __restrict__ Z1,
                                                             const int* __restrict__ node_off, const int* __restrict__ rowptr,
                                                             const int* __restrict__ pair_off, float* __restrict__ Dm, float* __restrict__ Dp,
                                                             float* __restrict__ dPQ, float* __restrict__ dG, float* __restrict__ dsum_part,
                                                             int H, const unsigned* __restrict__ amax = nullptr, float* __restrict__ dsc_out = nullptr) {
    extern __shared__ float accs[];  // row sums [n][128], then column sums [n][128]
    // fp16 plane format: publish the scales of the Fourier-block weight gradient's operands.  |Dm|, |Dp| <= 2 max|dZ1| and
    // |dZ1| = |dM1 silu'(Z1)| <= 1.1 max|dM1| (amax: written by the dM1 product's epilogue); the Fourier features are <= 1.
    if (dsc_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const float bnd = 2.2f * __uint_as_float(amax[0]);
        int ex = 14 - (int)ceilf(log2f(fmaxf(bnd, 1e-30f)));
        if (!(bnd == bnd) || bnd > 3e38f) ex = -100;
        ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
        dsc_out[0] = exp2f((float)ex);
        dsc_out[1] = exp2f(-(float)ex);
        dsc_out[2] = 16384.f;
        dsc_out[3] = 1.f / 16384.f;
    }
    const int g = blockIdx.x, tid = threadIdx.x, c = blockIdx.y * 128 + tid;
    const int o = node_off[g], n = node_off[g + 1] - o;
    if (n == 0 || c >= H) return;
    const int64_t e0 = rowptr[o];
    float* row = accs;
    float* col = accs + (size_t)n * 128;
    float ds = 0.f;
    for (int i = 0; i < n; ++i) {  // self edges first: they initialise the accumulators
        const size_t a = (size_t)(e0 + (int64_t)i * n + i) * H + c;
        const float v = dM1[a] * silu_grad(Z1[a]);
        row[i * 128 + tid] = v;
        col[i * 128 + tid] = v;
        ds += v;
    }
    int64_t p = pair_off[g];
    for (int i = 0; i < n; ++i) {
        float ri = row[i * 128 + tid], ci = col[i * 128 + tid];
        for (int j = i + 1; j < n; ++j, ++p) {
            const size_t a1 = (size_t)(e0 + (int64_t)i * n + j) * H + c, a2 = (size_t)(e0 + (int64_t)j * n + i) * H + c;
            const float v1 = dM1[a1] * silu_grad(Z1[a1]), v2 = dM1[a2] * silu_grad(Z1[a2]);
            Dm[(size_t)p * H + c] = v1 - v2;
            Dp[(size_t)p * H + c] = v1 + v2;
            ri += v1;                    // i -> j: row i, column j
            ci += v2;                    // j -> i: row j, column i
            row[j * 128 + tid] += v2;
            col[j * 128 + tid] += v1;
        }
        row[i * 128 + tid] = ri;
        col[i * 128 + tid] = ci;
    }
    float gs = 0.f;
    for (int i = 0; i < n; ++i) {
        const float r = row[i * 128 + tid];
        dPQ[(size_t)(o + i) * (2 * H) + c] = r;
        dPQ[(size_t)(o + i) * (2 * H) + H + c] = col[i * 128 + tid];
        gs += r;
    }
    dG[(size_t)g * H + c] = gs;
    dsum_part[(size_t)g * H + c] = ds;
}

// The same pass for crystals of at most PAIRS_NMAX atoms, restructured for parallelism: the kernel above walks a crystal's 190 pairs
// serially in each thread with (B x H / 128) two-wave blocks -- 340 blocks on 256 CUs at the benchmark set, a chain of load latencies
// (137 us per layer).  Here a block owns a (crystal, 32-column) slice: its dZ1 = dM1 silu'(Z1) tile [n x n][32] goes through LDS once
// (eight row workers x 32 columns, coalesced 128-byte pieces), then every output is formed from LDS in a fixed order: pairs round-robin
// over the workers, row / column sums per node, the per-crystal sums by one worker each.  4x the blocks, 4 waves each.
constexpr int PAIRS_NMAX = 24, PAIRS_W = 32, PAIRS_WORKERS = 8;
__global__ __launch_bounds__(256) void edge_bwd_pairs_tile_kernel(const float* __restrict__ dM1, const float* __restrict__ Z1,
                                                                  const int* __restrict__ node_off, const int* __restrict__ rowptr,
                                                                  const int* __restrict__ pair_off, float* __restrict__ Dm, float* __restrict__ Dp,
                                                                  float* __restrict__ dPQ, float* __restrict__ dG, float* __restrict__ dsum_part,
                                                                  int H, const unsigned* __restrict__ amax = nullptr, float* __restrict__ dsc_out = nullptr,
                                                                  Planes DmP = Planes(), Planes DpP = Planes()) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // dZ1 [n * n][32] | row sums [n][32]   (16-byte pieces: the base must be 16-aligned behind the static table)
    __shared__ unsigned short pij[PAIRS_NMAX * (PAIRS_NMAX - 1) / 2];   // pair k -> (i << 8) | j
    float pl_scale = 1.f;   // scale of the pair differences / sums (every block computes it: the plane-set form below needs it before block 0 has published it)
    if (amax) {
        const float bnd = 2.2f * __uint_as_float(amax[0]);
        int ex = 14 - (int)ceilf(log2f(fmaxf(bnd, 1e-30f)));
        if (!(bnd == bnd) || bnd > 3e38f) ex = -100;
        ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
        pl_scale = exp2f((float)ex);
        if (dsc_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // (scales of the weight-gradient operands: see the kernel above)
            dsc_out[0] = pl_scale;
            dsc_out[1] = exp2f(-(float)ex);
            dsc_out[2] = 16384.f;
            dsc_out[3] = 1.f / 16384.f;
            dsc_out[4] = PL_S_UNIT;          // (the forward's pair-mode Fourier plane set, read as it is by gemm_tn_planes)
            dsc_out[5] = 1.f / PL_S_UNIT;
        }
    }
    const int g = blockIdx.x, tid = threadIdx.x, lc = tid & (PAIRS_W - 1), w = tid >> 5, c = blockIdx.y * PAIRS_W + lc;
    const int o = node_off[g], n = node_off[g + 1] - o;
    if (n == 0) return;
    const bool cok = c < H;
    const int64_t e0 = rowptr[o];
    const int nn = n * n, np = n * (n - 1) / 2;
    float* rs = tile + (size_t)nn * PAIRS_W;
    // dZ1 tile: eight lanes x 16 bytes cover a row's 32 columns, a 256-thread pass 32 rows; four passes' loads are issued before the first
    // value is used (one 4-byte load per lane and row with the libm silu' in between ran this pass at 2 TB/s: 313 us per layer at 256 crystals)
    const int rq = tid >> 3, c4 = (tid & 7) * 4, cq = blockIdx.y * PAIRS_W + c4;
    const bool vec = (H & 3) == 0 && cq + 3 < H;
    for (int eb = 0; eb < nn; eb += 128) {
        f32x4 d[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = eb + 32 * u + rq;
            d[u] = z[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < nn) {
                const size_t a = (size_t)(e0 + e) * H + cq;
                if (vec) {
                    d[u] = *reinterpret_cast<const f32x4*>(dM1 + a);
                    z[u] = *reinterpret_cast<const f32x4*>(Z1 + a);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cq + k < H) {
                            d[u][k] = dM1[a + k];
                            z[u][k] = Z1[a + k];
                        }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = eb + 32 * u + rq;
            if (e < nn) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = d[u][k] * silu_grad_fast(z[u][k]);   // (columns past H: 0 * silu'(0) = 0)
                *reinterpret_cast<f32x4*>(tile + e * PAIRS_W + c4) = v;
            }
        }
    }
    for (int i = tid; i < n; i += 256) {   // pair table in the order of the pair list (i < j, row-major)
        int k = i * n - i * (i + 1) / 2;
        for (int j = i + 1; j < n; ++j, ++k) pij[k] = (unsigned short)((i << 8) | j);
    }
    __syncthreads();
    const int64_t p0 = pair_off[g];
    for (int k = rq; k < np; k += 32) {   // the pair rows of the weight-gradient operands: 16-byte pieces, 32 pairs per pass
        const int i = pij[k] >> 8, j = pij[k] & 255;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(tile + (i * n + j) * PAIRS_W + c4), v2 = *reinterpret_cast<const f32x4*>(tile + (j * n + i) * PAIRS_W + c4);
        const f32x4 dm = v1 - v2, dp = v1 + v2;
        const size_t a = (size_t)(p0 + k) * H + cq;
        if (DmP.base) {   // as plane sets (H % 32 == 0: the four columns lie in one 32-column tile): the weight gradient reads them as they are
            unsigned m01[3], m23[3], s01[3], s23[3];
            pl_split_pair(dm[0], dm[1], pl_scale, m01);
            pl_split_pair(dm[2], dm[3], pl_scale, m23);
            pl_split_pair(dp[0], dp[1], pl_scale, s01);
            pl_split_pair(dp[2], dp[3], pl_scale, s23);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                *reinterpret_cast<u32x2*>(DmP.base + DmP.elem((int)(p0 + k), cq, pl)) = u32x2{m01[pl], m23[pl]};
                *reinterpret_cast<u32x2*>(DpP.base + DpP.elem((int)(p0 + k), cq, pl)) = u32x2{s01[pl], s23[pl]};
            }
        } else if (vec) {
            *reinterpret_cast<f32x4*>(Dm + a) = dm;
            *reinterpret_cast<f32x4*>(Dp + a) = dp;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (cq + q < H) {
                    Dm[a + q] = dm[q];
                    Dp[a + q] = dp[q];
                }
        }
    }
    for (int i = w; i < n; i += PAIRS_WORKERS) {   // dPQ[i][0:H] = sum_j dZ1[(i,j)],  dPQ[i][H:2H] = sum_i' dZ1[(i',i)]
        float r = 0.f, cs = 0.f;
        for (int j = 0; j < n; ++j) {
            r += tile[(i * n + j) * PAIRS_W + lc];
            cs += tile[(j * n + i) * PAIRS_W + lc];
        }
        rs[i * PAIRS_W + lc] = r;
        if (cok) {
            dPQ[(size_t)(o + i) * (2 * H) + c] = r;
            dPQ[(size_t)(o + i) * (2 * H) + H + c] = cs;
        }
    }
    __syncthreads();
    if (cok && w == 0) {
        float gs = 0.f;
        for (int i = 0; i < n; ++i) gs += rs[i * PAIRS_W + lc];
        dG[(size_t)g * H + c] = gs;
    } else if (cok && w == 1) {
        float ds = 0.f;
        for (int i = 0; i < n; ++i) ds += tile[(i * n + i) * PAIRS_W + lc];
        dsum_part[(size_t)g * H + c] = ds;
    }
}

// gW[f][c] += v[f] for c < ncols   (self edges: every cosine feature is 1)
__global__ void row_broadcast_add_kernel(const float* __restrict__ v, float* __restrict__ gW, int ld, int H, int ncols) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * ncols) return;
    gW[(size_t)(idx / ncols) * ld + idx % ncols] += v[idx / ncols];
}

// gW[f][c] += sum_z P[z][f] for c < ncols: the column sums of part_reduce_kernel (same reduction order) broadcast over a row block straight away --
// one launch instead of a memset, the reduction and row_broadcast_add_kernel.  32 columns f per block; gridDim.y blocks share a row block's `ncols` target columns
// (each repeats the small reduction -- nsplit x 32 floats from L2 -- and adds into its own column slice: 16 blocks alone took 27 us for 32 x 384 updates each).
__global__ __launch_bounds__(256) void part_reduce_bcast_kernel(const float* __restrict__ P, int nsplit, int ldp, float* __restrict__ gW, int ld, int H, int ncols) {
    __shared__ float red[8][32];
    __shared__ float tot[32];
    const int cl = threadIdx.x & 31, c = blockIdx.x * 32 + cl, rg = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < H) {
        int z = rg;
        for (; z + 24 < nsplit; z += 32) {
            s0 += P[(size_t)z * ldp + c];
            s1 += P[(size_t)(z + 8) * ldp + c];
            s2 += P[(size_t)(z + 16) * ldp + c];
            s3 += P[(size_t)(z + 24) * ldp + c];
        }
        for (; z < nsplit; z += 8) s0 += P[(size_t)z * ldp + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) tot[cl] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
    __syncthreads();
    const int cper = (ncols + gridDim.y - 1) / gridDim.y, cbeg = blockIdx.y * cper, cw = min(ncols, cbeg + cper) - cbeg;   // this block's column slice
    for (int idx = threadIdx.x; idx < 32 * cw; idx += 256) {
        const int f = blockIdx.x * 32 + idx / cw;
        if (f < H) gW[(size_t)f * ld + cbeg + idx % cw] += tot[idx / cw];
    }
}

// part_reduce_kernel over `gridDim.y` independent problems of one shape: problem y reads P + y * p_stride and adds into out + y * out_stride
// (the LayerNorm weight / bias gradients of all layers in one launch: every layer's parameter block has the same size)
__global__ __launch_bounds__(256) void part_reduce_batched_kernel(const float* __restrict__ P, size_t p_stride, int nsplit, int ldp, float* __restrict__ out,
                                                                  size_t out_stride, int Nc) {
    __shared__ float red[8][32];
    P += (size_t)blockIdx.y * p_stride;
    out += (size_t)blockIdx.y * out_stride;
    const int cl = threadIdx.x & 31, c = blockIdx.x * 32 + cl, rg = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < Nc) {
        int z = rg;
        for (; z + 24 < nsplit; z += 32) {
            s0 += P[(size_t)z * ldp + c];
            s1 += P[(size_t)(z + 8) * ldp + c];
            s2 += P[(size_t)(z + 16) * ldp + c];
            s3 += P[(size_t)(z + 24) * ldp + c];
        }
        for (; z < nsplit; z += 8) s0 += P[(size_t)z * ldp + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < Nc)
        out[c] += ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
}

// general edge lists (knn branch): out-edges of i are its CSR row, in-edges are listed in `inedge` at the same offsets
__global__ void edge_dpq_csr_kernel(const float* __restrict__ dZ1, const int* __restrict__ rowptr, const int* __restrict__ inedge,
                                    float* __restrict__ dPQ, int N, int H) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * H) return;
    int i = (int)(idx / H), f = (int)(idx % H);
    float s = 0.f, t = 0.f;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        s += dZ1[(size_t)e * H + f];
        t += dZ1[(size_t)inedge[e] * H + f];
    }
    dPQ[(size_t)i * (2 * H) + f] = s;
    dPQ[(size_t)i * (2 * H) + H + f] = t;
}

// out[g][f] = sum over the nodes of crystal g of X[i][f]  (X row stride ldx)
__global__ void graph_sum_kernel(const float* __restrict__ X, int ldx, const int* __restrict__ node_off, float* __restrict__ out,
                                 int B, int H) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    int g = idx / H, f = idx % H;
    float s = 0.f;
    for (int i = node_off[g]; i < node_off[g + 1]; ++i) s += X[(size_t)i * ldx + f];
    out[idx] = s;
}

// gram term backward: gW1[f][2H+m] += sum_b dG[b][f] * gram_b[m];  gb1[f] += sum_b dG[b][f]
// 32 features x 8 graph groups per block (graph sums combined through LDS in a fixed order): with one thread per feature looping
// over all graphs the whole job was two workgroups and 96 us of latency.
// gridDim.y > 1: one layer per blockIdx.y -- dG of layer y at dG + y * dg_stride, its weight / bias gradient at gW1 / gb1 + y * w_stride (every layer's
// parameter block has the same size): all layers of a backward pass in ONE launch behind the loop instead of one launch per layer.
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ dG, const float* __restrict__ lattices, float* __restrict__ gW1,
                                                       int edge_in, float* __restrict__ gb1, int B, int H, size_t dg_stride = 0, size_t w_stride = 0) {
    __shared__ float red[8][10][32];
    dG += (size_t)blockIdx.y * dg_stride;
    gW1 += (size_t)blockIdx.y * w_stride;
    gb1 += (size_t)blockIdx.y * w_stride;
    const int fl = threadIdx.x & 31, bg = threadIdx.x >> 5, f = blockIdx.x * 32 + fl;
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.f;
    if (f < H)
        for (int b = bg; b < B; b += 8) {
            const float* Lm = lattices + (size_t)b * 9;
            float d = dG[(size_t)b * H + f];
            sb += d;
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                int r = m / 3, c = m % 3;
                acc[m] += d * (Lm[r * 3] * Lm[c * 3] + Lm[r * 3 + 1] * Lm[c * 3 + 1] + Lm[r * 3 + 2] * Lm[c * 3 + 2]);
            }
        }
#pragma unroll
    for (int m = 0; m < 9; ++m) red[bg][m][fl] = acc[m];
    red[bg][9][fl] = sb;
    __syncthreads();
    // 10 outputs x 32 features per block, one thread each
    for (int o = threadIdx.x; o < 320; o += 256) {
        const int m = o >> 5, ff = o & 31, fo = blockIdx.x * 32 + ff;
        if (fo >= H) continue;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][m][ff];
        if (m < 9) gW1[(size_t)fo * edge_in + 2 * H + m] += t;
        else gb1[fo] += t;
    }
}

// LayerNorm backward (one wave per row): dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w.
// dx is ADDED to `dx_acc` when `accumulate` (residual stream), else written.  Per-block partial sums
// of dw = sum dy*xhat and db = sum dy go to `part[blk][2H]` (reduced by tn_reduce_kernel).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats, const float* __restrict__ w,
                                                            float* __restrict__ dx, int accumulate, float* __restrict__ part, int N,
                                                            int H, int rows_per_block) {
    extern __shared__ float sm[];  // [4][2H]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = sm + (size_t)wave * 2 * H;
    for (int c = lane; c < 2 * H; c += 64) my[c] = 0.f;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += 4) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float g[8], xh[8];
        float s1 = 0.f, s2 = 0.f;
        int cnt = 0;
        for (int c = lane; c < H; c += 64) {
            float d = dy[(size_t)row * ld_dy + c];
            xh[cnt] = (x[(size_t)row * H + c] - mean) * rstd;
            g[cnt] = d * w[c];
            s1 += g[cnt];
            s2 += g[cnt] * xh[cnt];
            my[c] += d * xh[cnt];
            my[H + c] += d;
            ++cnt;
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        cnt = 0;
        for (int c = lane; c < H; c += 64) {
            float v = rstd * (g[cnt] - s1 - xh[cnt] * s2);
            size_t o = (size_t)row * H + c;
            dx[o] = accumulate ? dx[o] + v : v;
            ++cnt;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256)
        part[(size_t)blockIdx.x * 2 * H + c] = (sm[c] + sm[2 * H + c]) + (sm[4 * H + c] + sm[6 * H + c]);
}

// heads backward: dhf[i][f] = sum_a dT[i][a] Wt[a][f] + sum_c dX[i][c] Wc[c][f] + dgf[g(i)][f] / n_g
__global__ void heads_bwd_kernel(const float* __restrict__ dT, const float* __restrict__ dX, const float* __restrict__ dgf,
                                 const float* __restrict__ Wt, const float* __restrict__ Wc, const int* __restrict__ node2graph,
                                 const int* __restrict__ node_off, float* __restrict__ dhf, int N, int H) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * H) return;
    int i = (int)(idx / H), f = (int)(idx % H);
    float s = 0.f;
    const float* dt = dT + (size_t)i * MI_NUM_TYPES;
    for (int a = 0; a < MI_NUM_TYPES; ++a) s += dt[a] * Wt[(size_t)a * H + f];
#pragma unroll
    for (int c = 0; c < 3; ++c) s += dX[i * 3 + c] * Wc[(size_t)c * H + f];
    int g = node2graph[i];
    s += dgf[(size_t)g * H + f] / (float)(node_off[g + 1] - node_off[g]);
    dhf[idx] = s;
}

// lattice head backward (one block per crystal): out = reshape(lo,3,3) @ L  =>  dlo = dOut @ L^T;
// dgf[b] = Wl^T dlo;  dlo stored for the wgrad gWl[m][f] += sum_b dlo[b][m] * gf[b][f]
__global__ void lattice_head_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ lattices, const float* __restrict__ Wl,
                                        float* __restrict__ dlo_out, float* __restrict__ dgf, int H) {
    int b = blockIdx.x;
    __shared__ float dlo[9];
    if (threadIdx.x < 9) {
        int r = threadIdx.x / 3, c = threadIdx.x % 3;
        const float* Lm = lattices + (size_t)b * 9;
        const float* d = dOut + (size_t)b * 9;
        float v = d[r * 3] * Lm[c * 3] + d[r * 3 + 1] * Lm[c * 3 + 1] + d[r * 3 + 2] * Lm[c * 3 + 2];
        dlo[threadIdx.x] = v;
        dlo_out[(size_t)b * 12 + threadIdx.x] = v;
    }
    if (threadIdx.x >= 9 && threadIdx.x < 12) dlo_out[(size_t)b * 12 + threadIdx.x] = 0.f;
    __syncthreads();
    for (int f = threadIdx.x; f < H; f += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 9; ++m) s += dlo[m] * Wl[(size_t)m * H + f];
        dgf[(size_t)b * H + f] = s;
    }
}

// Adam (torch.optim.Adam defaults): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float eps, float gscale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * gscale;
    float mi_ = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi_;
    v[i] = vi;
    p[i] -= lr_over_bc1 * (mi_ / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}


#if MI_PLANES_FP16
// ------------------------------------------------------------------------------------------------------------------------------------
// C[Na x Kx] += A^T X over a long row list, both operands given as the PLANE SETS their producers wrote (A = dZ2, X = M1 of the layer: the
// weight gradient of edge_mlp.2, `loss.backward()` of pipeline/mat_invent.py:164).  gemm_tn_split_kernel reads fp32 rows and splits -- and, for
// M1 = silu(Z1), re-evaluates the activation of -- every operand element in each of the four column tiles that read it, transposing on the way
// into LDS: measured VALU-bound, 0.58 PF/s issued.  Here a 32-row slab of both operands goes to LDS by LDS-DMA exactly as it lies
// in memory ([column tile][plane][32 rows][32 columns], 2 KiB pieces), and the k-strided MFMA operands are gathered by the hardware's
// transposing read: ds_read_b64_tr_b16 hands lane i of a 16-lane group element (i & 3) of the 8-byte chunk that lane 4 j + (i >> 2) of the
// group addressed (j = 0..3, probed on the device: scripts/probes/tr_probe.hip) -- so with lane 4 j + c pointing at row k0 + j, columns m0 + 4 c ..,
// lane i receives A[k0 .. k0 + 3][m0 + i]: four consecutive k of its own column, two reads per 8-deep operand.  No VALU work on the operands.
// 256 (A columns) x 128 (X columns) output tile, eight waves (4 x 2, a wave owns 64 x 64), three 48 KiB stages two slabs ahead, one
// workgroup per CU; the row list is split as in gemm_tn_split_kernel (XCD-aware tile order, partial tiles summed in fixed order by tn_reduce).
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int TNP_STAGE = 49152, TNP_NST = 3;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_planes_kernel(Planes A, int a_col0, Planes X, int x_col0,
                                                                                                        float* __restrict__ P, int M, int rows_per_split, int gx,
                                                                                                        int gy, int nsplit, const float* __restrict__ sa,
                                                                                                        const float* __restrict__ sx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int id_ = blockIdx.x, slot_ = id_ >> 3, tile_ = slot_ % (gx * gy), bz = (slot_ / (gx * gy)) * 8 + (id_ & 7);
    if (bz >= nsplit) return;
    const int bx = tile_ % gx, by = tile_ / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kg = lane >> 5;
    const int m_begin = bz * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    const int nslab = (m_end - m_begin + 31) >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // LDS-DMA: a slab = 16 pieces of A (8 column tiles x 2 planes) + 8 of X, 2 KiB each = two 1 KiB instructions; a wave issues four of A's and two of X's
    const __amdgpu_buffer_rsrc_t rsa = uniform_rsrc(A.base, 0x7ffffff0), rsx = uniform_rsrc(X.base, 0x7ffffff0);
    const unsigned a_stride = ((unsigned)A.KT * 12288u + 2048u) * 2u, x_stride = ((unsigned)X.KT * 12288u + 2048u) * 2u;
    const int a_ct0 = (a_col0 + by * 256) >> 5, x_ct0 = (x_col0 + bx * 128) >> 5;
    auto issue = [&](int r0, int st) {
        const unsigned ra = (unsigned)(r0 >> 7) * a_stride + (unsigned)(r0 & 127) * 64u, rx = (unsigned)(r0 >> 7) * x_stride + (unsigned)(r0 & 127) * 64u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = wave * 4 + t, pc = q >> 1, half = q & 1;   // piece pc = column tile (pc >> 1), plane (pc & 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + st * TNP_STAGE + pc * 2048 + half * 1024), 16, lane * 16,
                                                     ra + (unsigned)(a_ct0 + (pc >> 1)) * 24576u + (unsigned)(pc & 1) * 8192u + (unsigned)half * 1024u, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = wave * 2 + t, pc = q >> 1, half = q & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(smem + st * TNP_STAGE + 32768 + pc * 2048 + half * 1024), 16,
                                                     lane * 16, rx + (unsigned)(x_ct0 + (pc >> 1)) * 24576u + (unsigned)(pc & 1) * 8192u + (unsigned)half * 1024u, 0, 0);
        }
    };
    // transposing reads: lane = 16 g + 4 j + c addresses row 8 (g >> 1) + j (+ 4 for the second half of an operand), columns 16 (g & 1) + 4 c of a 32 x 32 piece.
    // As inline asm with their own lgkmcnt waits: written as the compiler's builtin, every read was preceded by s_waitcnt vmcnt(0) -- the compiler
    // cannot tell that the LDS-DMA in flight targets another stage -- which serialised each slab behind the slab requested two ahead of it
    // (227 us for edge_mlp.2's gradient at 102 k edges); __syncthreads() likewise drains every counter, the bare s_barrier does not.
    const int g = lane >> 4, jj = (lane >> 2) & 3, cc = lane & 3;
    const unsigned lane_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)((8 * (g >> 1) + jj) * 64 + (16 * (g & 1) + 4 * cc) * 2);
    typedef unsigned long long u64;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define TNP_RD(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    issue(m_begin, 0);
    if (nslab > 1) issue(m_begin + 32, 1);
#pragma unroll 1
    for (int it = 0; it < nslab; ++it) {
        if (it + 1 < nslab) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // (this wave's six pieces of slab it + 1 may still be in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // slab `it` is in LDS for every wave; everyone is done with the stage slab it + 2 goes to (read in iteration it - 1)
        if (it + 2 < nslab) issue(m_begin + (it + 2) * 32, (it + 2) % TNP_NST);
        const unsigned ab = lane_base + (unsigned)(it % TNP_NST) * TNP_STAGE + (unsigned)(wm * 2) * 4096u;            // this wave's two A pieces (x 2 planes)
        const unsigned xb = lane_base + (unsigned)(it % TNP_NST) * TNP_STAGE + 32768u + (unsigned)(wn * 2) * 4096u;   // and its two X pieces
        u64 fa[2][8], fb[2][8];   // [k16 step][tile * 4 + plane * 2 + half]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            TNP_RD(fa[ks][0], ab, ks * 1024 + 0 * 2048 + 0);   TNP_RD(fa[ks][1], ab, ks * 1024 + 0 * 2048 + 256);
            TNP_RD(fa[ks][2], ab, ks * 1024 + 1 * 2048 + 0);   TNP_RD(fa[ks][3], ab, ks * 1024 + 1 * 2048 + 256);
            TNP_RD(fa[ks][4], ab, ks * 1024 + 2 * 2048 + 0);   TNP_RD(fa[ks][5], ab, ks * 1024 + 2 * 2048 + 256);
            TNP_RD(fa[ks][6], ab, ks * 1024 + 3 * 2048 + 0);   TNP_RD(fa[ks][7], ab, ks * 1024 + 3 * 2048 + 256);
            TNP_RD(fb[ks][0], xb, ks * 1024 + 0 * 2048 + 0);   TNP_RD(fb[ks][1], xb, ks * 1024 + 0 * 2048 + 256);
            TNP_RD(fb[ks][2], xb, ks * 1024 + 1 * 2048 + 0);   TNP_RD(fb[ks][3], xb, ks * 1024 + 1 * 2048 + 256);
            TNP_RD(fb[ks][4], xb, ks * 1024 + 2 * 2048 + 0);   TNP_RD(fb[ks][5], xb, ks * 1024 + 2 * 2048 + 256);
            TNP_RD(fb[ks][6], xb, ks * 1024 + 3 * 2048 + 0);   TNP_RD(fb[ks][7], xb, ks * 1024 + 3 * 2048 + 256);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // LDS reads return in order: with the second step's sixteen still in flight the first step's are complete.  The fragments ride through the
            // wait as read-write operands, so that no consumer can be scheduled in front of it.
            if (ks == 0)
                asm volatile("s_waitcnt lgkmcnt(15)\n s_nop 0" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[0][4]), "+v"(fa[0][5]), "+v"(fa[0][6]), "+v"(fa[0][7]),
                             "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fb[0][4]), "+v"(fb[0][5]), "+v"(fb[0][6]), "+v"(fb[0][7]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 0" : "+v"(acc[1][1]), "+v"(fa[1][0]),   // (behind the first step's last product: its reads overlap those MFMAs) "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fa[1][4]), "+v"(fa[1][5]), "+v"(fa[1][6]), "+v"(fa[1][7]),
                             "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]), "+v"(fb[1][4]), "+v"(fb[1][5]), "+v"(fb[1][6]), "+v"(fb[1][7]));
            f16x8 a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    a[i][pl] = __builtin_bit_cast(f16x8, u64x2{fa[ks][i * 4 + pl * 2], fa[ks][i * 4 + pl * 2 + 1]});
                    b[i][pl] = __builtin_bit_cast(f16x8, u64x2{fb[ks][i * 4 + pl * 2], fb[ks][i * 4 + pl * 2 + 1]});
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#if !MI_TF32_CLASS   // (the TF32-class build keeps the leading term only: 11-bit operands, f32 accumulate)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
#endif
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
    }
#undef TNP_RD
    const float sc_out = sa[1] * sx[1];
    const int PK = gx * 128;
    float* Pt = P + (size_t)bz * (gy * 256) * PK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = by * 256 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, k = bx * 128 + wn * 64 + j * 32 + l31;
                Pt[(size_t)n * PK + k] = acc[i][j][r] * sc_out;
            }
}

// whether the product can run there: whole 256 / 128-column tiles starting on 32-column boundaries, rows beyond M inside the operands' (zero) row padding
static bool gemm_tn_planes_ok(const Planes& A, int a_col0, const Planes& X, int x_col0, int64_t M, int Na, int Kx) {
    return A.base && X.base && (Na & 255) == 0 && (Kx & 127) == 0 && (a_col0 & 31) == 0 && (x_col0 & 31) == 0 && M >= 4096 &&
           (int64_t)planes_elems(M, (a_col0 + Na)) * 2 < 0x7ffffff0 && (int64_t)planes_elems(M, (x_col0 + Kx)) * 2 < 0x7ffffff0 &&
           A.KT * 32 >= a_col0 + Na && X.KT * 32 >= x_col0 + Kx;
}
extern int g_tn_target_tiles;
// C[Na, Kx] (ldc) += A[:, a_col0 : a_col0 + Na]^T X[:, x_col0 : x_col0 + Kx] / (sa sx); sa / sx = device-side {scale, 1 / scale} of the two plane sets
static int gemm_tn_planes(const Planes& A, int a_col0, const Planes& X, int x_col0, float* C, int ldc, int M, int Na, int Kx, const float* sa, const float* sx,
                          float* scratch, size_t scratch_floats, hipStream_t s) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)gemm_tn_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TNP_NST * TNP_STAGE); });
    MI_HIP(attr_err);
    const int gy = Na / 256, gx = Kx / 128;
    count_mfma(M, Na, Kx, MI_PLANES_TERMS);
    int nsplit = std::max(1, std::min(cdiv(M, 256), cdiv(tn_target_tiles() * 2 / 3, gx * gy)));   // (one workgroup per CU: two rounds of the chip -- of this launch's share of it)
    while (nsplit > 1 && (size_t)nsplit * Na * Kx > scratch_floats) --nsplit;
    MI_CHECK((size_t)nsplit * Na * Kx <= scratch_floats, MI_ENOMEM, "gemm_tn_planes scratch too small");
    const int rows = cdiv(cdiv(M, nsplit), 32) * 32;
    nsplit = cdiv(M, rows);
    hipLaunchKernelGGL(gemm_tn_planes_kernel, dim3(gx * gy * ((nsplit + 7) / 8 * 8)), dim3(512), TNP_NST * TNP_STAGE, s, A, a_col0, X, x_col0, scratch, M, rows, gx, gy,
                       nsplit, sa, sx);
    tn_reduce(scratch, nsplit, Na, Kx, C, ldc, Na, Kx, s);
    MI_KERNEL_CHECK();
    return MI_OK;
}
#endif

static int alloc_tape(mi_net* net, mi_batch* b) {
    if (b->tape.allocated) return MI_OK;
    const size_t N = b->N, B = b->B, E = (size_t)b->E_cap, H = net->H, L = net->L, F = net->F, TD = net->TD;
    Tape& t = b->tape;
    int rc = MI_OK;
#define T_(p, n) \
    if (rc == MI_OK) rc = dev_alloc(b, &t.p, (n))
    T_(cat, L * N * 2 * H);
    T_(Z1, L * E * H);
    T_(Z2, L * E * H);
    T_(Xpre, L * N * H);
    T_(Ypre, L * N * H);
    T_(lnstat, (L + 1) * N * 2);
    T_(dsc_layers, L * 8);
    T_(lnpart, L * (size_t)cdiv(N, 32) * 2 * H);
    T_(gf, B * H);
    T_(atom_types, N * MI_NUM_TYPES);
    T_(t_emb, B * TD);
    T_(lattices, B * 9);
    T_(frac, N * 3);
    T_(dh, N * H);
    T_(dY, N * H);
    T_(dXa, N * H);
    T_(Xa, N * H);
    T_(dcat, N * 2 * H);
    T_(dPQ, N * 2 * H);
    T_(dG, L * B * H);   // one [B][H] slot per layer: the lattice term's weight gradient of all layers is one launch behind the loop
    T_(dgf, B * H);
    T_(dlo, B * 12);
    T_(dtproj, B * H);
    T_(M1, E * H);
    if (MI_PLANES_FP16 && g_bwd_wgrad_planes && H % 256 == 0 && b->M1pl && E >= 8192) {   // every layer's M1 plane set (see Tape::M1pl_l); row padding zero
        t.m1pl_stride = planes_elems(E, H);
        T_(M1pl_l, L * t.m1pl_stride);
        if (rc == MI_OK && hipMemset(t.M1pl_l, 0, L * t.m1pl_stride * sizeof(unsigned short)) != hipSuccess) rc = MI_EHIP;
    }
    if (t.M1pl_l && b->Np >= 4096 && H % 256 == 0) {   // the pair differences / sums of dZ1 as plane sets (see Tape::DmPl)
        const size_t ne = planes_elems(b->Np, H);
        T_(DmPl, ne);
        T_(DpPl, ne);
        if (rc == MI_OK && (hipMemset(t.DmPl, 0, ne * sizeof(unsigned short)) != hipSuccess || hipMemset(t.DpPl, 0, ne * sizeof(unsigned short)) != hipSuccess)) rc = MI_EHIP;
    }
    T_(dM1, E * H);
    T_(FF, E * 6 * F);
    T_(nz_lat, B * 9);
    T_(nz_frac, N * 3);
    T_(nz_types, N * MI_NUM_TYPES);
    T_(tar_x, N * 3);
    T_(rnd_l, B * 9);
    T_(rnd_t, N * MI_NUM_TYPES);
    T_(d_l, B * 9);
    T_(d_x, N * 3);
    T_(d_t, N * MI_NUM_TYPES);
    T_(Lb, B);
    T_(KLb, B);
    t.scratch_floats = std::max<size_t>((size_t)1 << 22, 16 * 2 * H * std::max<size_t>(2 * H, 6 * F + 64) + 1024 * 2 * H);
    T_(scratch, t.scratch_floats);
#undef T_
    if (rc == MI_OK) t.allocated = true;
    return rc;
}

int net_tape_prepare(mi_net* net, mi_batch* b) { return alloc_tape(net, b); }

// Deferred node-level weight gradients (see Tape): `slots` micro-steps per contraction, 0 = every backward contracts its own rows.
int net_wgrad_window(mi_net* net, mi_batch* b, int slots) {
    MI_CHECK(slots >= 0 && slots <= 64, MI_EINVAL, "wgrad window: 0 .. 64 micro-steps");
    Tape& t = b->tape;
    MI_CHECK(t.wcur == 0, MI_ESTATE, "wgrad window changed while micro-steps are pending: mi_cspnet_wgrad_flush first");
    if (slots == 0 || b->N == 0) {
        t.wslots = 0;
        return MI_OK;
    }
    MI_TRY(alloc_tape(net, b));
    if (slots > t.wcap) {
        for (float** q : {&t.w_dY, &t.w_Xa, &t.w_dXa, &t.w_cat, &t.w_dPQ, &t.w_dlo, &t.w_gf, &t.w_dtype, &t.w_dcoord, &t.w_hf, &t.w_dh, &t.w_x1, &t.w_dtproj, &t.w_temb,
                          &t.w_eXa, &t.w_types})
            if (*q) {
                b->allocs.erase(std::remove(b->allocs.begin(), b->allocs.end(), (void*)*q), b->allocs.end());
                (void)hipFree(*q);
                *q = nullptr;
            }
        t.wcap = 0;
        const size_t rows = (size_t)net->L * slots * b->N, H = net->H;
        int rc = MI_OK;
        if (rc == MI_OK) rc = dev_alloc(b, &t.w_dY, rows * H);
        if (rc == MI_OK) rc = dev_alloc(b, &t.w_Xa, rows * H);
        if (rc == MI_OK) rc = dev_alloc(b, &t.w_dXa, rows * H);
        if (rc == MI_OK) rc = dev_alloc(b, &t.w_cat, rows * 2 * H);
        if (rc == MI_OK) rc = dev_alloc(b, &t.w_dPQ, rows * 2 * H);
        // the head / embedding operands of the same window (see Tape): [slots][N or B][width]
        const size_t sn = (size_t)slots * b->N, sb = (size_t)slots * b->B;
        if (g_bwd_head_window) {
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_dlo, sb * 12);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_gf, sb * H);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_dtype, sn * MI_NUM_TYPES);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_dcoord, sn * 3);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_hf, sn * H);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_dh, sn * H);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_x1, sn * H);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_dtproj, sb * H);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_temb, sb * net->TD);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_eXa, sn * H);
            if (rc == MI_OK) rc = dev_alloc(b, &t.w_types, sn * MI_NUM_TYPES);
        }
        if (rc != MI_OK) {
            t.wslots = 0;
            return rc;
        }
        t.wcap = slots;
    }
    t.wslots = slots;
    return MI_OK;
}

int net_wgrad_flush(mi_net* net, mi_batch* b, float* grad, hipStream_t s) {
    Tape& t = b->tape;
    if (t.wslots == 0 || t.wcur == 0) return MI_OK;
    const int H = net->H, L = net->L, M = t.wcur * b->N;
    auto G = [&](const std::string& name) { return grad + net->off(name); };
    float* sc = t.scratch;
    const size_t scf = t.scratch_floats;
    for (int l = 0; l < L; ++l) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        const size_t r0 = (size_t)l * t.wslots * b->N;
        const float *dY = t.w_dY + r0 * H, *Xa = t.w_Xa + r0 * H, *dXa = t.w_dXa + r0 * H, *cat = t.w_cat + r0 * 2 * H, *dPQ = t.w_dPQ + r0 * 2 * H;
        MI_TRY(gemm_tn_auto(dY, H, Xa, H, G(p + "node_mlp.2.weight"), H, M, H, H, sc, scf, s));
        MI_TRY(colsum_acc(dY, H, G(p + "node_mlp.2.bias"), M, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(dXa, H, cat, 2 * H, G(p + "node_mlp.0.weight"), 2 * H, M, H, 2 * H, sc, scf, s));
        MI_TRY(colsum_acc(dXa, H, G(p + "node_mlp.0.bias"), M, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(dPQ, 2 * H, cat, 2 * H, G(p + "edge_mlp.0.weight"), net->edge_in, M, H, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(dPQ + H, 2 * H, cat, 2 * H, G(p + "edge_mlp.0.weight") + H, net->edge_in, M, H, H, sc, scf, s));
    }
    if (t.head_window()) {   // heads and embedding over the same rows (net_backward skipped them micro-step by micro-step)
        const int MB = t.wcur * b->B, TD = net->TD, WA = H + TD;
        MI_TRY(gemm_tn_auto(t.w_dlo, 12, t.w_gf, H, G("lattice_out.weight"), H, MB, 9, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(t.w_dtype, MI_NUM_TYPES, t.w_hf, H, G("type_out.weight"), H, M, MI_NUM_TYPES, H, sc, scf, s));
        MI_TRY(colsum_acc(t.w_dtype, MI_NUM_TYPES, G("type_out.bias"), M, MI_NUM_TYPES, sc, scf, s));
        MI_TRY(gemm_tn_auto(t.w_dcoord, 3, t.w_hf, H, G("coord_out.weight"), H, M, 3, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(t.w_dh, H, t.w_x1, H, G("atom_latent_emb.weight"), WA, M, H, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(t.w_dtproj, H, t.w_temb, TD, G("atom_latent_emb.weight") + H, WA, MB, H, TD, sc, scf, s));
        MI_TRY(colsum_acc(t.w_dh, H, G("atom_latent_emb.bias"), M, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(t.w_eXa, H, t.w_types, MI_NUM_TYPES, G("node_embedding.weight"), MI_NUM_TYPES, M, H, MI_NUM_TYPES, sc, scf, s));
        MI_TRY(colsum_acc(t.w_eXa, H, G("node_embedding.bias"), M, H, sc, scf, s));
    }
    t.wcur = 0;
    return MI_OK;
}

static inline dim3 g1(int64_t n) { return dim3((unsigned)cdiv(n, 256)); }

int net_backward(mi_net* net, mi_batch* b, const float* d_lat, const float* d_coord, const float* d_type, float* grad, hipStream_t s) {
    MI_CHECK(b->tape.allocated && b->tape.valid, MI_ESTATE, "mi_cspnet_backward without a preceding training forward on this batch");
    const int H = net->H, L = net->L, N = b->N, B = b->B, TD = net->TD, F = net->F;
    const int64_t E = b->E;
    if (N == 0 || B == 0) return MI_OK;
    Tape& t = b->tape;
    const size_t NH = (size_t)N * H;
    auto G = [&](const std::string& name) { return grad + net->off(name); };
    float* sc = t.scratch;
    const size_t scf = t.scratch_floats;
    const bool defer = t.wslots > 0;
    MI_CHECK(!defer || t.wcur < t.wslots, MI_ESTATE, "wgrad window overrun");

    // Head / embedding weight gradients in the deferred window (Tape::w_hf ...): this micro-step's operand rows live in slot t.wcur -- the training forward wrote
    // x1 / hf / gf / the inputs there, this pass writes dlo / d h / dtproj / the embedding's dXa there -- and net_wgrad_flush contracts them with the rest
    const bool hw = defer && t.head_window();
    const size_t hs = hw ? (size_t)t.wcur : 0;
    float* const dlo = hw ? t.w_dlo + hs * B * 12 : t.dlo;
    float* const dh = hw ? t.w_dh + hs * NH : t.dh;
    // ---------------- heads ----------------
    hipLaunchKernelGGL(lattice_head_bwd_kernel, dim3(B), dim3(256), 0, s, d_lat, t.in_lat, net->p("lattice_out.weight"), dlo, t.dgf, H);
    MI_KERNEL_CHECK();
    hipLaunchKernelGGL(heads_bwd_kernel, g1(NH), dim3(256), 0, s, d_type, d_coord, t.dgf, net->p("type_out.weight"),
                       net->p("coord_out.weight"), b->node2graph, b->node_off, t.dY, N, H);  // dY = d hf
    MI_KERNEL_CHECK();
    if (hw) {   // (the fused micro-step seeds d_type / d_coord straight into the slot; any other caller's arrays are copied there)
        float* const dts = t.w_dtype + hs * N * MI_NUM_TYPES;
        float* const dcs = t.w_dcoord + hs * N * 3;
        if (d_type != dts) MI_HIP(hipMemcpyAsync(dts, d_type, (size_t)N * MI_NUM_TYPES * 4, hipMemcpyDeviceToDevice, s));
        if (d_coord != dcs) MI_HIP(hipMemcpyAsync(dcs, d_coord, (size_t)N * 3 * 4, hipMemcpyDeviceToDevice, s));
    } else {
        MI_TRY(gemm_tn_auto(dlo, 12, t.gf, H, G("lattice_out.weight"), H, B, 9, H, sc, scf, s));
        MI_TRY(gemm_tn_auto(d_type, MI_NUM_TYPES, b->hf, H, G("type_out.weight"), H, N, MI_NUM_TYPES, H, sc, scf, s));
        MI_TRY(colsum_acc(d_type, MI_NUM_TYPES, G("type_out.bias"), N, MI_NUM_TYPES, sc, scf, s));
        MI_TRY(gemm_tn_auto(d_coord, 3, b->hf, H, G("coord_out.weight"), H, N, 3, H, sc, scf, s));
    }

    auto ln_bwd = [&](const float* dy, int ld_dy, const float* x, const float* stats, const std::string& wname, float* dx, int accumulate) {
        const int rows_per_block = N >= 16384 ? 64 : (N >= 2048 ? 16 : 4), nblk = cdiv(N, rows_per_block);  // >= ~256 blocks when possible
        MI_CHECK((size_t)nblk * 2 * H <= scf, MI_ENOMEM, "LN scratch");
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), 8 * H * sizeof(float), s, dy, ld_dy, x, stats, net->p(wname + ".weight"),
                           dx, accumulate, sc, N, H, rows_per_block);
        // part[blk][0:H] -> dw, [H:2H] -> db ; weight and bias are adjacent in theta (weight first)
        hipLaunchKernelGGL(part_reduce_kernel<>, dim3(cdiv(2 * H, PART_REDUCE_COLS)), dim3(256), 0, s, sc, nblk, 2 * H, G(wname + ".weight"), 2 * H);
        MI_KERNEL_CHECK();
        return MI_OK;
    };
    // final LayerNorm: d h_L
    if (net->cfg.ln) {
        MI_TRY(ln_bwd(t.dY, H, b->h + (size_t)L * NH, t.lnstat + (size_t)L * N * 2, "final_layer_norm", dh, 0));
    } else {
        MI_HIP(hipMemcpyAsync(dh, t.dY, NH * 4, hipMemcpyDeviceToDevice, s));
    }

    // Fourier features are the same for every layer; in pair mode one row per unordered atom pair
    const bool pairs = g_edge_pairs && !b->knn && H % 4 == 0;
    const int64_t Np = b->Np;
    // (built by the first product that reads the fp32 rows: the plane-set form of the Fourier block's weight gradient -- the default at the benchmark size --
    //  reads the forward's pair-mode operand planes instead and never asks for them: one launch and 37 MB of writes fewer per micro-step)
    bool ff_built = false;
    auto ensure_ff = [&]() -> int {
        if (ff_built) return MI_OK;
        ff_built = true;
        if (pairs && Np > 0) hipLaunchKernelGGL(fourier_kernel, g1(Np * 3 * F), dim3(256), 0, s, t.in_frac, (const float*)nullptr, b->pair_i, b->pair_j, t.FF, Np, F);
        else if (!pairs && E > 0) hipLaunchKernelGGL(fourier_kernel, g1(E * 3 * F), dim3(256), 0, s, t.in_frac, b->fd, b->src, b->dst, t.FF, E, F);
        MI_KERNEL_CHECK();
        return MI_OK;
    };

    // The node-level part of the pass as ONE launch per layer boundary (node_bwd.hip): launch l runs what follows layer l's edge stage (the h_i / h_j
    // projections' data gradient, the LayerNorm gradient into the residual stream) and what precedes layer l - 1's (its node MLP's data gradients).
    const bool nb = net->cfg.ln && node_bwd_supported(net, b) && (size_t)cdiv(N, 32) * 2 * H <= scf;
    // the chain's per-workgroup partial sums of d ln_w | d ln_b: every layer's in a slot of its own (Tape::lnpart) -- ONE reduction launch for all layers behind the loop
    const size_t ln_slot = (size_t)cdiv(N, 32) * 2 * H;
    const bool ln_batched = nb && L >= 2 && t.lnpart != nullptr;
    float* const ln_tail = t.lnpart;
    const bool gram_batched = L >= 2 && net->off("csp_layer_1.edge_mlp.0.bias") - net->off("csp_layer_0.edge_mlp.0.bias") ==
                                            net->off("csp_layer_1.edge_mlp.0.weight") - net->off("csp_layer_0.edge_mlp.0.weight");
    // (the operand rows of layer `layer`'s node-level weight gradients: slot t.wcur of the deferred window, or the tape's single buffers)
    auto w_rows = [&](int layer, float** dYp, float** Xap, float** dXap) {
        const size_t r = defer ? (size_t)layer * t.wslots * N + (size_t)t.wcur * N : 0;
        *dYp = defer ? t.w_dY + r * H : t.dY;
        *Xap = defer ? t.w_Xa + r * H : t.Xa;
        *dXap = defer ? t.w_dXa + r * H : t.dXa;
    };
    if (nb) {
        // one pair of absmax slots per layer {max |d cat|, max |dM1|}, cleared once per pass (the seven-launch form clears its single pair per layer)
        MI_HIP(hipMemsetAsync(b->absmax + 2 * L + 2, 0, 2 * L * sizeof(unsigned), s));
        float *dY1, *Xa1, *dXa1;
        w_rows(L - 1, &dY1, &Xa1, &dXa1);
        MI_TRY(node_bwd(net, b, L, nullptr, dh, dY1, dXa1, Xa1, nullptr, b->absmax + 2 * L + 2 + 2 * (L - 1), s));
    }

    // ---------------- layers, last to first ----------------
    for (int l = L - 1; l >= 0; --l) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        // (deferred node-level weight gradients: this micro-step's operand rows live in slot t.wcur of the window)
        const size_t wr = defer ? (size_t)l * t.wslots * N + (size_t)t.wcur * N : 0;
        const float* cat = defer ? t.w_cat + wr * 2 * H : t.cat + (size_t)l * N * 2 * H;
        float* dYl = defer ? t.w_dY + wr * H : t.dY;
        float* Xa = defer ? t.w_Xa + wr * H : t.Xa;
        float* dXa = defer ? t.w_dXa + wr * H : t.dXa;
        float* dPQ = defer ? t.w_dPQ + wr * 2 * H : t.dPQ;
        unsigned* const am = nb ? b->absmax + 2 * L + 2 + 2 * l : b->absmax + 2 * L;   // this layer's {max |d cat|, max |dM1|}
        float* const dGl = t.dG + (size_t)l * B * H;   // this layer's d G (per crystal): kept until the batched lattice-term weight gradient behind the loop
        float* Z1 = t.Z1 + (size_t)l * E * H;
        float* Z2 = t.Z2 + (size_t)l * E * H;
        const float* Xpre = t.Xpre + (size_t)l * NH;
        const float* Ypre = t.Ypre + (size_t)l * NH;
        // node MLP (cspnet.py:80-82)
        if (nb) {   // (dY, Xa, dXa of this layer and d cat were written by the chain launch above this layer's edge stage)
            if (!defer) {
                MI_TRY(gemm_tn_auto(dYl, H, Xa, H, G(p + "node_mlp.2.weight"), H, N, H, H, sc, scf, s));
                MI_TRY(colsum_acc(dYl, H, G(p + "node_mlp.2.bias"), N, H, sc, scf, s));
                MI_TRY(gemm_tn_auto(dXa, H, cat, 2 * H, G(p + "node_mlp.0.weight"), 2 * H, N, H, 2 * H, sc, scf, s));
                MI_TRY(colsum_acc(dXa, H, G(p + "node_mlp.0.bias"), N, H, sc, scf, s));
            }
        } else {
        hipLaunchKernelGGL(silu_bwd_kernel, g1(NH), dim3(256), 0, s, dh, Ypre, dYl, (int64_t)NH);
        hipLaunchKernelGGL(silu_fwd_kernel, g1(NH), dim3(256), 0, s, Xpre, Xa, (int64_t)NH);
        MI_KERNEL_CHECK();
        if (!defer) {
            MI_TRY(gemm_tn_auto(dYl, H, Xa, H, G(p + "node_mlp.2.weight"), H, N, H, H, sc, scf, s));
            MI_TRY(colsum_acc(dYl, H, G(p + "node_mlp.2.bias"), N, H, sc, scf, s));
        }
        MI_TRY(gemm_nt(dYl, H, net->Wn2T + l * (size_t)H * H, H, dXa, H, N, H, H, GemmEpilogue(), s, &b->sk));
        // (silu' as this product's epilogue measured 2.5 % slower end to end than the separate vectorised pass: in the MFMA result
        // layout a lane owns one column of 16 rows, so the pre-activation comes in as 16 four-byte loads per tile)
        hipLaunchKernelGGL(silu_bwd_kernel, g1(NH), dim3(256), 0, s, dXa, Xpre, dXa, (int64_t)NH);
        MI_KERNEL_CHECK();
        if (!defer) {
            MI_TRY(gemm_tn_auto(dXa, H, cat, 2 * H, G(p + "node_mlp.0.weight"), 2 * H, N, H, 2 * H, sc, scf, s));
            MI_TRY(colsum_acc(dXa, H, G(p + "node_mlp.0.bias"), N, H, sc, scf, s));
        }
        MI_TRY(gemm_nt(dXa, H, net->Wn1T + l * (size_t)2 * H * H, H, t.dcat, 2 * H, N, 2 * H, H, GemmEpilogue(), s, &b->sk));
        }
        // edge stage (cspnet.py:59-79)
        bool edge_sums_done = false;  // dPQ and dG already produced by the fused pair-mode kernel
        if (E > 0) {
            // (64-row chunks when the scratch holds their partial sums: 4x the workgroups of 256-row chunks, 182 -> 88 us)
            const int crows = (size_t)cdiv(E, 64) * H <= scf ? 64 : 256;
            const int nchunk = (int)cdiv(E, crows);
            const bool dz2_sums = (size_t)nchunk * H <= scf && H % 4 == 0;  // Z2 := dZ2, with edge_mlp.2.bias's gradient (column sums) on the way
            // fp16 plane format: dZ2 is also written as a plane set (scale from max |d cat|), and its data gradient runs on the
            // pre-split plane GEMM (three fp16 terms) instead of the on-the-fly three-plane bf16 split (six terms)
            // (short edge lists are bound by the launch rate: the two extra launches cost more than the products save -- 18 crystals,
            // 2.7k edges: 3.55 -> 3.75 ms per unstacked micro-step; the stacked form, 32k edges, gains 2 %)
            const bool dz2_planes = MI_PLANES_FP16 && g_bwd_dz2_planes && dz2_sums && g_gemm_mode == MI_GEMM_SPLIT && net->W2Tpl && b->M1pl && H % 32 == 0 &&
                                    E >= 8192;
            Planes dzp;
            if (dz2_planes) {
                dzp = make_planes(b->M1pl, H, 1.f, b->dsc + 6);
                if (!nb) {   // (the fused chain raised max |d cat| in the epilogue of the product that wrote it; its slots were cleared once per pass)
                    MI_HIP(hipMemsetAsync(am, 0, 2 * sizeof(unsigned), s));
                    hipLaunchKernelGGL(absmax_bwd_kernel, dim3(std::min<int64_t>(256, cdiv((int64_t)N * 2 * H, 1024))), dim3(256), 0, s, t.dcat, (int64_t)N * 2 * H, am);
                }
            }
            // (both consumers of dZ2 on its plane set -- the data gradient below and edge_mlp.2's weight gradient: its fp32 rows are dead)
            bool w2_planes = false;
#if MI_PLANES_FP16
            const Planes m1l = t.M1pl_l ? make_planes(t.M1pl_l + (size_t)l * t.m1pl_stride, H, 1.f, t.dsc_layers + (size_t)l * 8) : Planes();
            w2_planes = dz2_planes && t.dsc_layers_valid && g_bwd_wgrad_f16 && g_bwd_wgrad_planes && t.M1pl_l && g_tn_xsilu && gemm_tn_planes_ok(dzp, 0, m1l, 0, E, H, H);
#endif
            if (dz2_sums) {
                hipLaunchKernelGGL(edge_dz2_colsum_kernel, dim3(1, nchunk), dim3(256), 0, s, t.dcat, b->src, b->rowptr, Z2, sc, E, H, crows, dzp,
                                   am, b->dsc + 6, w2_planes ? 0 : 1);
                hipLaunchKernelGGL(part_reduce_kernel<>, dim3(cdiv(H, PART_REDUCE_COLS)), dim3(256), 0, s, sc, nchunk, H, G(p + "edge_mlp.2.bias"), H);
            } else {
                hipLaunchKernelGGL(edge_dz2_kernel, g1(E * H), dim3(256), 0, s, t.dcat, b->src, b->rowptr, Z2, E, H);
            }
#if MI_PLANES_FP16
            if (w2_planes) {   // (both operands exist as plane sets: dZ2 just written above, M1 kept by this layer's training forward)
                MI_TRY(gemm_tn_planes(dzp, 0, m1l, 0, G(p + "edge_mlp.2.weight"), H, (int)E, H, H, b->dsc + 6, t.dsc_layers + (size_t)l * 8, sc, scf, s));
            } else
#endif
            if (g_tn_xsilu && gemm_tn_is_split(Z2, H, Z1, H, (int)E, H, H)) {  // M1 = silu(Z1) formed inside the product's operand load
                // (fp16 plane format: dZ2's scale was published by the dZ2 kernel, M1's by this layer's training forward)
                const bool w2_f16 = dz2_planes && t.dsc_layers_valid && g_bwd_wgrad_f16;
                MI_TRY(gemm_tn_auto(Z2, H, Z1, H, G(p + "edge_mlp.2.weight"), H, (int)E, H, H, sc, scf, s, true, w2_f16 ? b->dsc + 6 : nullptr,
                                    w2_f16 ? t.dsc_layers + (size_t)l * 8 : nullptr));
            } else {
                hipLaunchKernelGGL(silu_fwd_kernel, g1(E * H), dim3(256), 0, s, Z1, t.M1, E * H);
                MI_KERNEL_CHECK();
                MI_TRY(gemm_tn_auto(Z2, H, t.M1, H, G(p + "edge_mlp.2.weight"), H, (int)E, H, H, sc, scf, s));
            }
            if (!dz2_sums) MI_TRY(colsum_acc(Z2, H, G(p + "edge_mlp.2.bias"), (int)E, H, sc, scf, s));
            if (dz2_planes) {
                PlanesEpilogue pd;
                pd.C = t.dM1;
                pd.ldc = H;
                pd.absmax = am + 1;  // max |dM1|: bounds the pair-mode weight gradient's operands
                Planes w2t = make_planes(net->W2Tpl + (size_t)l * planes_elems(H, H), H);
                if (net->W2Tf) w2t.frag = net->W2Tf + (size_t)l * frag_elems(H, H);   // (from 16384 edges up: the 128 x 256 register-tile kernel)
                MI_TRY(gemm_planes(dzp, w2t, (int)E, H, H, pd, s));
            } else {
                MI_TRY(gemm_nt(Z2, H, net->W2T + l * (size_t)H * H, H, t.dM1, H, (int)E, H, H, GemmEpilogue(), s));
            }
            // fc pair mode: one pass over the crystal blocks of dM1 / Z1 yields every consumer of dZ1 (see edge_bwd_pairs_kernel)
            const bool fused_pairs = pairs && g_bwd_pairs_fused && b->nmax_fc <= 64 && (size_t)B * H <= scf - H;
            if (!fused_pairs) {
                hipLaunchKernelGGL(silu_bwd_kernel, g1(E * H), dim3(256), 0, s, t.dM1, Z1, t.dM1, E * H);  // dM1 := dZ1
                MI_KERNEL_CHECK();
            }
            if (pairs) {  // t.M1 is free again (its weight gradient is done): Dm | Dp live there
                float* gWff = G(p + "edge_mlp.0.weight") + 2 * H + 9;
                float *Dm = t.M1, *Dp = t.M1 + (size_t)Np * H;
                float* dsum = sc + scf - H;  // the tail of the scratch: the reductions below use its head
                // (the LDS-tile pair pass: its per-crystal partial sums of the self edges' dZ1 go into the cosine block in ONE launch -- no dsum, no memset)
                const bool dsum_folded = fused_pairs && b->nmax_fc <= PAIRS_NMAX && g_bwd_pairs_tile;
                if (!dsum_folded) MI_HIP(hipMemsetAsync(dsum, 0, H * sizeof(float), s));
                const bool wff_f16 = dz2_planes && fused_pairs && g_bwd_wgrad_f16;  // two-plane fp16 operands for the Fourier-block weight gradient
                bool wff_planes = false;   // ... and from plane sets: the pass below writes Dm / Dp as such, the Fourier operand is the forward's pair-mode plane set
#if MI_PLANES_FP16
                Planes dmp, dpp, ffp;
                if (wff_f16 && g_bwd_wgrad_planes && t.DmPl && b->nmax_fc <= PAIRS_NMAX && g_bwd_pairs_tile && g_edge_pairs && b->FFpl) {
                    dmp = make_planes(t.DmPl, H, 1.f, b->dsc + 8);
                    dpp = make_planes(t.DpPl, H, 1.f, b->dsc + 8);
                    ffp = make_planes(b->FFpl, 2 * net->Kh, PL_S_UNIT);
                    wff_planes = gemm_tn_planes_ok(dmp, 0, ffp, 0, Np, H, 3 * F) && gemm_tn_planes_ok(dpp, 0, ffp, net->Kh, Np, H, 3 * F);
                }
#endif
                if (fused_pairs && b->nmax_fc <= PAIRS_NMAX && g_bwd_pairs_tile) {
                    const size_t sh = ((size_t)b->nmax_fc * b->nmax_fc + b->nmax_fc) * PAIRS_W * sizeof(float);
#if MI_PLANES_FP16
                    if (wff_planes)
                        hipLaunchKernelGGL(edge_bwd_pairs_tile_kernel, dim3(B, cdiv(H, PAIRS_W)), dim3(256), sh, s, t.dM1, Z1, b->node_off, b->rowptr, b->pair_off, Dm,
                                           Dp, dPQ, dGl, sc, H, am + 1, b->dsc + 8, dmp, dpp);
                    else
#endif
                    hipLaunchKernelGGL(edge_bwd_pairs_tile_kernel, dim3(B, cdiv(H, PAIRS_W)), dim3(256), sh, s, t.dM1, Z1, b->node_off, b->rowptr, b->pair_off, Dm,
                                       Dp, dPQ, dGl, sc, H, wff_f16 ? am + 1 : nullptr, wff_f16 ? b->dsc + 8 : nullptr);
                    hipLaunchKernelGGL(part_reduce_bcast_kernel, dim3(cdiv(H, 32), 12), dim3(256), 0, s, sc, B, H, gWff + 3 * F, net->edge_in, H, 3 * F);
                    MI_KERNEL_CHECK();
                } else if (fused_pairs) {
                    hipLaunchKernelGGL(edge_bwd_pairs_kernel, dim3(B, cdiv(H, 128)), dim3(128), (size_t)2 * b->nmax_fc * 128 * sizeof(float), s, t.dM1,
                                       Z1, b->node_off, b->rowptr, b->pair_off, Dm, Dp, dPQ, dGl, sc, H,
                                       wff_f16 ? am + 1 : nullptr, wff_f16 ? b->dsc + 8 : nullptr);
                    hipLaunchKernelGGL(part_reduce_kernel<>, dim3(cdiv(H, PART_REDUCE_COLS)), dim3(256), 0, s, sc, B, H, dsum, H);
                    MI_KERNEL_CHECK();
                } else {
                    if (Np > 0) {
                        hipLaunchKernelGGL(pair_combine_kernel, g1(Np * (H / 4)), dim3(256), 0, s, t.dM1, b->pair_e1, b->pair_e2, Dm, Dp, Np, H);
                        MI_KERNEL_CHECK();
                    }
                    MI_TRY(colsum_acc(t.dM1, H, dsum, N, H, sc, scf - H, s, b->e_diag));
                }
                if (!dsum_folded) hipLaunchKernelGGL(row_broadcast_add_kernel, g1((int64_t)H * 3 * F), dim3(256), 0, s, dsum, gWff + 3 * F, net->edge_in, H, 3 * F);
                MI_KERNEL_CHECK();
#if MI_PLANES_FP16
                if (Np > 0 && wff_planes) {
                    MI_TRY(gemm_tn_planes(dmp, 0, ffp, 0, gWff, net->edge_in, (int)Np, H, 3 * F, b->dsc + 8, b->dsc + 12, sc, scf - H, s));
                    MI_TRY(gemm_tn_planes(dpp, 0, ffp, net->Kh, gWff + 3 * F, net->edge_in, (int)Np, H, 3 * F, b->dsc + 8, b->dsc + 12, sc, scf - H, s));
                } else
#endif
                if (Np > 0) {
                    MI_TRY(ensure_ff());
                    MI_TRY(gemm_tn_auto(Dm, H, t.FF, 6 * F, gWff, net->edge_in, (int)Np, H, 3 * F, sc, scf - H, s, false, wff_f16 ? b->dsc + 8 : nullptr,
                                        wff_f16 ? b->dsc + 10 : nullptr));
                    MI_TRY(gemm_tn_auto(Dp, H, t.FF + 3 * F, 6 * F, gWff + 3 * F, net->edge_in, (int)Np, H, 3 * F, sc, scf - H, s, false,
                                        wff_f16 ? b->dsc + 8 : nullptr, wff_f16 ? b->dsc + 10 : nullptr));
                }
            } else {
                MI_TRY(ensure_ff());
                MI_TRY(gemm_tn_auto(t.dM1, H, t.FF, 6 * F, G(p + "edge_mlp.0.weight") + 2 * H + 9, net->edge_in, (int)E, H, 6 * F, sc, scf, s));
            }
            if (!fused_pairs) {
                if (b->knn) hipLaunchKernelGGL(edge_dpq_csr_kernel, g1(NH), dim3(256), 0, s, t.dM1, b->rowptr, b->inedge, dPQ, N, H);
                else hipLaunchKernelGGL(edge_dpq_kernel, g1(NH), dim3(256), 0, s, t.dM1, b->rowptr, b->node2graph, b->node_off, dPQ, N, H);
                MI_KERNEL_CHECK();
            }
            edge_sums_done = fused_pairs;
        } else {
            MI_HIP(hipMemsetAsync(dPQ, 0, NH * 2 * 4, s));
        }
        if (!edge_sums_done) hipLaunchKernelGGL(graph_sum_kernel, g1((int64_t)B * H), dim3(256), 0, s, dPQ, 2 * H, b->node_off, dGl, B, H);
        if (!gram_batched) hipLaunchKernelGGL(gram_bwd_kernel, dim3(cdiv(H, 32)), dim3(256), 0, s, dGl, t.in_lat, G(p + "edge_mlp.0.weight"), net->edge_in,
                                              G(p + "edge_mlp.0.bias"), B, H, (size_t)0, (size_t)0);
        MI_KERNEL_CHECK();
        if (!defer) {
            MI_TRY(gemm_tn_auto(dPQ, 2 * H, cat, 2 * H, G(p + "edge_mlp.0.weight"), net->edge_in, N, H, H, sc, scf, s));
            MI_TRY(gemm_tn_auto(dPQ + H, 2 * H, cat, 2 * H, G(p + "edge_mlp.0.weight") + H, net->edge_in, N, H, H, sc, scf, s));
        }
        if (nb) {
            // d hn = dcat[:, :H] + dPQ Whh, dh_l = dh_{l+1} + LN'(d hn) -- and the node MLP's data gradients of the layer below, in the same launch
            float *dY1 = nullptr, *Xa1 = nullptr, *dXa1 = nullptr;
            if (l > 0) w_rows(l - 1, &dY1, &Xa1, &dXa1);
            MI_TRY(node_bwd(net, b, l, dPQ, dh, dY1, dXa1, Xa1, ln_batched ? ln_tail + (size_t)l * ln_slot : sc, l > 0 ? b->absmax + 2 * L + 2 + 2 * (l - 1) : nullptr, s));
            if (!ln_batched) hipLaunchKernelGGL(part_reduce_kernel<>, dim3(cdiv(2 * H, PART_REDUCE_COLS)), dim3(256), 0, s, sc, cdiv(N, 32), 2 * H, G(p + "layer_norm.weight"), 2 * H);
            MI_KERNEL_CHECK();
            continue;
        }
        // d hn = dcat[:, :H] + dPQ * Whh   -> dY (reuse)
        GemmEpilogue er;
        er.residual = t.dcat;
        er.ld_res = 2 * H;
        MI_TRY(gemm_nt(dPQ, 2 * H, net->WhhT + l * (size_t)2 * H * H, 2 * H, t.dY, H, N, H, 2 * H, er, s, &b->sk));
        // LayerNorm + residual stream: dh_l = dh_{l+1} + LN'(d hn)
        if (net->cfg.ln) {
            MI_TRY(ln_bwd(t.dY, H, b->h + (size_t)l * NH, t.lnstat + (size_t)l * N * 2, p + "layer_norm", dh, 1));
        } else {
            hipLaunchKernelGGL(axpy_kernel, g1(NH), dim3(256), 0, s, t.dY, dh, (int64_t)NH);
            MI_KERNEL_CHECK();
        }
    }

    if (gram_batched) {   // the lattice term's weight / bias gradient of every layer: one launch (gridDim.y = L) over the per-layer d G slots
        const size_t stride = net->off("csp_layer_1.edge_mlp.0.weight") - net->off("csp_layer_0.edge_mlp.0.weight");
        hipLaunchKernelGGL(gram_bwd_kernel, dim3(cdiv(H, 32), L), dim3(256), 0, s, t.dG, t.in_lat, G("csp_layer_0.edge_mlp.0.weight"), net->edge_in,
                           G("csp_layer_0.edge_mlp.0.bias"), B, H, (size_t)B * H, stride);
        MI_KERNEL_CHECK();
    }
    if (ln_batched) {   // (every layer's parameter block has the same size: a constant stride between the layers' LayerNorm weights)
        const size_t stride = net->off("csp_layer_1.layer_norm.weight") - net->off("csp_layer_0.layer_norm.weight");
        hipLaunchKernelGGL(part_reduce_batched_kernel, dim3(cdiv(2 * H, 32), L), dim3(256), 0, s, ln_tail, ln_slot, cdiv(N, 32), 2 * H,
                           G("csp_layer_0.layer_norm.weight"), stride, 2 * H);
        MI_KERNEL_CHECK();
    }
    // ---------------- embedding (cspnet.py:265-271) ----------------
    const int WA = H + TD;
    if (hw) {   // the data path only: d tproj and the embedding's dXa into the slot; the five contractions and three column sums run in net_wgrad_flush
        hipLaunchKernelGGL(graph_sum_kernel, g1((int64_t)B * H), dim3(256), 0, s, dh, H, b->node_off, t.w_dtproj + hs * B * H, B, H);
        MI_KERNEL_CHECK();
        MI_TRY(gemm_nt(dh, H, net->WaT, H, t.w_eXa + hs * NH, H, N, H, H, GemmEpilogue(), s, &b->sk));
    } else {
    MI_TRY(gemm_tn_auto(dh, H, b->x1, H, G("atom_latent_emb.weight"), WA, N, H, H, sc, scf, s));
    hipLaunchKernelGGL(graph_sum_kernel, g1((int64_t)B * H), dim3(256), 0, s, dh, H, b->node_off, t.dtproj, B, H);
    MI_KERNEL_CHECK();
    MI_TRY(gemm_tn_auto(t.dtproj, H, t.in_temb, TD, G("atom_latent_emb.weight") + H, WA, B, H, TD, sc, scf, s));
    MI_TRY(colsum_acc(dh, H, G("atom_latent_emb.bias"), N, H, sc, scf, s));
    MI_TRY(gemm_nt(dh, H, net->WaT, H, t.dXa, H, N, H, H, GemmEpilogue(), s, &b->sk));
    MI_TRY(gemm_tn_auto(t.dXa, H, t.in_types, MI_NUM_TYPES, G("node_embedding.weight"), MI_NUM_TYPES, N, H, MI_NUM_TYPES, sc, scf, s));
    MI_TRY(colsum_acc(t.dXa, H, G("node_embedding.bias"), N, H, sc, scf, s));
    }
    if (defer && ++t.wcur == t.wslots) MI_TRY(net_wgrad_flush(net, b, grad, s));
    return MI_OK;
}

// ---- fused per-sample loss, prior-anchor penalty, reward weighting and gradient seeds (K14) -------------
// DiffCSPModule.calc_sample_loss (diffusion.py:126-136), calc_kl_reg (:143-148) and the reward weighting of
// MatInvent.ft_step (mat_invent.py:158-163) in one kernel, one block per crystal:
//   L_b  = cl*mean9((pl-rl)^2) + cx*mean_i mean3((px-tx)^2) + ct*mean_i mean100((pt-rt)^2)
//   KL_b = mean9((pl-plp)^2) + mean_i mean3((px-pxp)^2) + mean_i mean100((pt-ptp)^2)
//   total = inv_denom * sum_b ( r_b L_b + sigma (1.1 - r_b) KL_b )          (inv_denom = 1/(B_global*accum))
// and writes d total / d(pl, px, pt) -- the upstream gradients of the network backward.
struct LossArgs {
    const float *pl, *px, *pt, *plp, *pxp, *ptp, *rl, *tx, *rt, *reward;
    const int* node_off;
    float *dl, *dx, *dt, *Lb, *KLb;
    float cl, cx, ct, sigma, inv_denom;
};
__global__ __launch_bounds__(256) void ft_loss_kernel(LossArgs a) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n0 = a.node_off[b], n1 = a.node_off[b + 1], n = n1 - n0;
    const float r = a.reward[b], w1 = r * a.inv_denom, w2 = a.sigma * (1.1f - r) * a.inv_denom;
    const float inv_n = n > 0 ? 1.0f / (float)n : 0.f;
    float sl = 0.f, skl = 0.f;
    if (tid < 9) {
        const int i = b * 9 + tid;
        const float e = a.pl[i] - a.rl[i], k = a.pl[i] - a.plp[i];
        sl = e * e;
        skl = k * k;
        a.dl[i] = (w1 * a.cl * 2.0f * e + w2 * 2.0f * k) / 9.0f;
    }
    float sx = 0.f, skx = 0.f;
    for (int i = n0 * 3 + tid; i < n1 * 3; i += 256) {
        const float e = a.px[i] - a.tx[i], k = a.px[i] - a.pxp[i];
        sx += e * e;
        skx += k * k;
        a.dx[i] = (w1 * a.cx * 2.0f * e + w2 * 2.0f * k) * inv_n / 3.0f;
    }
    float st = 0.f, skt = 0.f;
    for (int64_t i = (int64_t)n0 * MI_NUM_TYPES + tid; i < (int64_t)n1 * MI_NUM_TYPES; i += 256) {
        const float e = a.pt[i] - a.rt[i], k = a.pt[i] - a.ptp[i];
        st += e * e;
        skt += k * k;
        a.dt[i] = (w1 * a.ct * 2.0f * e + w2 * 2.0f * k) * inv_n / (float)MI_NUM_TYPES;
    }
    auto bsum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    sl = bsum(sl); skl = bsum(skl); sx = bsum(sx); skx = bsum(skx); st = bsum(st); skt = bsum(skt);
    if (tid == 0) {
        a.Lb[b] = a.cl * sl / 9.0f + a.cx * sx * inv_n / 3.0f + a.ct * st * inv_n / (float)MI_NUM_TYPES;
        a.KLb[b] = skl / 9.0f + skx * inv_n / 3.0f + skt * inv_n / (float)MI_NUM_TYPES;
    }
}
// stats[0] += accum-normalised loss, stats[1] += sum_b r_b L_b, stats[2] += sum_b (1.1 - r_b) KL_b  (mat_invent.py:168-170)
__global__ void ft_stats_kernel(const float* __restrict__ Lb, const float* __restrict__ KLb, const float* __restrict__ reward, int B,
                                float sigma, float inv_bglobal, float* __restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float d = 0.f, k = 0.f;
    for (int b = 0; b < B; ++b) {
        d += reward[b] * Lb[b];
        k += (1.1f - reward[b]) * KLb[b];
    }
    stats[0] += (d + sigma * k) * inv_bglobal;
    stats[1] += d;
    stats[2] += k;
}

int net_pack_transposes(mi_net* n, hipStream_t s) {
    const int H = n->H, L = n->L;
    if (!n->W2T) {
        MI_HIP(hipMalloc((void**)&n->W2T, (size_t)L * H * H * 4));
        MI_HIP(hipMalloc((void**)&n->Wn2T, (size_t)L * H * H * 4));
        MI_HIP(hipMalloc((void**)&n->Wn1T, (size_t)L * 2 * H * H * 4));
        MI_HIP(hipMalloc((void**)&n->WhhT, (size_t)L * 2 * H * H * 4));
        MI_HIP(hipMalloc((void**)&n->WaT, (size_t)H * H * 4));
        if (MI_PLANES_FP16 && H % 32 == 0) MI_HIP(hipMalloc((void**)&n->W2Tpl, (size_t)L * planes_elems(H, H) * sizeof(u16)));
        if (MI_PLANES_FP16 && H % 256 == 0) MI_HIP(hipMalloc((void**)&n->W2Tf, (size_t)L * frag_elems(H, H) * sizeof(u16)));
        if (MI_PLANES_FP16 && n->cfg.ln && (H == 128 || H == 256 || H == 512)) MI_HIP(hipMalloc((void**)&n->Wbw, (size_t)L * node_bwd_pack_elems(H) * sizeof(u16)));
    }
    for (int l = 0; l < L; ++l) {
        const std::string p = "csp_layer_" + std::to_string(l) + ".";
        hipLaunchKernelGGL(transpose_kernel, g1(H * H), dim3(256), 0, s, n->p(p + "edge_mlp.2.weight"), H, H, H, n->W2T + (size_t)l * H * H);
        if (n->W2Tpl) {
            Planes wp = make_planes(n->W2Tpl + (size_t)l * planes_elems(H, H), H);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)((H + 127) / 128 * 128) * wp.KT * 16, 256)), dim3(256), 0, s,
                               n->W2T + (size_t)l * H * H, H, H, H, wp, 0);
            if (n->W2Tf) MI_TRY(pack_frag_from_planes(wp, H, H, n->W2Tf + (size_t)l * frag_elems(H, H), s));
        }
        hipLaunchKernelGGL(transpose_kernel, g1(H * H), dim3(256), 0, s, n->p(p + "node_mlp.2.weight"), H, H, H, n->Wn2T + (size_t)l * H * H);
        hipLaunchKernelGGL(transpose_kernel, g1(2 * H * H), dim3(256), 0, s, n->p(p + "node_mlp.0.weight"), 2 * H, H, 2 * H,
                           n->Wn1T + (size_t)l * 2 * H * H);
        hipLaunchKernelGGL(transpose_kernel, g1(2 * H * H), dim3(256), 0, s, n->Whh + l * n->whh_stride(), H, 2 * H, H,
                           n->WhhT + (size_t)l * 2 * H * H);
        if (n->Wbw) MI_TRY(node_bwd_pack(n, l, n->p(p + "edge_mlp.0.weight"), n->p(p + "node_mlp.0.weight"), n->p(p + "node_mlp.2.weight"), s));
    }
    hipLaunchKernelGGL(transpose_kernel, g1(H * H), dim3(256), 0, s, n->p("atom_latent_emb.weight"), H + n->TD, H, H, n->WaT);
    MI_KERNEL_CHECK();
    return MI_OK;
}

// ---- forward noising (diffusion.py:81-119) --------------------------------------------------------
// lattice_params_to_matrix_torch (models/diffcsp/utils.py:68-96)
__device__ __forceinline__ void lattice_matrix(const float* len, const float* ang, float* M) {
    const float d2r = 0.017453292519943295f;
    float c0 = cosf(ang[0] * d2r), c1 = cosf(ang[1] * d2r), c2 = cosf(ang[2] * d2r);
    float s0 = sinf(ang[0] * d2r), s1 = sinf(ang[1] * d2r);
    float val = (c0 * c1 - c2) / (s0 * s1);
    val = fminf(1.f, fmaxf(-1.f, val));
    float gs = acosf(val);
    M[0] = len[0] * s1; M[1] = 0.f; M[2] = len[0] * c1;
    M[3] = -len[1] * s0 * cosf(gs); M[4] = len[1] * s0 * sinf(gs); M[5] = len[1] * c0;
    M[6] = 0.f; M[7] = 0.f; M[8] = len[2];
}

// d_log_p_wrapped_normal(x, sigma) (scheduler.py:39-43), 21 images
__device__ __forceinline__ float d_log_p_wn(float x, float sigma) {
    float num = 0.f, den = 0.f;
    const float s2 = sigma * sigma;
#pragma unroll
    for (int i = -10; i <= 10; ++i) {
        float v = x + (float)i;
        float e = expf(-(v * v) / 2.0f / s2);
        num += v / s2 * e;
        den += e;
    }
    return num / den;
}

struct NoiseArgs {
    const float *lengths, *angles, *frac0;
    const int* atom_types;            // [N] 1..100
    const float *rand_l, *rand_x, *rand_t;  // injected noise or NULL (Philox)
    const int* node_off;
    float *in_lat, *in_frac, *in_types, *tar_x;
    float *out_rand_l, *out_rand_t;   // the targets rand_l / rand_t as used (copies of the noise)
    float c0, c1, sigma, sigma_norm;
    uint64_t seed;
    uint32_t step;
    int64_t node_offset, graph_offset;
    // stacked timesteps (mi_ft_micro_steps_stacked): the batch holds `copies` replicas of a set of stack_B0 crystals / stack_N0
    // atoms, replica c noised for its own timestep -- schedule values sched[c], noise call `step + c`, and the counter-based draws
    // indexed by the ORIGINAL crystal / atom ids, so every replica sees exactly the noise of its own unstacked micro-step
    int stack_B0 = 0, stack_N0 = 0;
    float sc0[MI_MAX_STACK], sc1[MI_MAX_STACK], ssig[MI_MAX_STACK], ssn[MI_MAX_STACK];
    // per-crystal timesteps (add_noise without an explicit time, diffusion.py:83-84): sched[b] = {c0, c1, sigma, sigma_norm}
    const float* sched = nullptr;
};

__global__ __launch_bounds__(256) void add_noise_kernel(NoiseArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cp = a.stack_B0 ? b / a.stack_B0 : 0;  // replica (stacked timesteps), 0 otherwise
    const float c0 = a.sched ? a.sched[b * 4] : a.stack_B0 ? a.sc0[cp] : a.c0, c1 = a.sched ? a.sched[b * 4 + 1] : a.stack_B0 ? a.sc1[cp] : a.c1;
    const float sigma = a.sched ? a.sched[b * 4 + 2] : a.stack_B0 ? a.ssig[cp] : a.sigma;
    const float sigma_norm = a.sched ? a.sched[b * 4 + 3] : a.stack_B0 ? a.ssn[cp] : a.sigma_norm;
    const uint32_t step = a.step + (uint32_t)cp;
    const int64_t gsub = (int64_t)cp * a.stack_B0, nsub = (int64_t)cp * a.stack_N0;  // replica's first crystal / atom in the batch
    __shared__ float Lm[9];
    if (tid == 0) lattice_matrix(a.lengths + b * 3, a.angles + b * 3, Lm);
    __syncthreads();
    if (tid < 9) {
        int idx = b * 9 + tid;
        float z = a.rand_l ? a.rand_l[idx] : philox_normal1(a.seed, step, DRAW_FT_L, (uint64_t)(a.graph_offset - gsub) * 9 + idx);
        a.out_rand_l[idx] = z;
        a.in_lat[idx] = c0 * Lm[tid] + c1 * z;
    }
    const int n0 = a.node_off[b], n1 = a.node_off[b + 1];
    for (int idx = n0 * 3 + tid; idx < n1 * 3; idx += 256) {
        float z = a.rand_x ? a.rand_x[idx] : philox_normal1(a.seed, step, DRAW_FT_X, (uint64_t)(a.node_offset - nsub) * 3 + idx);
        float sx = sigma * z;
        a.in_frac[idx] = pymod1(a.frac0[idx] + sx);
        a.tar_x[idx] = d_log_p_wn(sx, sigma) / sqrtf(sigma_norm);
    }
    for (int64_t idx = (int64_t)n0 * MI_NUM_TYPES + tid; idx < (int64_t)n1 * MI_NUM_TYPES; idx += 256) {
        int i = (int)(idx / MI_NUM_TYPES), k = (int)(idx % MI_NUM_TYPES);
        float z = a.rand_t ? a.rand_t[idx] : philox_normal1(a.seed, step, DRAW_FT_T, (uint64_t)(a.node_offset - nsub) * MI_NUM_TYPES + idx);
        float onehot = (a.atom_types[i] - 1 == k) ? 1.f : 0.f;
        a.out_rand_t[idx] = z;
        a.in_types[idx] = c0 * onehot + c1 * z;
    }
}

// times[b] = t[b / B0]  (stacked timesteps)
struct StackTimes {
    int t[MI_MAX_STACK];
};
__global__ void fill_stack_times_kernel(int* __restrict__ times, StackTimes st, int B0, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) times[b] = st.t[b / B0];
}


}  // namespace mi

using namespace mi;

extern "C" {

int mi_cspnet_forward_train(mi_net* net, mi_batch* b, const float* t_emb, const float* atom_types, const float* frac,
                            const float* lattices, float* lattice_out, float* coord_out, float* type_out, void* stream) {
    MI_CHECK(net && b, MI_EINVAL, "null handle");
    MI_CHECK(b->H == net->H && b->L == net->L, MI_EINVAL, "batch was created for a different network");
    MI_TRY(net_tape_prepare(net, b));
    return net_forward(net, b, t_emb, atom_types, frac, lattices, lattice_out, coord_out, type_out, (hipStream_t)stream, true);
}

int mi_cspnet_backward(mi_net* net, mi_batch* b, const float* d_lattice_out, const float* d_coord_out, const float* d_type_out,
                       float* grad_theta, void* stream) {
    MI_CHECK(net && b && d_lattice_out && d_coord_out && d_type_out && grad_theta, MI_EINVAL, "null argument");
    MI_CHECK(net->W2T != nullptr, MI_ESTATE, "mi_net_set_params must run before backward");
    return net_backward(net, b, d_lattice_out, d_coord_out, d_type_out, grad_theta, (hipStream_t)stream);
}

int mi_batch_set_wgrad_window(mi_net* net, mi_batch* b, int micro_steps) {
    MI_CHECK(net && b, MI_EINVAL, "null handle");
    MI_CHECK(b->H == net->H && b->L == net->L, MI_EINVAL, "batch was created for a different network");
    return net_wgrad_window(net, b, micro_steps);
}

int mi_cspnet_wgrad_flush(mi_net* net, mi_batch* b, float* grad_theta, void* stream) {
    MI_CHECK(net && b && grad_theta, MI_EINVAL, "null argument");
    return net_wgrad_flush(net, b, grad_theta, (hipStream_t)stream);
}

int mi_batch_wgrad_pending(const mi_batch* b) { return b ? b->tape.wcur : 0; }

int mi_adam_step(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1,
                 float beta2, float eps, float grad_scale, void* stream) {
    MI_CHECK(theta && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, MI_EINVAL, "bad argument");
    if (n == 0) return MI_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, theta, grad, exp_avg, exp_avg_sq, n,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, grad_scale);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_add_noise(mi_batch* b, const float* lengths, const float* angles, const float* frac0, const int* atom_types, float c0, float c1,
                 float sigma, float sigma_norm, uint64_t seed, uint32_t step, const float* rand_l, const float* rand_x, const float* rand_t,
                 float* in_lattice, float* in_frac, float* in_types, float* tar_x, float* out_rand_l, float* out_rand_t, void* stream) {
    MI_CHECK(b && lengths && angles && frac0 && atom_types && in_lattice && in_frac && in_types && tar_x && out_rand_l && out_rand_t,
             MI_EINVAL, "null argument");
    if (b->B == 0) return MI_OK;
    NoiseArgs a{lengths, angles, frac0, atom_types, rand_l, rand_x, rand_t, b->node_off, in_lattice, in_frac, in_types, tar_x,
                out_rand_l, out_rand_t, c0, c1, sigma, sigma_norm, seed, step, b->node_offset, b->graph_offset};
    hipLaunchKernelGGL(add_noise_kernel, dim3(b->B), dim3(256), 0, (hipStream_t)stream, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_add_noise_per_crystal(mi_batch* b, const float* lengths, const float* angles, const float* frac0, const int* atom_types,
                             const float* sched, uint64_t seed, uint32_t step, const float* rand_l, const float* rand_x, const float* rand_t,
                             float* in_lattice, float* in_frac, float* in_types, float* tar_x, float* out_rand_l, float* out_rand_t,
                             void* stream) {
    MI_CHECK(b && lengths && angles && frac0 && atom_types && sched && in_lattice && in_frac && in_types && tar_x && out_rand_l && out_rand_t,
             MI_EINVAL, "null argument");
    if (b->B == 0) return MI_OK;
    NoiseArgs a{lengths, angles, frac0, atom_types, rand_l, rand_x, rand_t, b->node_off, in_lattice, in_frac, in_types, tar_x,
                out_rand_l, out_rand_t, 0.f, 0.f, 0.f, 0.f, seed, step, b->node_offset, b->graph_offset};
    a.sched = sched;
    hipLaunchKernelGGL(add_noise_kernel, dim3(b->B), dim3(256), 0, (hipStream_t)stream, a);
    MI_KERNEL_CHECK();
    return MI_OK;
}

// One fine-tune micro-step; `copies` > 0: stacked timesteps (see mi_ft_micro_steps_stacked), the schedule arrays then hold one value
// per replica and t / c0 / c1 / sigma_t / sigma_norm are ignored.
static int ft_micro_impl(mi_net* agent, mi_batch* ab, mi_net* prior, mi_batch* pb, const float* lengths, const float* angles,
                         const float* frac0, const int* atom_types, const float* reward, const float* time_freqs, int t, float c0, float c1,
                         float sigma_t, float sigma_norm, int copies, const int* ts, const float* c0s, const float* c1s, const float* sigmas,
                         const float* sigma_norms, uint64_t seed, uint32_t noise_step, const float* rand_l, const float* rand_x,
                         const float* rand_t, float cost_lattice, float cost_coord, float cost_type, float kl_sigma, int b_global,
                         int accum_steps, float* grad_theta, float* stats, float* out_sample_loss, float* out_kl, void* stream,
                         void* aux_stream) {
    MI_CHECK(agent && ab && prior && pb && lengths && angles && frac0 && atom_types && reward && time_freqs && grad_theta, MI_EINVAL,
             "null argument");
    TraceRange range("mi_ft_micro_step");
    MI_CHECK(ab != pb, MI_EINVAL, "agent and prior need separate batch handles (separate workspace)");
    MI_CHECK(ab->B == pb->B && ab->N == pb->N && b_global >= ab->B / (copies > 0 ? copies : 1) && accum_steps >= 1, MI_EINVAL,
             "inconsistent batch arguments");
    hipStream_t s = (hipStream_t)stream;
    const int B = ab->B, N = ab->N;
    if (B == 0 || N == 0) return MI_OK;
    MI_TRY(net_tape_prepare(agent, ab));
    Tape& tp = ab->tape;
    // with the head / embedding weight gradients deferred over a window (Tape::w_hf ...), what their contractions read of this micro-step is written straight
    // into the window's slot: the noised types, the time embedding, the gradient seeds of the type and coordinate heads
    const bool hw = tp.head_window();
    const size_t hs = hw ? (size_t)tp.wcur : 0;
    float* const nz_types = hw ? tp.w_types + hs * N * MI_NUM_TYPES : tp.nz_types;
    float* const temb = hw ? tp.w_temb + hs * B * agent->TD : ab->temb;
    float* const d_t = hw ? tp.w_dtype + hs * N * MI_NUM_TYPES : tp.d_t;
    float* const d_x = hw ? tp.w_dcoord + hs * N * 3 : tp.d_x;
    NoiseArgs na{lengths, angles, frac0, atom_types, rand_l, rand_x, rand_t, ab->node_off, tp.nz_lat, tp.nz_frac, nz_types, tp.tar_x,
                 tp.rnd_l, tp.rnd_t, c0, c1, sigma_t, sigma_norm, seed, noise_step, ab->node_offset, ab->graph_offset};
    if (copies > 0) {
        MI_CHECK(copies <= MI_MAX_STACK && B % copies == 0 && N % copies == 0 && ts && c0s && c1s && sigmas && sigma_norms, MI_EINVAL,
                 "stacked micro-step: %d replicas do not fit the batch (B = %d, N = %d) or the limit %d", copies, B, N, MI_MAX_STACK);
        StackTimes st;
        na.stack_B0 = B / copies;
        na.stack_N0 = N / copies;
        for (int c = 0; c < copies; ++c) {
            st.t[c] = ts[c];
            na.sc0[c] = c0s[c];
            na.sc1[c] = c1s[c];
            na.ssig[c] = sigmas[c];
            na.ssn[c] = sigma_norms[c];
        }
        hipLaunchKernelGGL(fill_stack_times_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, ab->times, st, na.stack_B0, B);
    } else {
        hipLaunchKernelGGL(fill_int_kernel2, dim3(cdiv(B, 256)), dim3(256), 0, s, ab->times, t, B);
    }
    MI_TRY(mi_time_embedding(ab->times, time_freqs, B, agent->TD, temb, stream));
    hipLaunchKernelGGL(add_noise_kernel, dim3(B), dim3(256), 0, s, na);
    MI_KERNEL_CHECK();
    if (aux_stream && aux_stream != stream) {
        // the frozen prior's forward is independent of the agent's: fork it onto the auxiliary stream (small fine-tune sets leave
        // most of the chip idle, so the two forwards overlap), join before the loss kernel reads both predictions
        hipStream_t s2 = (hipStream_t)aux_stream;
        if (!pb->ev_fork) {
            MI_HIP(hipEventCreateWithFlags(&pb->ev_fork, hipEventDisableTiming));
            MI_HIP(hipEventCreateWithFlags(&pb->ev_join, hipEventDisableTiming));
        }
        MI_HIP(hipEventRecord(pb->ev_fork, s));
        MI_HIP(hipStreamWaitEvent(s2, pb->ev_fork, 0));
        MI_TRY(net_forward(prior, pb, temb, nz_types, tp.nz_frac, tp.nz_lat, pb->pred_l, pb->pred_x, pb->pred_t, s2, false));
        MI_HIP(hipEventRecord(pb->ev_join, s2));
        tp.borrow_inputs = true;   // (the noised inputs live in this tape and the time embedding in this batch until the backward below has run)
        const int rc = net_forward(agent, ab, temb, nz_types, tp.nz_frac, tp.nz_lat, ab->pred_l, ab->pred_x, ab->pred_t, s, true);
        tp.borrow_inputs = false;
        MI_TRY(rc);
        MI_HIP(hipStreamWaitEvent(s, pb->ev_join, 0));
    } else {
        tp.borrow_inputs = true;
        const int rc = net_forward(agent, ab, temb, nz_types, tp.nz_frac, tp.nz_lat, ab->pred_l, ab->pred_x, ab->pred_t, s, true);
        tp.borrow_inputs = false;
        MI_TRY(rc);
        MI_TRY(net_forward(prior, pb, temb, nz_types, tp.nz_frac, tp.nz_lat, pb->pred_l, pb->pred_x, pb->pred_t, s, false));
    }
    LossArgs la{ab->pred_l, ab->pred_x, ab->pred_t, pb->pred_l, pb->pred_x, pb->pred_t, tp.rnd_l, tp.tar_x, tp.rnd_t, reward, ab->node_off,
                tp.d_l, d_x, d_t, tp.Lb, tp.KLb, cost_lattice, cost_coord, cost_type, kl_sigma,
                1.0f / ((float)b_global * (float)accum_steps)};
    hipLaunchKernelGGL(ft_loss_kernel, dim3(B), dim3(256), 0, s, la);
    if (stats) hipLaunchKernelGGL(ft_stats_kernel, dim3(1), dim3(64), 0, s, tp.Lb, tp.KLb, reward, B, kl_sigma, 1.0f / (float)b_global, stats);
    MI_KERNEL_CHECK();
    if (out_sample_loss) MI_HIP(hipMemcpyAsync(out_sample_loss, tp.Lb, B * 4, hipMemcpyDeviceToDevice, s));
    if (out_kl) MI_HIP(hipMemcpyAsync(out_kl, tp.KLb, B * 4, hipMemcpyDeviceToDevice, s));
    return net_backward(agent, ab, tp.d_l, d_x, d_t, grad_theta, s);
}

int mi_ft_micro_step(mi_net* agent, mi_batch* ab, mi_net* prior, mi_batch* pb, const float* lengths, const float* angles,
                     const float* frac0, const int* atom_types, const float* reward, const float* time_freqs, int t, float c0, float c1,
                     float sigma_t, float sigma_norm, uint64_t seed, uint32_t noise_step, const float* rand_l, const float* rand_x,
                     const float* rand_t, float cost_lattice, float cost_coord, float cost_type, float kl_sigma, int b_global, int accum_steps,
                     float* grad_theta, float* stats, float* out_sample_loss, float* out_kl, void* stream, void* aux_stream) {
    return ft_micro_impl(agent, ab, prior, pb, lengths, angles, frac0, atom_types, reward, time_freqs, t, c0, c1, sigma_t, sigma_norm, 0, nullptr,
                         nullptr, nullptr, nullptr, nullptr, seed, noise_step, rand_l, rand_x, rand_t, cost_lattice, cost_coord, cost_type,
                         kl_sigma, b_global, accum_steps, grad_theta, stats, out_sample_loss, out_kl, stream, aux_stream);
}

int mi_ft_micro_steps_stacked(mi_net* agent, mi_batch* ab, mi_net* prior, mi_batch* pb, const float* lengths, const float* angles,
                              const float* frac0, const int* atom_types, const float* reward, const float* time_freqs, int copies,
                              const int* t_host, const float* c0_host, const float* c1_host, const float* sigma_t_host,
                              const float* sigma_norm_host, uint64_t seed, uint32_t noise_step, const float* rand_l, const float* rand_x,
                              const float* rand_t, float cost_lattice, float cost_coord, float cost_type, float kl_sigma, int b_global,
                              int accum_steps, float* grad_theta, float* stats, void* stream, void* aux_stream) {
    MI_CHECK(copies >= 1, MI_EINVAL, "copies must be positive");
    return ft_micro_impl(agent, ab, prior, pb, lengths, angles, frac0, atom_types, reward, time_freqs, 0, 0.f, 0.f, 0.f, 0.f, copies, t_host,
                         c0_host, c1_host, sigma_t_host, sigma_norm_host, seed, noise_step, rand_l, rand_x, rand_t, cost_lattice, cost_coord,
                         cost_type, kl_sigma, b_global, accum_steps, grad_theta, stats, nullptr, nullptr, stream, aux_stream);
}

int mi_debug_gemm(int kind, const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, void* stream) {
    MI_CHECK(A && W && C, MI_EINVAL, "null argument");
    if (kind == 2 || kind == 3) {  // split both operands into tile-blocked bf16 planes (cached scratch), then the plane GEMM (3: 256-row variant)
        static u16 *pa = nullptr, *pw = nullptr;
        static size_t na = 0, nw = 0;
        hipStream_t s = (hipStream_t)stream;
        if (planes_elems(M, K) > na) { if (pa) (void)hipFree(pa); na = planes_elems(M, K); MI_HIP(hipMalloc((void**)&pa, na * 2)); }
        if (planes_elems(N, K) > nw) { if (pw) (void)hipFree(pw); nw = planes_elems(N, K); MI_HIP(hipMalloc((void**)&pw, nw * 2)); }
        Planes PA = make_planes(pa, K, PL_S_LN), PW = make_planes(pw, K, PL_SW);
        if (ldc >= 0) {
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)((M + 127) / 128 * 128) * PA.KT * 16, 256)), dim3(256), 0, s, A, lda, M, K, PA);
            hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)((N + 127) / 128 * 128) * PW.KT * 16, 256)), dim3(256), 0, s, W, ldw, N, K, PW);
        }
        if (MI_PLANES_FP16 && (N & 255) == 0 && (K & 63) == 0 && K >= 128) {   // + the fragment-order copy the register-tile form reads (mi_debug_set_planes_rt)
            static u16* pf = nullptr;
            static size_t nf = 0;
            if (frag_elems(N, K) > nf) { if (pf) (void)hipFree(pf); nf = frag_elems(N, K); MI_HIP(hipMalloc((void**)&pf, nf * 2)); }
            MI_TRY(pack_frag_from_planes(PW, N, K, pf, s));
            PW.frag = pf;
        }
        PlanesEpilogue pe;
        pe.C = C;
        pe.ldc = ldc < 0 ? -ldc : ldc;
        const int saved = g_planes_variant;
        g_planes_variant = kind == 3;
        const int rc = gemm_planes(PA, PW, M, N, K, pe, s);
        g_planes_variant = saved;
        return rc;
    }
#if MI_PLANES_FP16
    if (kind == 5) {  // the same product from PLANE SETS (gemm_tn_planes_kernel): both operands split here with scale 2^6 (|x| < 1023), M % 256 == 0, N % 128 == 0
        hipStream_t s = (hipStream_t)stream;
        MI_CHECK(M % 256 == 0 && N % 128 == 0 && K >= 4096, MI_EINVAL, "kind 5: the plane-set weight-gradient form needs M % 256 == 0, N % 128 == 0, K >= 4096");
        static u16 *pa = nullptr, *pw = nullptr;
        static float *sc = nullptr, *dsc = nullptr;
        static size_t na = 0, nw = 0;
        const size_t ea = planes_elems(K, M), ew = planes_elems(K, N), scf = (size_t)1 << 25;
        if (ea > na) { if (pa) (void)hipFree(pa); na = ea; MI_HIP(hipMalloc((void**)&pa, na * 2)); }
        if (ew > nw) { if (pw) (void)hipFree(pw); nw = ew; MI_HIP(hipMalloc((void**)&pw, nw * 2)); }
        if (!sc) MI_HIP(hipMalloc((void**)&sc, scf * sizeof(float)));
        if (!dsc) {
            MI_HIP(hipMalloc((void**)&dsc, 2 * sizeof(float)));
            const float h[2] = {64.f, 1.f / 64.f};
            MI_HIP(hipMemcpy(dsc, h, sizeof(h), hipMemcpyHostToDevice));
        }
        Planes PA = make_planes(pa, M, 64.f), PW = make_planes(pw, N, 64.f);
        hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)((K + 127) / 128 * 128) * PA.KT * 16, 256)), dim3(256), 0, s, A, lda, K, M, PA);
        hipLaunchKernelGGL(split_planes_kernel<>, dim3(cdiv((int64_t)((K + 127) / 128 * 128) * PW.KT * 16, 256)), dim3(256), 0, s, W, ldw, K, N, PW);
        MI_KERNEL_CHECK();
        MI_CHECK(gemm_tn_planes_ok(PA, 0, PW, 0, K, M, N), MI_EINVAL, "kind 5: shape outside the plane-set form");
        return gemm_tn_planes(PA, 0, PW, 0, C, ldc, K, M, N, dsc, dsc, sc, scf, s);
    }
#endif
    if (kind == 4) {  // weight-gradient form: C[M][N] += A^T W with A [K][M], W [K][N] (contraction over rows)
        static float* sc = nullptr;
        const size_t scf = (size_t)1 << 24;
        if (!sc) MI_HIP(hipMalloc((void**)&sc, scf * sizeof(float)));
        return gemm_tn_auto(A, lda, W, ldw, C, ldc, K, M, N, sc, scf, (hipStream_t)stream);
    }
    return kind == 0 ? gemm_nt_f32(A, lda, W, ldw, C, ldc, M, N, K, GemmEpilogue(), (hipStream_t)stream)
                     : gemm_nt_split(A, lda, W, ldw, C, ldc, M, N, K, GemmEpilogue(), (hipStream_t)stream);
}

}  // extern "C"

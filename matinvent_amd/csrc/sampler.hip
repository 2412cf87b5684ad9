// Reverse-diffusion sampler on gfx950: DiffCSPModule.sample (models/diffcsp/diffusion.py:273-399).
// Two score-network evaluations per step (corrector, predictor), state updates and the per-step
// log-probabilities, with no host synchronisation inside the chain.
//
// The update arithmetic mirrors the reference's separately-rounded fp32 tensor ops, so
// contraction into FMAs is disabled for this translation unit.
#pragma clang fp contract(off)

#include <string.h>

#include "net.h"

namespace mi {

// SinusoidalTimeEmbeddings.forward (diffusion.py:59-66): out[b] = [sin(t*f_k) | cos(t*f_k)]
__global__ void time_embedding_kernel(const int* __restrict__ times, const float* __restrict__ freqs, float* __restrict__ out,
                                      int B, int TD, int t_all = 0) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * TD) return;
    int b = idx / TD, k = idx % TD, half = TD / 2;
    float arg = (float)(times ? times[b] : t_all) * freqs[k < half ? k : k - half];   // times == NULL: every crystal at time t_all
    out[idx] = k < half ? sinf(arg) : cosf(arg);
}

// busy-wait for ~`cycles` shader clocks on one wave (stream-concurrency probe, mi_debug_spin)
__global__ void spin_kernel(long long cycles, int* sink) {
    const long long t0 = __builtin_readcyclecounter();
    int k = 0;
    while (__builtin_readcyclecounter() - t0 < cycles) ++k;
    if (sink && k == -1) *sink = k;
}

__global__ void fill_int_kernel(int* p, int v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void philox_fill_kernel(uint64_t seed, uint32_t step, uint32_t draw, int64_t off, int64_t n, int uniform,
                                   float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t e = (uint64_t)(off + i);
    out[i] = uniform ? philox_uniform1(seed, step, draw, e) : philox_normal1(seed, step, draw, e);
}

struct StepCoef {
    float c0, c1, sigma, sqrt_sn, step_corr, std_corr, step_pred, std_pred, std_corr_sq, std_pred_sq, sigma_sq, log_sigma;
};
__device__ __forceinline__ StepCoef load_coef(const float* coef, int t) {
    const float* c = coef + (size_t)t * MI_NCOEF;
    return StepCoef{c[MI_C_C0], c[MI_C_C1], c[MI_C_SIGMA], c[MI_C_SQRT_SN], c[MI_C_STEP_CORR], c[MI_C_STD_CORR],
                    c[MI_C_STEP_PRED], c[MI_C_STD_PRED], c[MI_C_STD_CORR_SQ], c[MI_C_STD_PRED_SQ], c[MI_C_SIGMA_SQ],
                    c[MI_C_LOG_SIGMA]};
}

// log_prob_wn (diffusion.py:25-29): log sum_{i=-10..10} exp(-(x - mu + i)^2 / 2 / sigma^2)
__device__ __forceinline__ float log_prob_wn(float x, float mu, float sigma_sq) {
    float p = 0.f;
    float d = x - mu;
#pragma unroll
    for (int i = -10; i <= 10; ++i) {
        float v = d + (float)i;
        p += expf(-(v * v) / 2.0f / sigma_sq);
    }
    return logf(p);
}
// torch.distributions.Normal(mu, sigma).log_prob(v)
__device__ __forceinline__ float normal_log_prob(float v, float mu, float var, float log_sigma) {
    float d = v - mu;
    return -(d * d) / (2.0f * var) - log_sigma - 0.91893853320467274178f;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- corrector (diffusion.py:320-334): x_{t-1/2} = x_t - eps * s + sqrt(2 eps) z ---------------
// one 64-lane block per crystal; also the corrector half of log_prob_x (:363-368)
__global__ __launch_bounds__(64) void corrector_kernel(const float* __restrict__ x_t, const float* __restrict__ pred_x,
                                                       const float* __restrict__ noise_x, const float* __restrict__ coef, int t,
                                                       uint64_t seed, int64_t node_offset, const int* __restrict__ node_off,
                                                       float* __restrict__ x_mid, float* __restrict__ lp_corr,
                                                       float* __restrict__ rec_mid, int keep_coords) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const StepCoef c = load_coef(coef, t);
    const int n0 = node_off[b], n1 = node_off[b + 1];
    float lp = 0.f;
    for (int idx = n0 * 3 + lane; idx < n1 * 3; idx += 64) {
        float z = 0.f;
        if (t > 1) z = noise_x ? noise_x[idx] : philox_normal1(seed, (uint32_t)t, DRAW_CORR_X, (uint64_t)node_offset * 3 + idx);
        float px = pred_x[idx] * c.sqrt_sn;
        float drift = x_t[idx] - c.step_corr * px;
        float xm = keep_coords ? x_t[idx] : drift + c.std_corr * z;  // CSP mode (:330): the coordinates are given and stay
        x_mid[idx] = xm;
        float xmw = pymod1(xm);
        if (rec_mid && t > 1) rec_mid[idx] = xmw;
        if (lp_corr && t > 1) lp += log_prob_wn(xmw, pymod1(drift), c.std_corr_sq);   // (lp_corr = NULL: nothing records log-probabilities -- 21 exponentials per coordinate not spent)
    }
    if (!lp_corr) return;
    lp = wave_sum(lp);
    // mean over the 3 coordinates, then mean over the atoms of the crystal
    if (lane == 0) lp_corr[b] = (n1 > n0) ? (lp / 3.0f) / (float)(n1 - n0) : 0.f;
}

// ---- predictor (diffusion.py:337-382) -------------------------------------------------------
struct PredictorArgs {
    const float *x_mid, *pred_x, *pred_l, *pred_t;
    const float *noise_x, *noise_l, *noise_t;  // slices for step t, or NULL (Philox)
    const float* coef;
    const int* node_off;
    const float* lp_corr;
    float *frac, *lattices, *atom_types;  // state, updated in place
    float *rec_types, *rec_frac, *rec_lat, *rec_lpl, *rec_lpt, *rec_lpx;  // slices (t-1 for state, t for log-probs)
    uint64_t seed;
    int64_t node_offset, graph_offset;
    int t;
    int keep_lattice, keep_coords;  // CSP mode (diffusion.py:283-287, 308-312, 348-349): that part of the state is never moved
};

__global__ __launch_bounds__(256) void predictor_kernel(PredictorArgs a) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = a.t;
    const StepCoef c = load_coef(a.coef, t);
    const int n0 = a.node_off[b], n1 = a.node_off[b + 1], n = n1 - n0;
    const float cnt = (float)(n > 0 ? n : 1);
    // the three log-probabilities of the step (diffusion.py:357-368) exist for a recording caller only: without one their arithmetic (21 exponentials per
    // coordinate, a logarithm per logit) and block reductions are skipped -- the state update is the same instructions either way
    const bool want_lp = t > 1 && (a.rec_lpl != nullptr || a.rec_lpt != nullptr || a.rec_lpx != nullptr);

    // lattice: l_{t-1} = c0 (l_t - c1 pred_l) + sigma z
    float lp_l = 0.f;
    if (tid < 9) {
        int idx = b * 9 + tid;
        float z = 0.f;
        if (t > 1) z = a.noise_l ? a.noise_l[idx] : philox_normal1(a.seed, (uint32_t)t, DRAW_PRED_L, (uint64_t)a.graph_offset * 9 + idx);
        float mu = c.c0 * (a.lattices[idx] - c.c1 * a.pred_l[idx]);
        float v = a.keep_lattice ? a.lattices[idx] : mu + c.sigma * z;
        a.lattices[idx] = v;
        if (a.rec_lat) a.rec_lat[idx] = v;
        if (want_lp) lp_l = normal_log_prob(v, mu, c.sigma_sq, c.log_sigma);
    }
    if (want_lp) lp_l = block_sum_256(lp_l, red);

    // coordinates: x_{t-1} = (x_{t-1/2} - step * s + std z) % 1
    float lp_x = 0.f;
    for (int idx = n0 * 3 + tid; idx < n1 * 3; idx += 256) {
        float z = 0.f;
        if (t > 1) z = a.noise_x ? a.noise_x[idx] : philox_normal1(a.seed, (uint32_t)t, DRAW_PRED_X, (uint64_t)a.node_offset * 3 + idx);
        float px = a.pred_x[idx] * c.sqrt_sn;
        float drift = a.x_mid[idx] - c.step_pred * px;
        float v = pymod1(a.keep_coords ? a.x_mid[idx] : drift + c.std_pred * z);
        if (want_lp) lp_x += log_prob_wn(v, pymod1(drift), c.std_pred_sq);
        v = pymod1(v);  // traj[t-1]['frac_coords'] = x_{t-1} % 1  (:386)
        a.frac[idx] = v;
        if (a.rec_frac) a.rec_frac[idx] = v;
    }
    if (want_lp) lp_x = block_sum_256(lp_x, red);

    // atom-type logits: one wave per atom; a lane owns a QUAD of consecutive logits = one Philox call (100 logits = 25 quads, and the
    // global element index of a logit row starts at a multiple of 4), instead of one call -- four Box-Muller normals -- per logit
    float lp_t = 0.f;
    for (int i = n0 + wave; i < n1; i += 4) {
        float s = 0.f;
        if (lane < MI_NUM_TYPES / 4) {
            const int64_t idx0 = (int64_t)i * MI_NUM_TYPES + 4 * lane;
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (t > 1) {
                if (a.noise_t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[q] = a.noise_t[idx0 + q];
                } else {
                    philox_normal4(a.seed, (uint32_t)t, DRAW_PRED_T, ((uint64_t)a.node_offset * MI_NUM_TYPES + idx0) >> 2, z);
                }
            }
            const f32x4 at = *reinterpret_cast<const f32x4*>(a.atom_types + idx0), pt = *reinterpret_cast<const f32x4*>(a.pred_t + idx0);
            f32x4 vout;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float mu = c.c0 * (at[q] - c.c1 * pt[q]);
                const float v = mu + c.sigma * z[q];
                vout[q] = v;
                if (want_lp) s += normal_log_prob(v, mu, c.sigma_sq, c.log_sigma);
            }
            *reinterpret_cast<f32x4*>(a.atom_types + idx0) = vout;
            if (a.rec_types) *reinterpret_cast<f32x4*>(a.rec_types + idx0) = vout;
        }
        if (want_lp) {
            s = wave_sum(s);
            lp_t += s / (float)MI_NUM_TYPES;  // mean over the 100 logits (:358)
        }
    }
    if (!want_lp) return;
    // every lane of a wave holds the same lp_t; add the four waves
    __syncthreads();
    if (lane == 0) red[wave] = lp_t;
    __syncthreads();
    if (tid == 0 && t > 1) {
        float lpt = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
        if (a.rec_lpl) a.rec_lpl[b] = (lp_l / 3.0f) / 3.0f;  // .mean(-1).mean(-1) (:357)
        if (a.rec_lpt) a.rec_lpt[b] = lpt;
        if (a.rec_lpx) a.rec_lpx[b] = a.lp_corr[b] + (lp_x / 3.0f) / cnt;
    }
}

__global__ void wrap_copy_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = pymod1(x[i]);
}

// x <- x mod 1 in place (the chain starts from the wrapped coordinates, diffusion.py:289): done here rather than by a torch op of the caller,
// which on a chain's side stream would be a packed-fp32 kernel beside the other chains' GEMMs
__global__ void wrap_inplace_kernel(float* x, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = pymod1(x[i]);
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_time_embedding(const int* times, const float* freqs, int B, int time_dim, float* out, void* stream) {
    MI_CHECK(times && freqs && out && time_dim % 2 == 0, MI_EINVAL, "bad argument");
    if (B <= 0) return MI_OK;
    hipLaunchKernelGGL(time_embedding_kernel, dim3(cdiv((int64_t)B * time_dim, 256)), dim3(256), 0, (hipStream_t)stream, times, freqs,
                       out, B, time_dim);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_philox_fill(uint64_t seed, uint32_t step, uint32_t draw_id, int64_t elem_offset, int64_t n, int uniform, float* out,
                   void* stream) {
    MI_CHECK(out && n >= 0 && elem_offset >= 0, MI_EINVAL, "bad argument");
    if (n == 0) return MI_OK;
    hipLaunchKernelGGL(philox_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, step, draw_id, elem_offset, n,
                       uniform, out);
    MI_KERNEL_CHECK();
    return MI_OK;
}

int mi_sampler_init_state(mi_batch* b, uint64_t seed, int T, float* atom_types, float* frac, float* lattices, void* stream) {
    MI_CHECK(b && atom_types && frac && lattices, MI_EINVAL, "null argument");
    // diffusion.py:277-279; step field of the counter = T + 1
    MI_TRY(mi_philox_fill(seed, (uint32_t)(T + 1), DRAW_X_T, b->node_offset * 3, (int64_t)b->N * 3, 1, frac, stream));
    MI_TRY(mi_philox_fill(seed, (uint32_t)(T + 1), DRAW_L_T, b->graph_offset * 9, (int64_t)b->B * 9, 0, lattices, stream));
    MI_TRY(mi_philox_fill(seed, (uint32_t)(T + 1), DRAW_T_T, b->node_offset * MI_NUM_TYPES, (int64_t)b->N * MI_NUM_TYPES, 0,
                          atom_types, stream));
    return MI_OK;
}

int mi_sampler_run(mi_net* net, mi_batch* b, const float* coef_host, int T, int t_start, int t_stop, const float* time_freqs,
                   uint64_t seed, const mi_sampler_noise* noise, const mi_sampler_record* rec, float* atom_types, float* frac,
                   float* lattices, void* stream) {
    MI_CHECK(net && b && coef_host && time_freqs && atom_types && frac && lattices, MI_EINVAL, "null argument");
    MI_CHECK(T >= 1 && t_start <= T && t_stop >= 0 && t_stop <= t_start, MI_EINVAL, "bad step range T=%d start=%d stop=%d", T,
             t_start, t_stop);
    hipStream_t s = (hipStream_t)stream;
    const int N = b->N, B = b->B;
    if (N == 0 || B == 0) return MI_OK;
    const size_t ncoef = (size_t)(T + 1) * MI_NCOEF;
    if (b->coef_T != T) {
        MI_TRY(dev_alloc(b, &b->coef, ncoef));
        b->coef_T = T;
        b->coef_h.clear();
    }
    // the per-step scalars change only with the schedule buffers / step size: upload (and wait -- coef_host may be transient)
    // when they differ from the copy this batch already holds, otherwise the chain is enqueued without any host synchronisation
    if (b->coef_h.size() != ncoef || memcmp(b->coef_h.data(), coef_host, ncoef * sizeof(float)) != 0) {
        b->coef_h.assign(coef_host, coef_host + ncoef);
        MI_HIP(hipMemcpyAsync(b->coef, b->coef_h.data(), ncoef * sizeof(float), hipMemcpyHostToDevice, s));
        MI_HIP(hipStreamSynchronize(s));
    }

    const size_t n3 = (size_t)N * 3, nA = (size_t)N * MI_NUM_TYPES, b9 = (size_t)B * 9;
    hipLaunchKernelGGL(wrap_inplace_kernel, dim3(cdiv(n3, 256)), dim3(256), 0, s, frac, (int64_t)n3);
    if (rec) {  // traj[t_start] = current state (diffusion.py:287-293)
        if (rec->atom_types) MI_HIP(hipMemcpyAsync(rec->atom_types + t_start * nA, atom_types, nA * 4, hipMemcpyDeviceToDevice, s));
        if (rec->lattices) MI_HIP(hipMemcpyAsync(rec->lattices + t_start * b9, lattices, b9 * 4, hipMemcpyDeviceToDevice, s));
        if (rec->frac_coords)
            hipLaunchKernelGGL(wrap_copy_kernel, dim3(cdiv(n3, 256)), dim3(256), 0, s, frac, rec->frac_coords + t_start * n3, (int64_t)n3);
    }
    struct NoSyncScope {   // the chain's graph builds (knn edge style) do not synchronise: see mi_knn_graph_status
        mi_batch* b;
        explicit NoSyncScope(mi_batch* b_) : b(b_) { b->knn_nosync = true; }
        ~NoSyncScope() { b->knn_nosync = false; }
    } nosync_scope(b);
    for (int t = t_start; t > t_stop; --t) {
        TraceRange range("mi_sampler_step");
        hipLaunchKernelGGL(time_embedding_kernel, dim3(cdiv((int64_t)B * net->TD, 256)), dim3(256), 0, s, (const int*)nullptr, time_freqs, b->temb, B,
                           net->TD, t);
        // corrector
        // (the Langevin corrector reads the coordinate score alone, diffusion.py:310-322: the type columns of the heads and the lattice head are not evaluated)
        MI_TRY(net_forward(net, b, b->temb, atom_types, frac, lattices, b->pred_l, b->pred_x, b->pred_t, s, false, false, true));
        hipLaunchKernelGGL(corrector_kernel, dim3(B), dim3(64), 0, s, frac, b->pred_x, noise ? noise->corr_x + t * n3 : nullptr,
                           b->coef, t, seed, b->node_offset, b->node_off, b->x_mid, (rec && rec->log_prob_x) ? b->lp_corr : nullptr,
                           (rec && rec->frac_coords_mid) ? rec->frac_coords_mid + t * n3 : nullptr, b->keep_coords);
        MI_KERNEL_CHECK();
        // predictor
        // the corrector moved the coordinates only (diffusion.py:320-322): layer-0 node features are those of the evaluation above
        MI_TRY(net_forward(net, b, b->temb, atom_types, b->x_mid, lattices, b->pred_l, b->pred_x, b->pred_t, s, false, true));
        PredictorArgs a;
        a.x_mid = b->x_mid;
        a.pred_x = b->pred_x;
        a.pred_l = b->pred_l;
        a.pred_t = b->pred_t;
        a.noise_x = noise ? noise->pred_x + t * n3 : nullptr;
        a.noise_l = noise ? noise->pred_l + t * b9 : nullptr;
        a.noise_t = noise ? noise->pred_t + t * nA : nullptr;
        a.coef = b->coef;
        a.node_off = b->node_off;
        a.lp_corr = b->lp_corr;
        a.frac = frac;
        a.lattices = lattices;
        a.atom_types = atom_types;
        a.rec_types = (rec && rec->atom_types) ? rec->atom_types + (t - 1) * nA : nullptr;
        a.rec_frac = (rec && rec->frac_coords) ? rec->frac_coords + (t - 1) * n3 : nullptr;
        a.rec_lat = (rec && rec->lattices) ? rec->lattices + (t - 1) * b9 : nullptr;
        a.rec_lpl = (rec && rec->log_prob_l) ? rec->log_prob_l + (size_t)t * B : nullptr;
        a.rec_lpt = (rec && rec->log_prob_t) ? rec->log_prob_t + (size_t)t * B : nullptr;
        a.rec_lpx = (rec && rec->log_prob_x) ? rec->log_prob_x + (size_t)t * B : nullptr;
        a.seed = seed;
        a.node_offset = b->node_offset;
        a.graph_offset = b->graph_offset;
        a.t = t;
        a.keep_lattice = b->keep_lattice;
        a.keep_coords = b->keep_coords;
        hipLaunchKernelGGL(predictor_kernel, dim3(B), dim3(256), 0, s, a);
        MI_KERNEL_CHECK();
    }
    return MI_OK;
}

int mi_sampler_set_keep(mi_batch* b, int keep_lattice, int keep_coords) {
    MI_CHECK(b, MI_EINVAL, "null handle");
    b->keep_lattice = keep_lattice != 0;
    b->keep_coords = keep_coords != 0;
    return MI_OK;
}

int mi_debug_spin(long long cycles, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cycles, (int*)nullptr);
    MI_KERNEL_CHECK();
    return MI_OK;
}

}  // extern "C"

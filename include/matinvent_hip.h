/*
 * matinvent_hip.h -- C ABI of the MI355X (gfx950) hot path of MatInvent's RL-diffusion loop.
 *
 * The reference (schwallergroup/matinvent) has no FFI: its plug-in boundary is Python
 * duck-typing.  These entry points are what a ctypes binding for that
 * boundary binds; each one names the reference interface it replaces (path:line relative to
 * the reference checkout).  Plain pointers and sizes only, no torch types.
 *
 * Conventions
 *   - every `float*` / `int*` argument is a DEVICE pointer unless its name ends in `_host`;
 *   - fp32 everywhere, indices int32, row-major, contiguous;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued on it, nothing synchronises unless stated;
 *   - every function returns 0 on success, a negative MI_E* code otherwise;
 *     mi_last_error() returns a static, thread-local description;
 *   - B crystals, N = sum(num_atoms) atoms, fully-connected edges E = sum(num_atoms^2),
 *     H hidden width, L layers, F Fourier frequencies, TD time-embedding width,
 *     A = 100 atom-type logits (MAX_ATOMIC_NUM, models/diffcsp/cspnet.py:9).
 */
#ifndef MATINVENT_HIP_H
#define MATINVENT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument / unsupported hyper-parameter */
#define MI_EHIP (-2)     /* a HIP runtime call failed */
#define MI_ENOMEM (-3)    /* a device allocation failed or a caller-provided scratch / arena is too small */
#define MI_ECAPACITY (-5) /* a periodic neighbour graph exceeded the capacity it was created with (the caller may drop the batch) */
#define MI_ESTATE (-4)   /* call order violated (e.g. forward before mi_net_set_params) */

#define MI_NUM_TYPES 100

typedef struct mi_net mi_net;       /* score network: hyper-parameters + packed weights   */
typedef struct mi_batch mi_batch;   /* one batch of crystals: index tables + workspace    */

const char* mi_last_error(void);
int mi_version(void);
/* roctx ranges for host-side phases (the library brackets its own sampler steps and fine-tune micro-steps; the host mirror uses this
 * pair around the gradient all-reduce -- pipeline/mat_invent.py:166,177 is where the reference steps its optimizer).  No-ops unless
 * MI_ROCTX=1 is set when the library is loaded; show up under `rocprofv3 --marker-trace`. */
int mi_trace_push(const char* name);
int mi_trace_pop(void);

/* ---------------------------------------------------------------------------------------
 * Score network  --  replaces CSPNet (models/diffcsp/cspnet.py:94-294) as built by
 * DiffCSPModule.__init__ (models/diffcsp/diffusion.py:73: smooth=True, pred_type=True,
 * latent_dim += time_dim), fc edge style, dis_emb='sin', act 'silu', ip=True.
 * ------------------------------------------------------------------------------------- */
typedef struct mi_net_config {
    int hidden_dim;   /* H: any multiple of 64 in 64 .. 512                         */
    int num_layers;   /* L >= 1                                                  */
    int num_freqs;    /* F >= 1                                                  */
    int time_dim;     /* TD (latent_dim + time_dim of the reference), % 4 == 0   */
    int ln;           /* 1: LayerNorm in every layer + final (cspnet.py:87,276)  */
} mi_net_config;

int mi_net_create(const mi_net_config* cfg, mi_net** out);
void mi_net_destroy(mi_net* net);

/* Number of fp32 parameters and the flat layout: the reference's `decoder.*` state_dict
 * order (node_embedding.{weight,bias}, atom_latent_emb.{weight,bias}, per layer
 * edge_mlp.0.{w,b}, edge_mlp.2.{w,b}, node_mlp.0.{w,b}, node_mlp.2.{w,b},
 * layer_norm.{w,b}; coord_out.weight, lattice_out.weight, final_layer_norm.{w,b},
 * type_out.{w,b}).  mi_net_param_offset fills offset/numel of the i-th tensor in that
 * order and returns its name through a static string; returns MI_EINVAL past the end. */
int64_t mi_net_num_params(const mi_net* net);
int mi_net_num_tensors(const mi_net* net);
int mi_net_param_info(const mi_net* net, int index, const char** name, int64_t* offset, int64_t* numel,
                      int* rows, int* cols);

/* Bind the flat parameter vector `theta` (device, mi_net_num_params floats, 16-byte aligned)
 * and (re)build the MFMA-ready packed copies.  Must be called again after `theta` changes
 * (i.e. after every optimizer step).  `fourier_freqs_host` = F floats, the table
 * 2*pi*arange(F) of SinusoidsEmbedding (cspnet.py:16); copied on first call, may be NULL
 * afterwards. */
int mi_net_set_params(mi_net* net, const float* theta, const float* fourier_freqs_host, void* stream);

/* ---------------------------------------------------------------------------------------
 * Batch  --  replaces the PyG `Batch` bookkeeping the reference rebuilds on every decoder
 * call: num_atoms -> node2graph (repeat_interleave, cspnet.py:269) and the fully connected
 * edge list (block_diag + dense_to_sparse, cspnet.py:239-241, row-major incl. self loops).
 * `node_offset`/`graph_offset` are the global indices of this shard's first atom / crystal
 * (data-parallel sharding; they only enter the counter-based noise, so that a sharded run
 * draws the same numbers as a single-GPU run).
 * ------------------------------------------------------------------------------------- */
int mi_batch_create(const mi_net* net, const int* num_atoms_host, int B, int64_t node_offset,
                    int64_t graph_offset, mi_batch** out);
/* knn edge style (K17): the batch owns a periodic neighbour list that every forward rebuilds on the device from
 * (frac, lattices) -- CSPNet.gen_edges knn branch (cspnet.py:243-257) -> radius_graph_pbc (utils.py:335-514, 27 images,
 * cutoff = smallest inter-plane spacing + 0.01) + get_max_neighbors_mask (utils.py:517-601, `max_neighbors` with the
 * +0.01 band on d^2) + reorder_symmetric_edges (cspnet.py:159-234).  `edge_cap_per_node` bounds the kept, mask-selected
 * neighbours per centre atom (buffers are sized for 2 * N * cap edges; exceeding it is an MI_ECAPACITY error, never a
 * silent truncation).  At most 64 atoms per crystal. */
int mi_batch_create_knn(const mi_net* net, const int* num_atoms_host, int B, int64_t node_offset, int64_t graph_offset,
                        int max_neighbors, int edge_cap_per_node, mi_batch** out);
/* Build the list for the given coordinates without running the network (one host synchronisation); *num_edges = E''. */
int mi_knn_graph(mi_batch* b, const float* frac, const float* lattices, void* stream, int64_t* num_edges);
/* The reverse sampler (mi_sampler_run) rebuilds a knn batch's list in every evaluation WITHOUT a host round trip: its launches are sized for the capacity
 * and read the edge count on the device (the reference pays a host synchronisation per evaluation, models/diffcsp/cspnet.py:243-257 -> utils.py:335-514:
 * nonzero / masked_select).  A list over capacity cannot raise in the middle of an enqueued chain; it sets a sticky flag instead and contributes no edges.
 * This call synchronises `stream`, returns MI_ECAPACITY (and clears the flag) if any build since the last call exceeded the capacity, MI_OK otherwise.
 * matinvent_amd.diffcsp asks it wherever a chain's results are about to be read. */
int mi_knn_graph_status(mi_batch* b, void* stream);
/* Copy out the current list: edges [2][E''] int32 (row 0 = aggregation/source node, row 1 = neighbour) and
 * edge_vec [E''][3] (the `frac_diff` CSPNet.forward consumes).  MI_EDGE_ORDER_REFERENCE reproduces gen_edges' order
 * bit for bit; MI_EDGE_ORDER_CSR is the source-sorted order the kernels iterate in.  Either pointer may be NULL. */
#define MI_EDGE_ORDER_REFERENCE 0
#define MI_EDGE_ORDER_CSR 1
int mi_knn_graph_read(const mi_batch* b, int* edges, float* edge_vec, int order, void* stream);
/* K18: geometric validity quantities of a batch of structures (the step right after the sampler): per crystal
 * out[b][4] = { longest cell edge (A) -- the reference keeps structures with max(abc) < 25, pipeline/filters/opt_filter.py:53-55;
 * shortest interatomic distance over all pairs and 27 periodic images (A); cell volume |det L| (A^3); atom count }.
 * The thresholds of the external `structure_validity` check (distance / volume) are applied by the caller. */
int mi_structure_check(const mi_batch* b, const float* frac, const float* lattices, float* out, void* stream);
/* the same for any contiguous crystal layout: node_off [B+1] (device) = first atom of each crystal (MatterGen-side records) */
int mi_structure_check_offsets(const int* node_off, int B, const float* frac, const float* lattices, float* out, void* stream);
void mi_batch_destroy(mi_batch* b);
int mi_batch_num_nodes(const mi_batch* b);
int64_t mi_batch_num_edges(const mi_batch* b);
const int* mi_batch_node2graph(const mi_batch* b);   /* device, [N] */

/* CSPNet.forward (cspnet.py:260-294).
 *   t_emb [B,TD], atom_types [N,A] (continuous logits, smooth=True), frac [N,3],
 *   lattices [B,3,3]  ->  lattice_out [B,3,3] (already multiplied by L, :289),
 *   coord_out [N,3], type_out [N,A]. */
int mi_cspnet_forward(mi_net* net, mi_batch* b, const float* t_emb, const float* atom_types,
                      const float* frac, const float* lattices, float* lattice_out, float* coord_out,
                      float* type_out, void* stream);

/* Debug / parity taps: copy out the node features after layer `layer` ([N,H]; layer = L
 * means after the final LayerNorm) of the most recent forward on this batch. */
int mi_cspnet_tap(mi_net* net, mi_batch* b, int layer, float* out, void* stream);

/* SinusoidalTimeEmbeddings.forward (diffusion.py:53-66) for integer times.
 * `freqs` = device table exp(arange(TD/2) * -log(1e4)/(TD/2-1)) (built by the host mirror
 * with the same torch ops as the reference); times [B] int32 -> out [B,TD]. */
int mi_time_embedding(const int* times, const float* freqs, int B, int time_dim, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Reverse sampler  --  replaces DiffCSPModule.sample (diffusion.py:273-399).
 *
 * `coef` is a host table [T+1][MI_NCOEF] of the per-step scalars the reference derives from
 * its scheduler buffers at diffusion.py:297-343 (computed by the host mirror with the same
 * fp32 ops):  see MI_C_* below.  The state (atom_types [N,A], frac [N,3], lattices [B,3,3])
 * is updated in place from t = t_start down to t_stop+1; two network evaluations per step.
 *
 * Noise: if `noise` is NULL, draws come from the built-in counter-based Philox4x32-10 stream
 * (key = seed, counter = (element>>2, 0, draw_id, step); DESIGN.md "RNG").  Otherwise
 * `noise` points to caller-provided arrays (teacher-forced parity tests), indexed by step t.
 * Recording: if `rec` is non-NULL, every field that the reference stores in traj[t]
 * (diffusion.py:377-390) is written for each step.
 * ------------------------------------------------------------------------------------- */
#define MI_NCOEF 16
#define MI_C_C0 0          /* 1/sqrt(alpha_t)                       diffusion.py:302 */
#define MI_C_C1 1          /* (1-alpha_t)/sqrt(1-alphabar_t)        :303            */
#define MI_C_SIGMA 2       /* beta_scheduler.sigmas[t]              :305            */
#define MI_C_SQRT_SN 3     /* sqrt(sigmas_norm[t])                  :328            */
#define MI_C_STEP_CORR 4   /* step_lr*(sigma_x/sigma_begin)^2       :324            */
#define MI_C_STD_CORR 5    /* sqrt(2*step_corr)                     :325            */
#define MI_C_STEP_PRED 6   /* sigma_x^2 - sigma_{x,t-1}^2           :342            */
#define MI_C_STD_PRED 7    /* sqrt(adj^2*(sx^2-adj^2)/sx^2)         :343            */
#define MI_C_STD_CORR_SQ 8 /* std_corr**2 (for log_prob_wn)         :26-28          */
#define MI_C_STD_PRED_SQ 9 /* std_pred**2                                           */
#define MI_C_SIGMA_SQ 10   /* sigma**2      (Normal.log_prob var)                   */
#define MI_C_LOG_SIGMA 11  /* log(sigma)                                            */

typedef struct mi_sampler_noise {   /* all device pointers; slice t of each array is used at step t */
    const float* corr_x;   /* [T+1][N][3]   diffusion.py:322 */
    const float* pred_l;   /* [T+1][B][9]   :337             */
    const float* pred_t;   /* [T+1][N][A]   :338             */
    const float* pred_x;   /* [T+1][N][3]   :339             */
} mi_sampler_noise;

typedef struct mi_sampler_record {  /* all device pointers, may individually be NULL */
    float* atom_types;       /* [T+1][N][A]  traj[t]['atom_types']       */
    float* frac_coords;      /* [T+1][N][3]  traj[t]['frac_coords']      */
    float* lattices;         /* [T+1][B][9]  traj[t]['lattices']         */
    float* frac_coords_mid;  /* [T+1][N][3]  traj[t]['frac_coords_mid']  (t > 1) */
    float* log_prob_l;       /* [T+1][B]                                 (t > 1) */
    float* log_prob_t;       /* [T+1][B]                                          */
    float* log_prob_x;       /* [T+1][B]                                          */
} mi_sampler_record;

/* Initial state x_T ~ U[0,1), l_T, t_T ~ N(0,1) from the Philox stream (diffusion.py:277-279). */
int mi_sampler_init_state(mi_batch* b, uint64_t seed, int T, float* atom_types, float* frac,
                          float* lattices, void* stream);

int mi_sampler_run(mi_net* net, mi_batch* b, const float* coef_host, int T, int t_start, int t_stop,
                   const float* time_freqs, uint64_t seed, const mi_sampler_noise* noise,
                   const mi_sampler_record* rec, float* atom_types, float* frac, float* lattices,
                   void* stream);

/* CSP mode of DiffCSPModule.sample (diffusion.py:78-79, 283-287, 308-312, 330, 348-349): with keep_lattice / keep_coords the
 * lattice / the fractional coordinates handed to mi_sampler_run are the known ones and are never moved (the network is still
 * evaluated on them and the per-step log-probabilities are those of the reference's formulas).  Sticky per batch handle. */
int mi_sampler_set_keep(mi_batch* b, int keep_lattice, int keep_coords);

/* Fill `out` with n standard normals (uniform = 1: U[0,1)) of draw (step, draw_id), elements
 * [elem_offset, elem_offset + n) -- exposes the noise contract for tests. */
int mi_philox_fill(uint64_t seed, uint32_t step, uint32_t draw_id, int64_t elem_offset, int64_t n,
                   int uniform, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fine-tune step  --  replaces the autograd pass, optimizer and noising that
 * MatInvent.ft_step drives (pipeline/mat_invent.py:150-177).
 *
 * mi_cspnet_forward_train: CSPNet.forward that also keeps the activations its backward needs
 *   (held by the batch; one pending backward per batch -- any later forward on the same batch
 *   invalidates it).
 * mi_cspnet_backward: given dLoss/d(lattice_out, coord_out, type_out) ACCUMULATES dLoss/dtheta
 *   into grad_theta (flat, mi_net_num_params floats; `+=`, as `.backward()` does into `.grad`).
 *   Gradients w.r.t. the noised inputs are not produced (ft_step never uses them).
 * mi_adam_step: torch.optim.Adam with its defaults (mat_invent.py:136): bias-corrected, no weight
 *   decay / amsgrad, on flat buffers; `step` is 1-based; the gradient is read as grad*grad_scale.
 * mi_add_noise: DiffCSPModule.add_noise for one timestep (diffusion.py:81-119) with the per-step
 *   scalars c0 = sqrt(alphabar), c1 = sqrt(1-alphabar), sigma, sigmas_norm passed in.
 *   atom_types [N] int32 in 1..100.  rand_* = injected noise (rand_l [B,9], rand_x [N,3],
 *   rand_t [N,A]) or NULL for the Philox stream (draw ids 7/8/9, counter step = `step`).
 *   Outputs: noised lattice [B,9], frac [N,3], type logits [N,A]; targets tar_x [N,3] and the
 *   noise actually used as lattice / type targets (out_rand_l [B,9], out_rand_t [N,A]).
 * ------------------------------------------------------------------------------------- */
int mi_cspnet_forward_train(mi_net* net, mi_batch* b, const float* t_emb, const float* atom_types,
                            const float* frac, const float* lattices, float* lattice_out,
                            float* coord_out, float* type_out, void* stream);
int mi_cspnet_backward(mi_net* net, mi_batch* b, const float* d_lattice_out, const float* d_coord_out,
                       const float* d_type_out, float* grad_theta, void* stream);
/* Weight gradients of the node-level linears over a WINDOW of micro-steps (no reference counterpart: autograd contracts every
 * `.backward()` on its own, pipeline/mat_invent.py:164; between two optimizer steps, :166-167, the weights are constant, so the
 * gradient of a linear over k micro-steps is one contraction over k x N rows).  mi_batch_set_wgrad_window(net, b, k), k in 1..64:
 * training forwards / backwards on `b` keep the operand rows of node_mlp.{0,2} and of the node part of edge_mlp.0 per micro-step
 * (7 N H floats per layer and micro-step) -- and, since round 6, those of the heads and the embedding (lattice_out, type_out, coord_out,
 * atom_latent_emb, node_embedding: (4 H + 203) N floats per micro-step, written in place into the window by the forward and the backward) -- and
 * contract them when k micro-steps are pending -- into the grad_theta of THAT backward call -- or when mi_cspnet_wgrad_flush is called (before
 * anything reads grad_theta: the optimizer step, an all-reduce).  k = 0 (default) restores the immediate form.  Every other gradient (the edge-level
 * weights, the LayerNorms, the lattice term) is accumulated by mi_cspnet_backward as before.  Same sums up to fp32 summation order.
 * mi_batch_wgrad_pending: micro-steps not yet contracted. */
int mi_batch_set_wgrad_window(mi_net* net, mi_batch* b, int micro_steps);
int mi_cspnet_wgrad_flush(mi_net* net, mi_batch* b, float* grad_theta, void* stream);
int mi_batch_wgrad_pending(const mi_batch* b);
int mi_adam_step(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                 float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
int mi_add_noise(mi_batch* b, const float* lengths, const float* angles, const float* frac0,
                 const int* atom_types, float c0, float c1, float sigma, float sigma_norm, uint64_t seed,
                 uint32_t step, const float* rand_l, const float* rand_x, const float* rand_t,
                 float* in_lattice, float* in_frac, float* in_types, float* tar_x, float* out_rand_l,
                 float* out_rand_t, void* stream);

/* mi_add_noise with one timestep PER CRYSTAL (DiffCSPModule.add_noise(batch) without `time`, diffusion.py:83-84, which draws
 * the times with numpy on the host): sched [B][4] (device) = {sqrt(alphabar_t), sqrt(1-alphabar_t), sigma_t, sigmas_norm_t} of
 * each crystal's time. */
int mi_add_noise_per_crystal(mi_batch* b, const float* lengths, const float* angles, const float* frac0, const int* atom_types,
                             const float* sched, uint64_t seed, uint32_t step, const float* rand_l, const float* rand_x,
                             const float* rand_t, float* in_lattice, float* in_frac, float* in_types, float* tar_x,
                             float* out_rand_l, float* out_rand_t, void* stream);

/* ---------------------------------------------------------------------------------------
 * Profiling hook used by bench.py: when enabled, the dominant kernel (edge-message MLP) is
 * bracketed by hipEvents on its launch stream; mi_profile_read synchronises and returns the
 * number of launches and their summed duration in milliseconds since the last reset.
 * ------------------------------------------------------------------------------------- */
/* One fine-tune timestep of MatInvent.ft_step (pipeline/mat_invent.py:152-164) enqueued end to end:
 * add_noise -> agent forward (kept) -> frozen-prior forward -> fused per-sample loss / anchor penalty /
 * reward weighting + gradient seeds -> agent backward, i.e.
 *     grad_theta += d/dtheta [ sum_b ( r_b L_b + kl_sigma (1.1 - r_b) KL_b ) / (b_global * accum_steps) ].
 * `t` = diffusion time (T - timestep index, diffusion.py:86-87); c0/c1/sigma_t/sigma_norm as for mi_add_noise;
 * `reward` [B] device; rand_* optional injected noise.  `stats` (device, 3 floats, may be NULL) accumulates
 * the three quantities the reference logs (:168-170): the accum-normalised loss, sum_b r_b L_b,
 * sum_b (1.1-r_b) KL_b.  out_sample_loss / out_kl ([B], may be NULL) receive L_b / KL_b.
 * `ab` / `pb` must be distinct batch handles of the agent / prior networks for the same crystals.
 * `aux_stream` (may be NULL): a second stream onto which the frozen prior's forward is forked, so that it overlaps the agent's
 * forward (worthwhile for small fine-tune sets, which leave most of the chip idle); joined before the loss. */
int mi_ft_micro_step(mi_net* agent, mi_batch* ab, mi_net* prior, mi_batch* pb, const float* lengths,
                     const float* angles, const float* frac0, const int* atom_types, const float* reward,
                     const float* time_freqs, int t, float c0, float c1, float sigma_t, float sigma_norm,
                     uint64_t seed, uint32_t noise_step, const float* rand_l, const float* rand_x,
                     const float* rand_t, float cost_lattice, float cost_coord, float cost_type, float kl_sigma,
                     int b_global, int accum_steps, float* grad_theta, float* stats, float* out_sample_loss,
                     float* out_kl, void* stream, void* aux_stream);

/* `copies` consecutive timesteps of one gradient-accumulation window (mat_invent.py:146-167: the weights only change at the
 * optimizer step, so the window's timesteps are independent) as ONE micro-step over a batch holding `copies` replicas of the
 * fine-tune set: ab / pb are batch handles for the atom counts repeated `copies` times (same node / graph offsets as the
 * unstacked handles), the input arrays and `reward` are the replicated ones, replica c is noised for diffusion time t_host[c]
 * with the schedule values *_host[c] and noise call `noise_step + c`, its draws indexed by the ORIGINAL crystal / atom ids -- so
 * the accumulated gradient and `stats` are those of `copies` successive mi_ft_micro_step calls (up to fp32 summation order).
 * For small fine-tune sets (the reference's default is 18 crystals), where a single timestep is bound by the host's launch rate. */
#define MI_MAX_STACK 16
int mi_ft_micro_steps_stacked(mi_net* agent, mi_batch* ab, mi_net* prior, mi_batch* pb, const float* lengths,
                              const float* angles, const float* frac0, const int* atom_types, const float* reward,
                              const float* time_freqs, int copies, const int* t_host, const float* c0_host,
                              const float* c1_host, const float* sigma_t_host, const float* sigma_norm_host, uint64_t seed,
                              uint32_t noise_step, const float* rand_l, const float* rand_x, const float* rand_t,
                              float cost_lattice, float cost_coord, float cost_type, float kl_sigma, int b_global,
                              int accum_steps, float* grad_theta, float* stats, void* stream, void* aux_stream);

/* ---------------------------------------------------------------------------------------
 * Arithmetic paths.  Both reproduce the reference to fp32 round-off (tests state the bounds).
 *   GEMM mode (process-wide): MI_GEMM_SPLIT (default) evaluates every fp32 product on the bf16 matrix
 *     pipe from three bf16 planes per operand (six terms; measured max error below the f32 MFMA's);
 *     MI_GEMM_F32 uses v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain (1/16 the bf16 rate).
 *   Edge mode (per network): MI_EDGE_GEMM (default) runs the per-edge MLP as two tiled GEMMs over
 *     the edge list with gather / SiLU epilogues; MI_EDGE_FUSED_F32 is the register-chained
 *     f32-MFMA kernel that keeps the intermediate in registers.
 * ------------------------------------------------------------------------------------- */
#define MI_GEMM_F32 0
#define MI_GEMM_SPLIT 1
#define MI_EDGE_FUSED_F32 0
#define MI_EDGE_GEMM 1
int mi_set_gemm_mode(int mode);
int mi_net_set_edge_mode(mi_net* net, int mode);
/* fc edge style on the plane-GEMM path: run the Fourier-block GEMM over unordered node pairs (default on).  The reversed edge's
 * features are (-sin, +cos) of the same arguments, so one operand row yields both directed edges; off = one row per edge. */
int mi_set_edge_pairs(int on);
/* How many crystal groups of one fine-tune set the caller runs CONCURRENTLY on separate streams (matinvent_amd/finetune.py: the data-parallel
 * arithmetic of pipeline/mat_invent.py:150-177 inside one GPU; default 1).  The weight-gradient contractions split their row lists into enough
 * workgroups to fill the chip; with n groups in flight each launch needs 1 / n of them, and every split it does not make saves a partial tile's
 * write and re-read (measured at 256 crystals x 20 atoms on four groups: 21.1k -> 22.0k crystal-timesteps/s).  Results change by fp32
 * summation order only.  Returns the previous value. */
int mi_set_concurrent_groups(int n);

/* Element format of the pre-split plane sets this library was built with: 2 = two fp16 planes with per-class power-of-two scales
 * (three MFMA terms per product; the default), 3 = three bf16 planes (six terms, no range limits; build with -DMI_PLANES_FP16=0). */
int mi_plane_format(void);
/* MFMA terms per product of two plane sets: 3 in the product library (fp32-class: the dropped fourth term is 2^-22 relative), 1 in the
 * TF32-CLASS build lib/libmatinvent_hip_tf32.so (11-bit operands -- what the reference runs after torch.set_float32_matmul_precision("high"),
 * pipeline/mat_invent.py:127; a labelled secondary benchmark line with its own stated tolerance, never the default). */
int mi_terms_per_product(void);
/* Saturation guard of the two-plane fp16 operand format: every fp32 -> plane conversion that had to clamp to the fp16 range (or
 * met a NaN / inf) increments a device-side counter.  Synchronises the device, returns the number of such conversions since the
 * last reset (all networks, all streams of the current device) and clears it when `reset` != 0.  A non-zero count means results
 * computed since the reset are NOT within the stated fp32-class tolerance (use the three-plane bf16 build, -DMI_PLANES_FP16=0, for
 * networks whose weights exceed 1023 or whose LayerNorm outputs exceed 8188); zero means no conversion lost range. */
int mi_saturation_events(int64_t* count, int reset);
int mi_profile_enable(mi_net* net, int on);
int mi_profile_read(mi_net* net, int64_t* launches, double* total_ms, double* union_ms);


/* =======================================================================================
 * MatterGen-shaped path  --  what the files under models/mattergen/ adapt from the un-vendored package
 * `mattergen @ 5bb2b397` (PARITY UNPINNED; the arithmetic below follows the published
 * GemNet-T / MatterGen description as restated, definition by definition, in oracle/mattergen_oracle.py).
 * Replaces, behind MatterGenModule (models/mattergen/pl_module.py:17-125):
 *   `self.diffusion_module.model(noisy_batch, t)`          (pl_module.py:42,73)   -> mi_gemnet_forward
 *   autograd through it (pipeline/mat_invent.py:164)                               -> mi_gemnet_backward
 *   `self.diffusion_module.corruption.sample_marginal`     (pl_module.py:68)      -> mi_mg_sample_marginal
 *   `PredictorCorrector.sample` as driven by draw_samples_from_sampler (models/mattergen/sample.py:27-64) -> mi_mg_sampler_run
 * Shapes: B crystals, N atoms, E directed edges of the periodic radius graph (rebuilt by every forward), fractional
 * positions pos [N,3], cell [B,3,3] (rows = lattice vectors), atomic numbers [N] int32 in 1..100 or 101 = the D3PM mask
 * state, diffusion time t [B] in (0, 1].  Outputs: pos score x std [N,3] (fractional), cell score x std [B,3,3]
 * (symmetric), type logits [N,101].
 * ======================================================================================= */
#define MI_MG_CLASSES 101
#define MI_MG_MASK 101
typedef struct mi_gemnet mi_gemnet;
typedef struct mi_gbatch mi_gbatch;
typedef struct mi_gemnet_config {
    int emb_atom, emb_edge, emb_trip, emb_rbf, emb_cbf, emb_bil;   /* 512 / 512 / 64 / 16 / 16 / 64                     */
    int num_radial, num_spherical, num_blocks;                     /* 128 / 7 / 4                                       */
    int num_before_skip, num_after_skip, num_concat, num_atom;     /* 1 / 2 / 1 / 3 residual layers                     */
    int max_neighbors, max_images;                                 /* 50 nearest per target atom; <= 5 images per side  */
    float cutoff;                                                  /* 7.0 A                                             */
} mi_gemnet_config;
int mi_gemnet_create(const mi_gemnet_config* cfg, mi_gemnet** out);
void mi_gemnet_destroy(mi_gemnet* net);
int64_t mi_gemnet_num_params(const mi_gemnet* net);
int mi_gemnet_num_tensors(const mi_gemnet* net);
/* i-th tensor of the flat parameter vector (the order of oracle/mattergen_oracle.py::param_list): name, offset, numel, rows, cols */
int mi_gemnet_param_info(const mi_gemnet* net, int index, const char** name, int64_t* offset, int64_t* numel, int* rows, int* cols);
/* bind the flat parameter vector (device); call again after every optimizer step (transposed copies for the backward) */
int mi_gemnet_set_params(mi_gemnet* net, const float* theta, void* stream);
int mi_gbatch_create(const mi_gemnet* net, const int* num_atoms_host, int B, int64_t node_offset, int64_t graph_offset,
                     mi_gbatch** out);
/* Global ids of the batch's first atom / crystal (the Philox counters of the noising and sampler kernels): lets one batch handle --
 * and its activation arenas -- serve several equal-shaped chunks of a larger set (the chunked fine-tune loop). */
int mi_gbatch_set_offsets(mi_gbatch* b, int64_t node_offset, int64_t graph_offset);
void mi_gbatch_destroy(mi_gbatch* b);
/* Periodic radius graph of (pos, cell): cutoff, the max_neighbors nearest per target atom, symmetrised; one host
 * synchronisation (the edge count sizes the later launches).  *num_edges = E. */
int mi_gemnet_graph(mi_gemnet* net, mi_gbatch* b, const float* pos, const float* cell, void* stream, int64_t* num_edges);
/* copy out the current graph (any pointer may be NULL): src/dst [E], img [E][3], swap [E] (index of the reverse edge),
 * rowptr [N+1] (edges are sorted by dst), D [E], V [E][3] (unit vector from dst to the periodic image of src) */
int mi_gemnet_graph_read(const mi_gbatch* b, int* src, int* dst, int* img, int* swap, int* rowptr, float* D, float* V, void* stream);
/* The denoiser.  `train` is a bit set: 1 keeps the activations for mi_gemnet_backward (one pending backward per batch handle);
 * 2 (inference only) runs WITHOUT the host synchronisation that reads the graph's edge count: the edge-level launches are sized by the
 * capacity 2 max_neighbors N and read the count on the device, and a crystal over a graph capacity is taken out of the graph (no
 * edges) instead of failing the call -- its flag stays in the handle until mi_gbatch_graph_status reads it.  The reverse-diffusion
 * chain (models/mattergen/sample.py:27-64 drives 2 x 1000 such evaluations per batch) uses this form. */
int mi_gemnet_forward(mi_gemnet* net, mi_gbatch* b, const float* pos, const float* cell, const int* atomic_numbers,
                      const float* t, float* out_pos, float* out_cell, float* out_logits, int train, void* stream);
/* grad_theta += dLoss/dtheta given dLoss/d(out_pos, out_cell, out_logits) (any may be NULL = zero) */
int mi_gemnet_backward(mi_gemnet* net, mi_gbatch* b, const float* d_pos, const float* d_cell, const float* d_logits,
                       float* grad_theta, void* stream);
/* Per-crystal graph-capacity flags after mi_mg_sampler_run / mi_gemnet_forward: bad_host[i] != 0 -- crystal i exceeded a capacity of
 * the periodic graph at some evaluation (1: more than max_neighbors kept pairs of one atom, 2: more than 512 atoms inside the cutoff
 * even after shrinking it, 4: an in-degree above 128) and ran without edges from then on: its sample is invalid, the others are
 * unaffected.  The counterpart of the reference's per-crystal invalid_filter (pipeline/filters/opt_filter.py:49-61), which drops
 * collapsed crystals one by one after sampling.  bad_host (B ints) and n_bad may be NULL.  Waits for the work queued on `stream`. */
int mi_gbatch_graph_status(mi_gbatch* b, int* bad_host, int* n_bad, void* stream);
/* parity taps of the most recent forward: "h<i>" [N,emb_atom], "m<i>" [E,emb_edge] after block i (0 = embedding), "rbf" [E,num_radial] */
int mi_gemnet_tap(mi_gbatch* b, const char* name, float* out, int64_t capacity, int64_t* numel, void* stream);

/* Corruption constants (oracle/mattergen_oracle.py::Corruption): wrapped VE-SDE on positions (std = sigma_min^(1-t) sigma_max^t
 * n^(-1/3)), VP-SDE on the cell towards (n / limit_density)^(1/3) I with std sqrt(limit_var_scale) n^(1/3), D3PM absorbing
 * state on the types (d3pm_steps discrete steps). */
typedef struct mi_mg_corruption {
    float sigma_min, sigma_max, beta_min, beta_max, limit_density, limit_var_scale;
    int d3pm_steps;
} mi_mg_corruption;
/* corruption.sample_marginal (pl_module.py:68) for per-crystal times t [B].  noise_* = injected draws (pos [N,3] normal,
 * cell [B,9] normal, types [N] uniform) or NULL for the Philox stream (draw ids 10 / 11 / 12, counter step = `step`).
 * Outputs: noisy pos / cell / types, and what the loss needs: delta [N,3] (= std z), eps [B,9] (symmetric noise),
 * masked [N] (0/1). */
int mi_mg_sample_marginal(mi_gbatch* b, const mi_mg_corruption* c, const float* pos0, const float* cell0, const int* types0,
                          const float* t, uint64_t seed, uint32_t step, const float* noise_pos, const float* noise_cell,
                          const float* noise_types, float* pos, float* cell, int* types, float* delta, float* eps, int* masked,
                          void* stream);
typedef struct mi_mg_sampler_noise {   /* device pointers, [n_steps][...] each; NULL struct = Philox stream */
    const float *corr_pos, *corr_cell, *pred_pos, *pred_cell, *pred_u1, *pred_u2;
} mi_mg_sampler_noise;
/* initial state of the reverse chain: pos ~ U[0,1), cell = limit mean + limit std x symmetric noise, types = mask */
int mi_mg_sampler_init(mi_gbatch* b, const mi_mg_corruption* c, uint64_t seed, const float* init_pos, const float* init_cell,
                       float* pos, float* cell, int* types, void* stream);
/* predictor-corrector steps i_start .. i_stop-1 of the time grid ts_host [n_steps] (= linspace(1, eps_t, n_steps) built by the host
 * mirror with the same fp32 ops as the oracle): Langevin corrector
 * (positions snr 0.4, cell snr 0.2), ancestral predictor, two denoiser evaluations per step; state updated in place,
 * mean_pos / mean_cell receive the last predictor mean (what MatterGenSampler returns, sample.py:49-50). */
int mi_mg_sampler_run(mi_gemnet* net, mi_gbatch* b, const mi_mg_corruption* c, int n_steps, int i_start, int i_stop, const float* ts_host,
                      uint64_t seed, const mi_mg_sampler_noise* noise, float* pos, float* cell, int* types, float* mean_pos,
                      float* mean_cell, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MATINVENT_HIP_H */

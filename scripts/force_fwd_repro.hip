// Does force_fwd_kernel (csrc/gemnet.hip: the position head of the MatterGen-shaped denoiser) reproduce itself on FROZEN inputs while three other
// streams keep the chip busy with long matrix-pipe kernels?  Round 3 saw one quarter-wave (16 atoms) of its output differ in ~1 of 3 trials of four
// concurrent forwards and could not tell "this kernel is hit" from "an upstream tensor was transiently different".  This program has
// nothing upstream: the same launch 10^4 times, every output compared on the device.
//   mode 0: the head alone on frozen Fe / V / rowptr / cell.
//   mode 1: the head behind its producers, as in the forward: Fe is rewritten (one kernel) and accumulated four times (rowdot_short's
//           read-modify-write form) in front of every launch of the head -- kernel -> kernel visibility under concurrency.
//   hipcc --offload-arch=gfx950 -O3 scripts/force_fwd_repro.hip -o /tmp/ffr && /tmp/ffr [launches] [mode] [busy streams] [busy kind] [head variant] [no memset]
//   busy kind: 0 = matrix-pipe loop, 64 KiB LDS, 256 registers (two workgroups fill a CU); 1 = plain-fma VALU loop, no LDS, few registers;
//              2 = matrix-pipe loop without LDS at <= 128 registers (the head's waves fit beside it)
//              3 / 4 = kind 0 without its LDS + fp16 tail / with it at four accumulators; 5 = matrix pipe + LDS read only; 6 = matrix pipe + fp16 VALU only;
//              7 = the head itself on the other streams; 8 = kind 6's fp16 VALU work without the matrix pipe; 9 = kind 6 with a fixed vector element
//   head variant: 0 = as in the library (IEEE division in inv3); 1 = v_rcp_f32 instead of the division; 2 = no inverse (pos = force sums)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int VAR>
__device__ __forceinline__ void inv3(const float* L, float* I) {
    const float a = L[0], b = L[1], c = L[2], d = L[3], e = L[4], f = L[5], g = L[6], h = L[7], i = L[8];
    const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * B + c * C;
    const float id = VAR == 1 ? __builtin_amdgcn_rcpf(det) : 1.0f / det;
    I[0] = A * id; I[1] = -(b * i - c * h) * id; I[2] = (b * f - c * e) * id;
    I[3] = B * id; I[4] = (a * i - c * g) * id;  I[5] = -(a * f - c * d) * id;
    I[6] = C * id; I[7] = -(a * h - b * g) * id; I[8] = (a * e - b * d) * id;
}
// pos[a] = (sum_{e into a} F[e] V[e]) @ inv(L)   -- the kernel under test, as in csrc/gemnet.hip
template <int VAR>
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
    if (VAR == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) pos[(size_t)a * 3 + c] = f[c];
        return;
    }
    float I[9];
    inv3<VAR>(cell + (size_t)n2g[a] * 9, I);
#pragma unroll
    for (int c = 0; c < 3; ++c) pos[(size_t)a * 3 + c] = f[0] * I[c] + f[1] * I[3 + c] + f[2] * I[6 + c];
}
__global__ void set_kernel(float* __restrict__ y, const float* __restrict__ base, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = base[i];
}
__global__ void acc_kernel(float* __restrict__ y, const float* __restrict__ add, int n, float w) {   // y[r] = y[r] + s (rowdot_short, acc = 1)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = y[i] + w * add[i];
}
__global__ void compare_kernel(const unsigned* __restrict__ a, const unsigned* __restrict__ ref, int n, unsigned* __restrict__ bad, int launch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != ref[i]) {
        const unsigned k = atomicAdd(bad, 1u);
        if (k < 30) {   // {launch, element, got, expected} of the first mismatches
            bad[4 + 4 * k] = (unsigned)launch;
            bad[5 + 4 * k] = (unsigned)i;
            bad[6 + 4 * k] = a[i];
            bad[7 + 4 * k] = ref[i];
        }
    }
}
// what the other streams run: a long matrix-pipe loop with the register / LDS footprint of the dense-layer kernel (two workgroups per CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void busy_kernel(float* out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    lds[threadIdx.x] = (float)lane;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        a[it & 7] += (_Float16)lds[(threadIdx.x + it) & 255] * (_Float16)1e-4f;
    }
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void busy_valu_kernel(float* out, int iters) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.01f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int t = 0; t < 12; ++t) v[q] = __builtin_fmaf(v[q], 0.999f, 0.001f);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void busy_mfma_slim_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}

// kinds 3 / 4: the matrix-pipe loop of kind 0 (eight accumulators, two waves per SIMD) WITHOUT the LDS traffic / with the LDS read but four accumulators
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void busy_mfma8_nolds_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void busy_mfma4_lds_kernel(float* out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    lds[threadIdx.x] = (float)lane;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        a[it & 7] += (_Float16)lds[(threadIdx.x + it) & 255] * (_Float16)1e-4f;
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}

// kinds 5 / 6 split kind 4's loop tail: 5 = matrix-pipe loop + an LDS read per iteration feeding an fp32 add (no fp16 VALU work);
// 6 = matrix-pipe loop + the fp16 VALU work on a register value (no LDS traffic)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void busy_mfma4_ldsonly_kernel(float* out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    lds[threadIdx.x] = (float)lane;
    __syncthreads();
    float t = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        t += lds[(threadIdx.x + it) & 255];
    }
    float s = t;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void busy_mfma4_f16valu_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float x = 0.001f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        x = x * 1.0001f + 1e-7f;
        a[it & 7] += (_Float16)x * (_Float16)1e-4f;
    }
    float s = x;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}

// kinds 8 / 9: kind 6's fp16 VALU work WITHOUT the matrix pipe (8), and kind 6 with a fixed vector element instead of the rotating one (9)
__global__ __launch_bounds__(256) void busy_f16valu_only_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (lane + i));
    float x = 0.001f * lane;
    for (int it = 0; it < 8 * iters; ++it) {
        x = x * 1.0001f + 1e-7f;
        a[it & 7] += (_Float16)x * (_Float16)1e-4f;
    }
    float s = x;
    for (int i = 0; i < 8; ++i) s += (float)a[i];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void busy_mfma4_f16fixed_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float x = 0.001f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        x = x * 1.0001f + 1e-7f;
        a[0] += (_Float16)x * (_Float16)1e-4f;
    }
    float s = x;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 10000, mode = argc > 2 ? atoi(argv[2]) : 0, nbusy = argc > 3 ? atoi(argv[3]) : 3;
    const int busy_kind = argc > 4 ? atoi(argv[4]) : 0, head_var = argc > 5 ? atoi(argv[5]) : 0, no_memset = argc > 6 ? atoi(argv[6]) : 0;
    const int B = 64, n = 20, N = B * n;
    srand(3);
    std::vector<int> rowptr(N + 1, 0), n2g(N);
    for (int a = 0; a < N; ++a) { rowptr[a + 1] = rowptr[a] + 44 + rand() % 14; n2g[a] = a / n; }
    const int E = rowptr[N];
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    std::vector<float> F(E), Fadd(E), V((size_t)E * 3), cell((size_t)B * 9);
    for (auto& x : F) x = rnd();
    for (auto& x : Fadd) x = rnd();
    for (auto& x : V) x = rnd();
    for (int b = 0; b < B; ++b) for (int q = 0; q < 9; ++q) cell[b * 9 + q] = (q % 4 == 0 ? 7.f : 0.f) + 0.3f * rnd();
    int *d_rowptr, *d_n2g;
    float *d_Fbase, *d_Fadd, *d_F, *d_V, *d_cell, *d_pos, *d_ref, *d_ref2, *d_sink;
    unsigned* d_bad;
    CK(hipMalloc(&d_rowptr, (N + 1) * 4)); CK(hipMalloc(&d_n2g, N * 4));
    CK(hipMalloc(&d_Fbase, E * 4)); CK(hipMalloc(&d_Fadd, E * 4)); CK(hipMalloc(&d_F, E * 4)); CK(hipMalloc(&d_V, (size_t)E * 12));
    CK(hipMalloc(&d_cell, B * 36)); CK(hipMalloc(&d_pos, N * 12)); CK(hipMalloc(&d_ref, N * 12)); CK(hipMalloc(&d_ref2, N * 12)); CK(hipMalloc(&d_bad, 128 * 4)); CK(hipMalloc(&d_sink, 64));
    auto head = [&](hipStream_t s, float* out) {
        if (head_var == 1) hipLaunchKernelGGL(force_fwd_kernel<1>, dim3((N + 255) / 256), dim3(256), 0, s, d_F, d_V, d_rowptr, d_n2g, d_cell, out, N);
        else if (head_var == 2) hipLaunchKernelGGL(force_fwd_kernel<2>, dim3((N + 255) / 256), dim3(256), 0, s, d_F, d_V, d_rowptr, d_n2g, d_cell, out, N);
        else hipLaunchKernelGGL(force_fwd_kernel<0>, dim3((N + 255) / 256), dim3(256), 0, s, d_F, d_V, d_rowptr, d_n2g, d_cell, out, N);
    };
    CK(hipMemcpy(d_rowptr, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_n2g, n2g.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_Fbase, F.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_Fadd, Fadd.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_V, V.data(), (size_t)E * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cell, cell.data(), B * 36, hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 128 * 4));
    CK(hipFuncSetAttribute((const void*)busy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)busy_mfma4_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)busy_mfma4_ldsonly_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    auto produce = [&](hipStream_t s) {
        hipLaunchKernelGGL(set_kernel, dim3((E + 255) / 256), dim3(256), 0, s, d_F, d_Fbase, E);
        if (mode == 1)
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(acc_kernel, dim3((E + 255) / 256), dim3(256), 0, s, d_F, d_Fadd, E, 0.25f * (k + 1));
    };
    // reference: the head alone on an idle device
    produce(s0);
    head(s0, d_ref);
    CK(hipStreamSynchronize(s0));
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int k = 0; k < nbusy; ++k)
        th.emplace_back([&, k] {
            hipStream_t s;
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            int q = 0;
            while (!stop.load()) {   // ~0.5 ms per launch, 2 x 256 workgroups: the footprint of one dense layer of a 64-crystal chain
                if (busy_kind == 1) hipLaunchKernelGGL(busy_valu_kernel, dim3(2048), dim3(256), 0, s, d_sink, 1500);
                else if (busy_kind == 2) hipLaunchKernelGGL(busy_mfma_slim_kernel, dim3(1024), dim3(256), 0, s, d_sink, 3000);
                else if (busy_kind == 3) hipLaunchKernelGGL(busy_mfma8_nolds_kernel, dim3(512), dim3(256), 0, s, d_sink, 1500);
                else if (busy_kind == 4) hipLaunchKernelGGL(busy_mfma4_lds_kernel, dim3(512), dim3(256), 65536, s, d_sink, 3000);
                else if (busy_kind == 5) hipLaunchKernelGGL(busy_mfma4_ldsonly_kernel, dim3(512), dim3(256), 65536, s, d_sink, 3000);
                else if (busy_kind == 6) hipLaunchKernelGGL(busy_mfma4_f16valu_kernel, dim3(512), dim3(256), 0, s, d_sink, 3000);
                else if (busy_kind == 7) head(s, d_ref2);   // the same kernel as the victim, on another stream
                else if (busy_kind == 8) hipLaunchKernelGGL(busy_f16valu_only_kernel, dim3(2048), dim3(256), 0, s, d_sink, 3000);
                else if (busy_kind == 9) hipLaunchKernelGGL(busy_mfma4_f16fixed_kernel, dim3(512), dim3(256), 0, s, d_sink, 3000);
                else hipLaunchKernelGGL(busy_kernel, dim3(512), dim3(256), 65536, s, d_sink, 1500);
                if ((++q & 7) == 0) CK(hipStreamSynchronize(s));
            }
            CK(hipStreamSynchronize(s));
        });
    for (int it = 0; it < launches; ++it) {
        if (mode == 1) produce(s0);
        if (!no_memset) CK(hipMemsetAsync(d_pos, 0xFF, N * 12, s0));
        head(s0, d_pos);
        hipLaunchKernelGGL(compare_kernel, dim3((N * 3 + 255) / 256), dim3(256), 0, s0, (const unsigned*)d_pos, (const unsigned*)d_ref, N * 3, d_bad, it);
        if ((it & 255) == 255) CK(hipStreamSynchronize(s0));
    }
    CK(hipStreamSynchronize(s0));
    stop.store(true);
    for (auto& t : th) t.join();
    unsigned bad[128];
    CK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    printf("mode %d, busy kind %d x %d streams, head variant %d, memset %d: %d launches of the head (%d atoms, %d edges): %u mismatching output words\n", mode, busy_kind, nbusy,
           head_var, !no_memset, launches, N, E, bad[0]);
    for (unsigned k = 0; k < (bad[0] < 12 ? bad[0] : 12); ++k) {
        float got, ex;
        memcpy(&got, &bad[6 + 4 * k], 4);
        memcpy(&ex, &bad[7 + 4 * k], 4);
        printf("  launch %u element %u (atom %u, lane %u, component %u): got %08x (%g) expected %08x (%g)\n", bad[4 + 4 * k], bad[5 + 4 * k], bad[5 + 4 * k] / 3, (bad[5 + 4 * k] / 3) & 63,
               bad[5 + 4 * k] % 3, bad[6 + 4 * k], got, bad[7 + 4 * k], ex);
    }
    return bad[0] != 0;
}

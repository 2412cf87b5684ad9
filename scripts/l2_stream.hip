// How many bytes per second do the CUs pull through L2 -> L1 with 16-byte loads?  (round 3: every large kernel of the pinned path -- the two
// edge GEMMs at 128 x 128 tiles, the node chain -- moves 8-10 TB/s of operand bytes from L2 into the CUs, whatever its loop looks like.)
// Each workgroup streams a window of `span` bytes of one buffer `reps` times with dwordx4 loads, eight in flight per lane; the windows of
// different workgroups start at different offsets.  span = 2 MiB: L2-resident (a layer's weights); 64 MiB: Infinity Cache; 1 GiB: HBM.
//   hipcc --offload-arch=gfx950 -O3 scripts/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ buf, size_t span16, size_t total16, int reps, unsigned* out) {
    const size_t start = ((size_t)blockIdx.x * 7919 * 4096) & (total16 - 1);   // (windows are powers of two)
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + 7 * 256 < span16; i += 8 * 256) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = buf[(start + i + u * 256) & (total16 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
int main() {
    const size_t total = (size_t)1 << 30;
    u32x4* buf; unsigned* out;
    (void)hipMalloc(&buf, total); (void)hipMalloc(&out, 4); (void)hipMemset(buf, 1, total);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (size_t window : {(size_t)2 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)1 << 30}) {   // the bytes ALL workgroups share
        for (int wgs : {256, 512, 1024, 2048}) {
            const size_t span = (size_t)1 << 20;  // bytes per workgroup and rep
            const int reps = 16;
            for (int it = 0; it < 2; ++it) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), 0, 0, buf, span / 16, window / 16, reps, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            }
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("window %5zu MiB  %4d workgroups: %.2f TB/s\n", window >> 20, wgs, (double)wgs * span * reps / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

// probe of ds_read_b64_tr_b16 on gfx950: LDS element i holds the value i (u16); every lane reads at a chosen byte offset; print what each lane gets
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int* addr, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;
    unsigned a = base + addr[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    out[threadIdx.x] = v;
}
int main() {
    int* da; unsigned long long* dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 64 * 8);
    for (int pat = 0; pat < 3; ++pat) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) a[l] = pat == 0 ? l * 8 : pat == 1 ? (l & 15) * 64 + (l >> 4) * 8 : (l & 31) * 64 + (l >> 5) * 8;
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
        std::vector<unsigned long long> o(64);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (lane byte offset: %s)\n", pat, pat == 0 ? "l*8" : pat == 1 ? "(l&15)*64 + (l>>4)*8" : "(l&31)*64 + (l>>5)*8");
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d (off %4d = elem %4d): %4llu %4llu %4llu %4llu\n", l, a[l], a[l] / 2, o[l] & 0xffff, (o[l] >> 16) & 0xffff, (o[l] >> 32) & 0xffff, o[l] >> 48);
        }
    }
    return 0;
}

// Probe the semantics of ds_read_b64_tr_b16 on gfx950: LDS holds u16 values equal to their own element index; every lane
// passes the byte address of element (lane * 4) (i.e. its "own" 8 contiguous bytes); print what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(size_t)(&lds[l * 4]) ;                        // lane-linear 8-byte slots
    else addr = (unsigned)(size_t)(&lds[(l & 15) * 64 + (l >> 4) * 4]);             // row stride 64 elements: lane i -> row i
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (lane: 4 received element indices)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  l%02d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}

// What a store's WIDTH costs on gfx950 (round 6): the same 64 MB written by coalesced dword-, 8-byte- and 16-byte-per-lane stores, and by
// dword stores in 128-byte half-wave runs (the pattern of a contraction epilogue: lane = channel, 32 channels per row).  HIP-event time per
// launch here; `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over this binary shows whether a narrow store makes the L2 fetch the
// line it overwrites (profiles/r06_store_width.txt).   hipcc --offload-arch=gfx950 -O3 -o store_width store_width.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void st1(float* o, long n, float v) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) o[i] = v;
}
__global__ __launch_bounds__(256) void st2(float2* o, long n, float v) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n / 2; i += gridDim.x * 256L) o[i] = make_float2(v, v);
}
__global__ __launch_bounds__(256) void st4(float4* o, long n, float v) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n / 4; i += gridDim.x * 256L) o[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void st4u(float* o, long n, float v) {      // 16-byte stores at a dword-aligned (odd) address
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n / 4 - 1; i += gridDim.x * 256L) *(f4u*)(o + 4 * i + 1) = f4u{v, v, v, v};
}
// 32 lanes = one 128-byte row segment; the other half-wave writes the row D rows further (rows in groups of 2 D).  C, D powers of two:
// shifts only (a first version decoded its rows with 64-bit divisions and measured those: 2 TB/s whatever D)
__global__ __launch_bounds__(256) void st1rowsD(float* o, long n, float v, int lgC, int lgD) {
    const long rows = n >> lgC;
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, nw = gridDim.x * 4L;
    const int lgcb = lgC - 5;
    for (long r2 = wave; r2 < (rows >> 1) << lgcb; r2 += nw) {
        const long rp = r2 >> lgcb, cb = r2 & ((1 << lgcb) - 1);
        const long g = rp >> lgD, i = rp & ((1L << lgD) - 1);
        o[((((g << 1) + half) << lgD) + i << lgC) + (cb << 5) + col] = v;
    }
}
int main() {
    const long n = 16L << 20;      // 64 MB
    float* d; hipMalloc(&d, n * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"dword (256 B per wave instruction)", "8 bytes per lane", "16 bytes per lane", "16 bytes per lane, dword-aligned address"};
    for (int k = 0; k < 4; ++k) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) {
                if (k == 0) st1<<<2048, 256>>>(d, n, 1.f + it);
                if (k == 1) st2<<<2048, 256>>>((float2*)d, n, 1.f + it);
                if (k == 2) st4<<<2048, 256>>>((float4*)d, n, 1.f + it);
                if (k == 3) st4u<<<2048, 256>>>(d, n, 1.f + it);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms / 10 < best) best = ms / 10;
        }
        printf("%-48s %7.1f us per launch = %5.2f TB/s written\n", names[k], best * 1e3, n * 4 / (best * 1e-3) / 1e12);
    }
    const int Ds[6] = {0, 2, 3, 6, 10, 14};
    for (int C = 6; C <= 7; ++C)
        for (int k = 0; k < 6; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) st1rowsD<<<2048, 256>>>(d, n, 1.f + it, C, Ds[k]);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms / 10 < best) best = ms / 10;
            }
            printf("dword, 128-byte half-wave rows, C = %3d, halves %5d rows apart  %7.1f us per launch = %5.2f TB/s written\n", 1 << C, 1 << Ds[k], best * 1e3,
                   n * 4 / (best * 1e-3) / 1e12);
        }
    return 0;
}

// Micro-benchmark: sustained fp32 MFMA (v_mfma_f32_32x32x2_f32) rate of the whole chip, first with no memory traffic
// at all (the practical ceiling of any fp32 implicit-GEMM kernel on this part) and then with the side activities of a
// real tile loop added one at a time (LDS fragment reads, a block barrier per K-step, loader waves writing LDS, loader
// waves streaming from global memory).  Also reports the shader clock (s_memtime vs the 100 MHz s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o scripts/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

// mode bit0: LDS fragment reads, bit1: barrier per step, bit2: loader waves write LDS, bit3: loader waves read global
__global__ __launch_bounds__(512) void tile_loop(float* out, const float* src, unsigned long long* clk, int iters, int mode, int rnd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 384 * 20 * 4; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        smem[i] = rnd ? ((int)(h & 0xffffff) - 0x800000) * (1.f / 0x800000) : 1e-3f * (i & 127);   // random data toggles more bits
    }
    __syncthreads();
    if (wid >= 4) {
        const int lt = tid - 256, lrow = lt >> 2, lk = (lt & 3) * 4;
        f32x4 r[6];
        for (int i = 0; i < 6; ++i) r[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
        const float* p = src + ((size_t)blockIdx.x * 384 + lrow) * 16 + lk;
        for (int it = 0; it < iters; ++it) {
            if (mode & 4) {
                float* S = smem + (it & 3) * 384 * 20;
#pragma unroll
                for (int i = 0; i < 6; ++i) *(f32x4*)(S + (lrow + 64 * i) * 20 + lk) = r[i];
            }
            if (mode & 8) {
#pragma unroll
                for (int i = 0; i < 6; ++i) r[i] = *(const f32x4*)(p + (size_t)i * 64 * 16 + (size_t)(it & 63) * 256 * 384 * 16);
            }
            if (mode & 2) __syncthreads();
        }
        if (r[0][0] == 123.456f) out[1] = r[0][0];
        return;
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wm = wid >> 1, wn = wid & 1;
    const int a_lds = (wm * 128 + (lane & 31)) * 20 + (lane >> 5) * 4;
    const int b_lds = 256 * 20 + (wn * 64 + (lane & 31)) * 20 + (lane >> 5) * 4;
    f32x4 af[2][4], bf[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[c][mi] = *(const f32x4*)(smem + a_lds + mi * 32 * 20 + 8 * c);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[c][ni] = *(const f32x4*)(smem + b_lds + ni * 32 * 20 + 8 * c);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const float* St = smem + (it & 3) * 384 * 20;
        const float* Sn = smem + ((it + 1) & 3) * 384 * 20;
        if (mode & 1) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[1][mi] = *(const f32x4*)(St + a_lds + mi * 32 * 20 + 8);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[1][ni] = *(const f32x4*)(St + b_lds + ni * 32 * 20 + 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][mi][j], bf[0][ni][j], acc[mi * 2 + ni], 0, 0, 0);
        if (mode & 1) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[0][mi] = *(const f32x4*)(Sn + a_lds + mi * 32 * 20);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[0][ni] = *(const f32x4*)(Sn + b_lds + ni * 32 * 20);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][mi][j], bf[1][ni][j], acc[mi * 2 + ni], 0, 0, 0);
        if (mode & 2) __syncthreads();
    }
    if (blockIdx.x == 17 && tid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const int rnd = argc > 1 ? atoi(argv[1]) : 0;
    printf("operand data: %s\n", rnd ? "pseudo-random in [-1,1)" : "smooth ramp");
    float *out, *src; unsigned long long* clk;
    const size_t src_floats = (size_t)64 * 256 * 384 * 16;     // 100 MB: streams through L2/MALL
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&src, src_floats * 4)); CK(hipMalloc(&clk, 16));
    CK(hipMemset(src, 0, src_floats * 4));
    const size_t lds = 384 * 20 * 4 * sizeof(float);
    CK(hipFuncSetAttribute((const void*)tile_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"pure MFMA", "+LDS fragment reads", "+barrier/step", "+loader ds_write", "+loader global loads"};
    const int modes[] = {0, 1, 3, 7, 15};
    const int iters = 20000;
    for (int v = 0; v < 5; ++v) {
        const int threads = (modes[v] & 12) ? 512 : 256;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(tile_loop, dim3(256), dim3(threads), lds, 0, out, src, clk, iters, modes[v], rnd);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double flops = 256.0 * 4 * (double)iters * 64 * (2.0 * 32 * 32 * 2);
            if (rep == 1)
                printf("%-24s %8.3f ms  %6.1f TFLOP/s   shader clock %.0f MHz   cycles/step %.0f (ideal 4096)\n", names[v], ms,
                       flops / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / iters);
        }
    }
    return 0;
}

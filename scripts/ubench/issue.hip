// Micro-benchmark (measurement only, not part of the library): what does one non-MFMA instruction cost a wave that keeps the fp32
// matrix pipe busy with v_mfma_f32_32x32x2_f32 (64 pipe cycles each), at one wave per SIMD -- and does a SECOND wave on the same
// SIMD absorb that cost?   usage: issue            (prints cycles per MFMA slot for every (kind, count, placement))
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 none, 1 v_pk_add_f32, 2 ds_write_b64, 3 global_load_dword (L2 hit), 4 v_add_f32, 5 ds_read_b128, 6 ds_write_b128, 7 v_cndmask,
//       8 / 9 / 10 buffer_load_dword / x2 / x4 (offen + SGPR soffset), 11 ds_read2_b32, 12 ds_write_b64 on consecutive lanes, 13 s_add (SALU)
template <int KIND, int K>
__device__ __forceinline__ void side_ops(f32x2* r, float* lds, const float* g, int lane, float& sink, __amdgpu_buffer_rsrc_t rs, int lane0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r[k & 7]) : "v"(r[(k + 1) & 7]), "v"(r[(k + 2) & 7]));
        if (KIND == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(lane * 16), "v"(r[k & 7]), "n"((k & 7) * 2048) : "memory");
        if (KIND == 3) { float v; asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(v) : "v"(g + lane), "n"((k & 7) * 256) : "memory"); sink += 0.f * v; }
        if (KIND == 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[k & 7].x) : "v"(r[(k + 1) & 7].x), "v"(r[(k + 2) & 7].y));
        if (KIND == 5) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lane * 16), "n"((k & 7) * 2048) : "memory"); }
        if (KIND == 6) { f32x4 v = {r[0].x, r[1].x, r[2].x, r[3].x}; asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lane * 16), "v"(v), "n"((k & 7) * 2048) : "memory"); }
        if (KIND == 8) { float v; asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(lane * 4), "s"(rs), "s"(k * 256), "n"((k & 7) * 256) : "memory"); }
        if (KIND == 9) { f32x2 v; asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(lane * 8), "s"(rs), "s"(k * 512), "n"((k & 7) * 512) : "memory"); }
        if (KIND == 10) { f32x4 v; asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(lane * 16), "s"(rs), "s"(k * 1024), "n"((k & 3) * 1024) : "memory"); }
        if (KIND == 11) { f32x2 v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(lane * 4), "n"((k & 7) * 8), "n"((k & 7) * 8 + 64) : "memory"); }
        if (KIND == 12) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(lane * 8), "v"(r[k & 7]), "n"((k & 7) * 2048) : "memory");
        if (KIND == 13) { int t; asm volatile("s_add_i32 %0, %1, %2" : "=s"(t) : "s"(k), "s"(lane0) : "scc"); }
        if (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[k & 7].x) : "v"(r[(k + 1) & 7].x), "v"(r[(k + 2) & 7].y));
    }
}

// PLACE 0: the MFMA wave issues the side instructions itself (256 threads, 1 wave / SIMD)
// PLACE 1: 512 threads: waves 0-3 only multiply, waves 4-7 only issue the side instructions (2 waves / SIMD)
template <int KIND, int K, int PLACE>
__global__ __launch_bounds__(PLACE ? 512 : 256) void bench(unsigned long long* out, const float* g, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f32x2 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f32x2{(float)lane, (float)i};
    float sink = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 1 << 20, 0x00020000);
    unsigned long long t0 = 0, t1 = 0;
    if (PLACE == 0 || wid < 4) {
        constexpr int NA = PLACE ? 7 : 16;          // two waves per SIMD: 256 registers each
        f32x16 acc[NA];
#pragma unroll
        for (int p = 0; p < NA; ++p)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[p][q] = 0.f;
        const float a = (float)lane, b = 1.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                acc[p % NA] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p % NA], 0, 0, 0);
                if (PLACE == 0) side_ops<KIND, K>(r, lds, g, lane, sink, rs, __builtin_amdgcn_readfirstlane(wid));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KIND == 3 || KIND == 5 || KIND >= 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NA; ++p) s += acc[p][0] + acc[p][15];
        sink += s;
    } else {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                side_ops<KIND, K>(r, lds, g, lane, sink, rs, __builtin_amdgcn_readfirstlane(wid));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KIND == 3 || KIND == 5 || KIND >= 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    sink += r[0].x + r[1].y + r[2].x + r[3].y + r[4].x + r[5].y + r[6].x + r[7].y;
    if (lane == 0 && blockIdx.x == 0) out[wid] = t1 - t0;
    if (sink == 12345.678f) out[15] = 1;
}

template <int KIND, int K, int PLACE>
static void run(unsigned long long* dout, const float* g, const char* name) {
    const int iters = 2000;
    const size_t lds = 96 * 1024;       // one block per CU
    (void)hipFuncSetAttribute((const void*)bench<KIND, K, PLACE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bench<KIND, K, PLACE>), dim3(256), dim3(PLACE ? 512 : 256), lds, 0, dout, g, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[16];
    (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-14s K=%d %s  cycles per MFMA slot: mfma wave %.1f", name, K, PLACE ? "second wave " : "same wave   ", (double)h[0] / (iters * 16.0));
    if (PLACE) printf("   side wave %.1f", (double)h[4] / (iters * 16.0));
    printf("\n");
}
#define ALLK(KIND, NAME)                                                                         \
    run<KIND, 1, 0>(dout, g, NAME); run<KIND, 2, 0>(dout, g, NAME); run<KIND, 4, 0>(dout, g, NAME); run<KIND, 8, 0>(dout, g, NAME); \
    run<KIND, 1, 1>(dout, g, NAME); run<KIND, 2, 1>(dout, g, NAME); run<KIND, 4, 1>(dout, g, NAME); run<KIND, 8, 1>(dout, g, NAME);
int main() {
    unsigned long long* dout; float* g;
    (void)hipMalloc((void**)&dout, 128); (void)hipMalloc((void**)&g, 1 << 20);
    (void)hipMemset(g, 0, 1 << 20);
    run<0, 0, 0>(dout, g, "none"); run<0, 0, 1>(dout, g, "none");
    ALLK(1, "v_pk_add_f32") ALLK(13, "s_add_i32") ALLK(2, "ds_write_b64") ALLK(12, "ds_write_b64c") ALLK(6, "ds_write_b128") ALLK(5, "ds_read_b128") ALLK(11, "ds_read2_b32")
    ALLK(3, "global_load") ALLK(8, "buffer_load_x1") ALLK(9, "buffer_load_x2") ALLK(10, "buffer_load_x4")
    return 0;
}

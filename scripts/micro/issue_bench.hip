// issue_bench.hip -- vector-instruction issue cost on gfx950, per instruction class and waves per SIMD.
// Each wave runs REPS x 16 independent instructions of one kind (8 register chains, inline asm so nothing is
// folded) and stamps s_memtime around the loop; cost = cycles / instructions / waves per SIMD.
// Diagnostic (scripts/micro/run_issue_bench.sh); the weights bench.py's roofline_issue uses come from its output.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 8192;

#define CHAIN8(INSTR)                                         \
    asm volatile(INSTR(0) "\n\t" INSTR(1) "\n\t" INSTR(2) "\n\t" INSTR(3) "\n\t" \
                 INSTR(4) "\n\t" INSTR(5) "\n\t" INSTR(6) "\n\t" INSTR(7)        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b), "v"(c) : "vcc")

// operand strings: %0..%7 chains, %8 = b, %9 = c
#define I_FMA_F64(i)   "v_fma_f64 %" #i ", %" #i ", %8, %9"
#define I_MUL_F64(i)   "v_mul_f64 %" #i ", %" #i ", %8"
#define I_ADD_F64(i)   "v_add_f64 %" #i ", %" #i ", %8"
#define I_MAX_F64(i)   "v_max_f64 %" #i ", %" #i ", %8"
#define I_RCP_F64(i)   "v_rcp_f64 %" #i ", %" #i
#define I_RSQ_F64(i)   "v_rsq_f64 %" #i ", %" #i
#define I_CMP_F64(i)   "v_cmp_gt_f64 vcc, %" #i ", %8"
#define I_CVT_F32_F64(i) "v_cvt_f32_f64 %" #i ", %8"

extern __shared__ char s_pad[];          // dynamic LDS only bounds the workgroups per CU (see main)

template <int op>
__global__ void __launch_bounds__(1024) k_f64(unsigned long long *out, double b_in, double c_in)
{
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = b_in, c = c_in;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if constexpr (op == 0) { CHAIN8(I_FMA_F64); CHAIN8(I_FMA_F64); }
        if constexpr (op == 1) { CHAIN8(I_MUL_F64); CHAIN8(I_MUL_F64); }
        if constexpr (op == 2) { CHAIN8(I_ADD_F64); CHAIN8(I_ADD_F64); }
        if constexpr (op == 3) { CHAIN8(I_MAX_F64); CHAIN8(I_MAX_F64); }
        if constexpr (op == 4) { CHAIN8(I_RCP_F64); CHAIN8(I_RCP_F64); }
        if constexpr (op == 5) { CHAIN8(I_RSQ_F64); CHAIN8(I_RSQ_F64); }
        if constexpr (op == 6) { CHAIN8(I_CMP_F64); CHAIN8(I_CMP_F64); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = 0;
}

#define I_FMA_F32(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9"
#define I_MUL_F32(i)   "v_mul_f32 %" #i ", %" #i ", %8"
#define I_ADD_F32(i)   "v_add_f32 %" #i ", %" #i ", %8"
#define I_MIN3_F32(i)  "v_min3_f32 %" #i ", %" #i ", %8, %9"
#define I_RCP_F32(i)   "v_rcp_f32 %" #i ", %" #i
#define I_RSQ_F32(i)   "v_rsq_f32 %" #i ", %" #i
#define I_SQRT_F32(i)  "v_sqrt_f32 %" #i ", %" #i
#define I_CMP_F32(i)   "v_cmp_gt_f32 vcc, %" #i ", %8"
#define I_CNDMASK(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define I_AND_B32(i)   "v_and_b32 %" #i ", %" #i ", %8"
#define I_LSHL_OR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 1, %8"
#define I_MOV_B32(i)   "v_mov_b32 %" #i ", %8"
#define I_CVT_F64_F32(i) "v_cvt_f32_i32 %" #i ", %" #i

template <int op>
__global__ void __launch_bounds__(1024) k_f32(unsigned long long *out, float b_in, float c_in)
{
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = b_in, c = c_in;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if constexpr (op == 0) { CHAIN8(I_FMA_F32); CHAIN8(I_FMA_F32); }
        if constexpr (op == 1) { CHAIN8(I_MUL_F32); CHAIN8(I_MUL_F32); }
        if constexpr (op == 2) { CHAIN8(I_ADD_F32); CHAIN8(I_ADD_F32); }
        if constexpr (op == 3) { CHAIN8(I_MIN3_F32); CHAIN8(I_MIN3_F32); }
        if constexpr (op == 4) { CHAIN8(I_RCP_F32); CHAIN8(I_RCP_F32); }
        if constexpr (op == 5) { CHAIN8(I_RSQ_F32); CHAIN8(I_RSQ_F32); }
        if constexpr (op == 6) { CHAIN8(I_SQRT_F32); CHAIN8(I_SQRT_F32); }
        if constexpr (op == 7) { CHAIN8(I_CMP_F32); CHAIN8(I_CMP_F32); }
        if constexpr (op == 8) { CHAIN8(I_CNDMASK); CHAIN8(I_CNDMASK); }
        if constexpr (op == 9) { CHAIN8(I_AND_B32); CHAIN8(I_AND_B32); }
        if constexpr (op == 10) { CHAIN8(I_LSHL_OR); CHAIN8(I_LSHL_OR); }
        if constexpr (op == 11) { CHAIN8(I_MOV_B32); CHAIN8(I_MOV_B32); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;
}

// packed float32 and cross-lane: 64-bit register pairs
typedef float v2f __attribute__((ext_vector_type(2)));
#define I_PK_FMA(i)    "v_pk_fma_f32 %" #i ", %" #i ", %8, %9"
#define I_PK_MUL(i)    "v_pk_mul_f32 %" #i ", %" #i ", %8"
#define I_PK_ADD(i)    "v_pk_add_f32 %" #i ", %" #i ", %8"
#define I_MOV_B64(i)   "v_mov_b64 %" #i ", %8"

template <int op>
__global__ void __launch_bounds__(1024) k_pk(unsigned long long *out, float b_in, float c_in)
{
    v2f a0 = { threadIdx.x * 1e-3f + 1.0f, 2.0f }, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f b = { b_in, b_in }, c = { c_in, c_in };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if constexpr (op == 0) { CHAIN8(I_PK_FMA); CHAIN8(I_PK_FMA); }
        if constexpr (op == 1) { CHAIN8(I_PK_MUL); CHAIN8(I_PK_MUL); }
        if constexpr (op == 2) { CHAIN8(I_PK_ADD); CHAIN8(I_PK_ADD); }
        if constexpr (op == 3) { CHAIN8(I_MOV_B64); CHAIN8(I_MOV_B64); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 12345.678f) out[0] = 0;
}

// v_readlane into an SGPR (the per-step values of k_evaluate) and mixes
template <int op>
__global__ void __launch_bounds__(1024) k_lane(unsigned long long *out, int b_in)
{
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int s0, s1, s2, s3;
            asm volatile("v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 5\n\tv_readlane_b32 %2, %6, 7\n\tv_readlane_b32 %3, %7, 9"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            acc += s0 ^ s1 ^ s2 ^ s3;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (acc == 123456789) out[0] = 0;
}

template <class K, class... A>
static void launch(K kern, int grid, int block, size_t lds, A... args)
{
    HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, args...);
    HIP_OK(hipGetLastError());
}

int main()
{
    int n_cu = 0;
    HIP_OK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned long long *d_out;
    HIP_OK(hipMalloc(&d_out, sizeof(unsigned long long) * 65536));
    std::vector<unsigned long long> h(65536);
    hipEvent_t ev0, ev1;
    HIP_OK(hipEventCreate(&ev0)); HIP_OK(hipEventCreate(&ev1));
    auto report = [&](const char *name, int wps, int grid, int block, double n_instr) {
        HIP_OK(hipEventRecord(ev1, 0));
        HIP_OK(hipDeviceSynchronize());
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ev0, ev1));
        const int n_waves = grid * block / 64;
        HIP_OK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * n_waves, hipMemcpyDeviceToHost));
        double sum = 0, mx = 0, mn = 1e30;
        for (int i = 0; i < n_waves; ++i) { sum += (double)h[i]; mx = h[i] > mx ? (double)h[i] : mx; mn = h[i] < mn ? (double)h[i] : mn; }
        const double avg = sum / n_waves;
        // the same from the host's clock: launch time x 2.4 GHz over the instructions each SIMD issued (an upper bound:
        // it holds the launch overhead and assumes the peak clock)
        const double wall = (double)ms * 1e-3 * 2.4e9 / (n_instr * (double)n_waves / (4.0 * n_cu));
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_simd\": %.3f, \"wall_cycles_at_2p4GHz\": %.3f, \"launch_ms\": %.4f, \"wave_cycles_avg\": %.0f, \"wave_cycles_min\": %.0f, \"wave_cycles_max\": %.0f}\n",
               name, wps, avg / n_instr / wps, wall, ms, avg, mn, mx);
    };
    const double N = (double)REPS * 16.0;
    // Waves per SIMD are pinned by LDS, not by luck of placement: a workgroup of 256 * w threads (w <= 4) that asks for
    // 81 KB of the CU's 160 KB can only sit alone on its CU, so each of its SIMDs holds exactly w waves while it runs
    // (n_cu workgroups: one round).  w = 8: two workgroups of 1024 threads with 70 KB each per CU, 2 * n_cu of them.
    const int W[] = { 1, 2, 4, 8 };
    for (int wi = 0; wi < 4; ++wi) {
        const int wps = W[wi];
        const int block = wps == 8 ? 1024 : 256 * wps, grid = wps == 8 ? 2 * n_cu : n_cu;
        const size_t lds = wps == 8 ? 70 * 1024 : 81 * 1024;
#define RUN(KERN, NAME, ...) do { HIP_OK(hipEventRecord(ev0, 0)); launch(KERN, grid, block, lds, d_out, __VA_ARGS__); report(NAME, wps, grid, block, N); } while (0)
        RUN(k_f64<0>, "v_fma_f64", 1.0000001, 1e-9); RUN(k_f64<1>, "v_mul_f64", 1.0000001, 1e-9);
        RUN(k_f64<2>, "v_add_f64", 1.0000001, 1e-9); RUN(k_f64<3>, "v_max_f64", 1.0000001, 1e-9);
        RUN(k_f64<4>, "v_rcp_f64", 1.0000001, 1e-9); RUN(k_f64<5>, "v_rsq_f64", 1.0000001, 1e-9);
        RUN(k_f64<6>, "v_cmp_f64", 1.0000001, 1e-9);
        RUN(k_f32<0>, "v_fma_f32", 1.0000001f, 1e-9f); RUN(k_f32<1>, "v_mul_f32", 1.0000001f, 1e-9f);
        RUN(k_f32<2>, "v_add_f32", 1.0000001f, 1e-9f); RUN(k_f32<3>, "v_min3_f32", 1.0000001f, 1e-9f);
        RUN(k_f32<4>, "v_rcp_f32", 1.0000001f, 1e-9f); RUN(k_f32<5>, "v_rsq_f32", 1.0000001f, 1e-9f);
        RUN(k_f32<6>, "v_sqrt_f32", 1.0000001f, 1e-9f); RUN(k_f32<7>, "v_cmp_f32", 1.0000001f, 1e-9f);
        RUN(k_f32<8>, "v_cndmask_b32", 1.0000001f, 1e-9f); RUN(k_f32<9>, "v_and_b32", 1.0000001f, 1e-9f);
        RUN(k_f32<10>, "v_lshl_or_b32", 1.0000001f, 1e-9f); RUN(k_f32<11>, "v_mov_b32", 1.0000001f, 1e-9f);
        RUN(k_pk<0>, "v_pk_fma_f32", 1.0000001f, 1e-9f); RUN(k_pk<1>, "v_pk_mul_f32", 1.0000001f, 1e-9f);
        RUN(k_pk<2>, "v_pk_add_f32", 1.0000001f, 1e-9f); RUN(k_pk<3>, "v_mov_b64", 1.0000001f, 1e-9f);
        RUN(k_lane<0>, "v_readlane_b32", 1);
#undef RUN
    }
    HIP_OK(hipFree(d_out));
    return 0;
}

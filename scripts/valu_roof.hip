// valu_roof.hip -- issue cost of wave64 VALU instructions on gfx950, per instruction class: the denominator of the step
// kernel's VALU roof (bench.py roofline.issue_cycles_per_class).
//
//   hipcc --offload-arch=gfx950 -O2 scripts/valu_roof.hip -o scripts/_build/valu_roof && scripts/_build/valu_roof > profiles/rNN_valu_roof.json
//
// Every kernel runs ITERS trips of an asm block holding 16 instructions of one class on 8 independent register chains
// (so that a lone wave is limited by issue, not by dependency latency, wherever the pipe allows it), bracketed by
// s_memtime.  Launched as 256 * k workgroups of 256 threads: k waves on every SIMD of the 256 CUs (k = 1, 2, 4).  Reported
// per class: SIMD cycles per wave64 instruction = k-wave elapsed cycles / (k * instructions per wave), from the in-kernel
// cycle counter (median over waves) and, as a cross-check, from HIP-event wall time at the clock the counter implies.
// A `dep` variant (one chain) gives the dependent-issue latency; the `mix` rows pair fp64 FMAs with scalar / LDS / 32-bit
// instructions to see which kinds share the VALU's issue slots.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kIters = 1024;

// 16 instructions on chains 0..7 (twice round): OP3(d, a, b) must be one instruction writing d
#define BLOCK16(OP)                                                                                                  \
    OP("%0", "%8", "%9") OP("%1", "%8", "%9") OP("%2", "%8", "%9") OP("%3", "%8", "%9") OP("%4", "%8", "%9")          \
    OP("%5", "%8", "%9") OP("%6", "%8", "%9") OP("%7", "%8", "%9") OP("%0", "%8", "%9") OP("%1", "%8", "%9")          \
    OP("%2", "%8", "%9") OP("%3", "%8", "%9") OP("%4", "%8", "%9") OP("%5", "%8", "%9") OP("%6", "%8", "%9")          \
    OP("%7", "%8", "%9")

#define KERNEL(NAME, TYPE, INIT, OP)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, TYPE seed) {                                 \
        TYPE a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;                  \
        TYPE x = INIT, y = seed;                                                                                      \
        asm volatile("" : "+v"(x), "+v"(y));                                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kIters; ++i)                                                                              \
            asm volatile(BLOCK16(OP)                                                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(x), "v"(y)                                                                             \
                         : "vcc");                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                     \
        if ((threadIdx.x & 63u) == 0u) { unsigned long long* o_ = out + 4 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); o_[0] = t0; o_[1] = t1; o_[2] = __builtin_amdgcn_s_getreg(63492); o_[3] = __builtin_amdgcn_s_getreg(63508); }                    \
    }

#define OP_FMA64(d, a, b) "v_fma_f64 " d ", " d ", " a ", " b "\n\t"
#define OP_ADD64(d, a, b) "v_add_f64 " d ", " d ", " a "\n\t"
#define OP_MUL64(d, a, b) "v_mul_f64 " d ", " d ", " a "\n\t"
#define OP_MAX64(d, a, b) "v_max_f64 " d ", " d ", " a "\n\t"
#define OP_RCP64(d, a, b) "v_rcp_f64 " d ", " d "\n\t"
#define OP_SQRT64(d, a, b) "v_sqrt_f64 " d ", " d "\n\t"
#define OP_CMP64(d, a, b) "v_cmp_lt_f64 vcc, " d ", " a "\n\t"
#define OP_LSHL64(d, a, b) "v_lshlrev_b64 " d ", 1, " d "\n\t"
#define OP_MOV64(d, a, b) "v_mov_b64 " d ", " a "\n\t"
#define OP_FMA32(d, a, b) "v_fma_f32 " d ", " d ", " a ", " b "\n\t"
#define OP_ADD32(d, a, b) "v_add_f32 " d ", " d ", " a "\n\t"
#define OP_MUL32(d, a, b) "v_mul_f32 " d ", " d ", " a "\n\t"
#define OP_MAX32(d, a, b) "v_max_f32 " d ", " d ", " a "\n\t"
#define OP_MAX3_32(d, a, b) "v_max3_f32 " d ", " d ", " a ", " b "\n\t"
#define OP_RCP32(d, a, b) "v_rcp_f32 " d ", " d "\n\t"
#define OP_CMP32(d, a, b) "v_cmp_lt_f32 vcc, " d ", " a "\n\t"
#define OP_CNDMASK(d, a, b) "v_cndmask_b32 " d ", " d ", " a ", vcc\n\t"
#define OP_MOV32(d, a, b) "v_mov_b32 " d ", " a "\n\t"
#define OP_ADDU32(d, a, b) "v_add_u32 " d ", " d ", " a "\n\t"
#define OP_AND32(d, a, b) "v_and_b32 " d ", " d ", " a "\n\t"
#define OP_LSHL32(d, a, b) "v_lshlrev_b32 " d ", 1, " d "\n\t"
#define OP_MULLO(d, a, b) "v_mul_lo_u32 " d ", " d ", " a "\n\t"
#define OP_ADDC(d, a, b) "v_addc_co_u32 " d ", vcc, " d ", " d ", vcc\n\t"
#define OP_CVT_F64_F32(d, a, b) "v_cvt_f64_f32 " d ", v40\n\t"
#define OP_CVT_F32_F64(d, a, b) "v_cvt_f32_f64 v43, " d "\n\t"
#define OP_CVT_I32_F64(d, a, b) "v_cvt_i32_f64 v43, " d "\n\t"
#define OP_PKFMA(d, a, b) "v_pk_fma_f32 " d ", " d ", " a ", " b "\n\t"
#define OP_PKADD(d, a, b) "v_pk_add_f32 " d ", " d ", " a "\n\t"
#define OP_PKMUL(d, a, b) "v_pk_mul_f32 " d ", " d ", " a "\n\t"
// mixes: one fp64 FMA + one instruction of another kind
#define OP_MIX_SALU(d, a, b) "v_fma_f64 " d ", " d ", " a ", " b "\n\ts_add_u32 s20, s20, 1\n\t"
#define OP_MIX_SALU2(d, a, b) "v_fma_f64 " d ", " d ", " a ", " b "\n\ts_add_u32 s20, s20, 1\n\ts_and_b32 s21, s21, s20\n\t"
#define OP_MIX_FMA32(d, a, b) "v_fma_f64 " d ", " d ", " a ", " b "\n\tv_add_u32 v40, v40, v41\n\t"
#define OP_MIX_NOP(d, a, b) "v_fma_f64 " d ", " d ", " a ", " b "\n\ts_nop 0\n\t"
#define OP_MIX_LDS(d, a, b) "v_fma_f64 " d ", " d ", " a ", " b "\n\tds_read_b32 v40, v42\n\t"
#define OP_SALU_ONLY(d, a, b) "s_add_u32 s20, s20, 1\n\t"

// (v_cndmask reads a lane mask: VCC as left by whatever ran before / VCC set to a pattern / an SGPR pair / VCC written by a
// v_cmp right before it -- the usual select idiom)
#define KERNEL_PRE(NAME, TYPE, INIT, PRE, OP)                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, TYPE seed) {                                 \
        TYPE a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;                  \
        TYPE x = INIT, y = seed;                                                                                      \
        asm volatile("" : "+v"(x), "+v"(y));                                                                          \
        asm volatile(PRE ::: "vcc", "s20", "s21");                                                                    \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kIters; ++i)                                                                              \
            asm volatile(BLOCK16(OP)                                                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(x), "v"(y)                                                                             \
                         : "vcc", "s20", "s21");                                                                      \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                     \
        if ((threadIdx.x & 63u) == 0u) { unsigned long long* o_ = out + 4 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); o_[0] = t0; o_[1] = t1; o_[2] = __builtin_amdgcn_s_getreg(63492); o_[3] = __builtin_amdgcn_s_getreg(63508); } \
    }
#define OP_CNDMASK_S(d, a, b) "v_cndmask_b32_e64 " d ", " d ", " a ", s[20:21]\n\t"
#define OP_CMP_CNDMASK(d, a, b) "v_cmp_lt_u32 vcc, " d ", " a "\n\tv_cndmask_b32 " d ", " d ", " a ", vcc\n\t"
#define OP_CMP_CNDMASK_S(d, a, b) "v_cmp_lt_u32_e64 s[20:21], " d ", " a "\n\tv_cndmask_b32_e64 " d ", " d ", " a ", s[20:21]\n\t"
#define OP_CMP64_S(d, a, b) "v_cmp_lt_u32_e64 s[20:21], " d ", " a "\n\t"
#define OP_MINU32(d, a, b) "v_min_u32 " d ", " d ", " a "\n\t"
#define OP_MAD_U32(d, a, b) "v_mad_u32_u24 " d ", " d ", " a ", " b "\n\t"
#define OP_OR3(d, a, b) "v_or3_b32 " d ", " d ", " a ", " b "\n\t"
#define OP_LSHLADD(d, a, b) "v_lshl_add_u32 " d ", " d ", 2, " a "\n\t"
#define OP_BFE(d, a, b) "v_bfe_u32 " d ", " d ", 3, 5\n\t"
#define OP_READLANE(d, a, b) "v_readlane_b32 s20, " d ", 3\n\t"
#define OP_WRITELANE(d, a, b) "v_writelane_b32 " d ", s20, 3\n\t"
#define OP_CMP_2CND(d, a, b) "v_cmp_lt_u32 vcc, " d ", " a "\n\tv_cndmask_b32 " d ", " d ", " a ", vcc\n\tv_cndmask_b32 " d ", " d ", " b ", vcc\n\t"
#define OP_CMP_2CND_S(d, a, b) "v_cmp_lt_u32_e64 s[20:21], " d ", " a "\n\tv_cndmask_b32_e64 " d ", " d ", " a ", s[20:21]\n\tv_cndmask_b32_e64 " d ", " d ", " b ", s[20:21]\n\t"
#define OP_CMP64_2CND(d, a, b) "v_cmp_lt_f64 vcc, %8, %9\n\tv_cndmask_b32 " d ", " d ", " a ", vcc\n\tv_cndmask_b32 " d ", " d ", " b ", vcc\n\t"
KERNEL_PRE(k_cmp_2cnd, unsigned, 3u, "s_mov_b64 vcc, 0", OP_CMP_2CND)
KERNEL_PRE(k_cmp_2cnd_s, unsigned, 3u, "s_mov_b64 s[20:21], 0", OP_CMP_2CND_S)
KERNEL_PRE(k_cndmask_vccset, unsigned, 3u, "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555", OP_CNDMASK)
KERNEL_PRE(k_cndmask_s, unsigned, 3u, "s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555", OP_CNDMASK_S)
KERNEL_PRE(k_cmp_cndmask, unsigned, 3u, "s_mov_b64 vcc, 0", OP_CMP_CNDMASK)
KERNEL_PRE(k_cmp_cndmask_s, unsigned, 3u, "s_mov_b64 s[20:21], 0", OP_CMP_CNDMASK_S)
KERNEL_PRE(k_cmp_s, unsigned, 3u, "s_mov_b64 s[20:21], 0", OP_CMP64_S)
KERNEL_PRE(k_minu32, unsigned, 3u, "s_mov_b64 vcc, 0", OP_MINU32)
KERNEL_PRE(k_mad_u32, unsigned, 3u, "s_mov_b64 vcc, 0", OP_MAD_U32)
KERNEL_PRE(k_or3, unsigned, 3u, "s_mov_b64 vcc, 0", OP_OR3)
KERNEL_PRE(k_lshladd, unsigned, 3u, "s_mov_b64 vcc, 0", OP_LSHLADD)
KERNEL_PRE(k_bfe, unsigned, 3u, "s_mov_b64 vcc, 0", OP_BFE)
KERNEL_PRE(k_readlane, unsigned, 3u, "s_mov_b32 s20, 0", OP_READLANE)
KERNEL_PRE(k_writelane, unsigned, 3u, "s_mov_b32 s20, 7", OP_WRITELANE)
KERNEL(k_fma64, double, 1.0000001, OP_FMA64)
KERNEL(k_add64, double, 1.0000001, OP_ADD64)
KERNEL(k_mul64, double, 1.0000001, OP_MUL64)
KERNEL(k_max64, double, 1.0000001, OP_MAX64)
KERNEL(k_rcp64, double, 1.0000001, OP_RCP64)
KERNEL(k_sqrt64, double, 1.0000001, OP_SQRT64)
KERNEL(k_cmp64, double, 1.0000001, OP_CMP64)
KERNEL(k_lshl64, unsigned long long, 3ull, OP_LSHL64)
KERNEL(k_mov64, unsigned long long, 3ull, OP_MOV64)
KERNEL(k_fma32, float, 1.0001f, OP_FMA32)
KERNEL(k_add32, float, 1.0001f, OP_ADD32)
KERNEL(k_mul32, float, 1.0001f, OP_MUL32)
KERNEL(k_max32, float, 1.0001f, OP_MAX32)
KERNEL(k_max3_32, float, 1.0001f, OP_MAX3_32)
KERNEL(k_rcp32, float, 1.0001f, OP_RCP32)
KERNEL(k_cmp32, float, 1.0001f, OP_CMP32)
KERNEL(k_cndmask, unsigned, 3u, OP_CNDMASK)
KERNEL(k_mov32, unsigned, 3u, OP_MOV32)
KERNEL(k_addu32, unsigned, 3u, OP_ADDU32)
KERNEL(k_and32, unsigned, 3u, OP_AND32)
KERNEL(k_lshl32, unsigned, 3u, OP_LSHL32)
KERNEL(k_mullo, unsigned, 3u, OP_MULLO)
KERNEL(k_addc, unsigned, 3u, OP_ADDC)
KERNEL(k_pkfma, double, 1.0000001, OP_PKFMA)
KERNEL(k_pkadd, double, 1.0000001, OP_PKADD)
KERNEL(k_pkmul, double, 1.0000001, OP_PKMUL)

// mixes clobber fixed registers outside the operand list: declared in a wrapper macro of their own
#define KERNEL_MIX(NAME, OP)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, double seed) {                               \
        __shared__ unsigned lds[256];                                                                                 \
        lds[threadIdx.x] = threadIdx.x;                                                                               \
        __syncthreads();                                                                                              \
        double a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;                \
        double x = 1.0000001, y = seed;                                                                               \
        asm volatile("" : "+v"(x), "+v"(y));                                                                          \
        asm volatile("v_mov_b32 v40, 1\n\tv_mov_b32 v41, 3\n\tv_lshlrev_b32 v42, 2, %0\n\ts_mov_b32 s20, 0\n\ts_mov_b32 s21, 0" ::"v"(threadIdx.x) \
                     : "v40", "v41", "v42", "v43", "s20", "s21");                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kIters; ++i)                                                                              \
            asm volatile(BLOCK16(OP) "s_waitcnt lgkmcnt(0)\n\t"                                                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(x), "v"(y)                                                                             \
                         : "vcc", "v40", "v41", "v42", "v43", "s20", "s21", "scc", "memory");                             \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));                     \
        if ((threadIdx.x & 63u) == 0u) { unsigned long long* o_ = out + 4 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); o_[0] = t0; o_[1] = t1; o_[2] = __builtin_amdgcn_s_getreg(63492); o_[3] = __builtin_amdgcn_s_getreg(63508); }                    \
    }
KERNEL_MIX(k_mix_fma64_salu, OP_MIX_SALU)
KERNEL_MIX(k_mix_fma64_2salu, OP_MIX_SALU2)
KERNEL_MIX(k_mix_fma64_valu32, OP_MIX_FMA32)
KERNEL_MIX(k_mix_fma64_nop, OP_MIX_NOP)
KERNEL_MIX(k_mix_fma64_lds, OP_MIX_LDS)
KERNEL_MIX(k_salu_only, OP_SALU_ONLY)
KERNEL_MIX(k_cvt_f64_f32, OP_CVT_F64_F32)
KERNEL_MIX(k_cvt_f32_f64, OP_CVT_F32_F64)
KERNEL_MIX(k_cvt_i32_f64, OP_CVT_I32_F64)

// dependent chains: ONE register, 16 instructions in a row each waiting for the last
#define DEP16(OP) OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") \
    OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2")               \
    OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2") OP("%0", "%1", "%2")               \
    OP("%0", "%1", "%2")
#define KERNEL_DEP(NAME, TYPE, INIT, OP)                                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, TYPE seed) {                                 \
        TYPE a0 = seed, x = INIT, y = seed;                                                                           \
        asm volatile("" : "+v"(x), "+v"(y));                                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kIters; ++i) asm volatile(DEP16(OP) : "+v"(a0) : "v"(x), "v"(y) : "vcc");                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        asm volatile("" ::"v"(a0));                                                                                   \
        if ((threadIdx.x & 63u) == 0u) { unsigned long long* o_ = out + 4 * (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); o_[0] = t0; o_[1] = t1; o_[2] = __builtin_amdgcn_s_getreg(63492); o_[3] = __builtin_amdgcn_s_getreg(63508); }                    \
    }
KERNEL_DEP(k_dep_fma64, double, 1.0000001, OP_FMA64)
KERNEL_DEP(k_dep_add64, double, 1.0000001, OP_ADD64)
KERNEL_DEP(k_dep_fma32, float, 1.0001f, OP_FMA32)
KERNEL_DEP(k_dep_addu32, unsigned, 3u, OP_ADDU32)
KERNEL_DEP(k_dep_rcp64, double, 1.0000001, OP_RCP64)

struct Row {
    std::string name, klass;
    int per_block;   // instructions of the measured class per asm block
    int extra;       // other instructions in the block (mixes)
    void (*launch)(int blocks, unsigned long long* out, hipStream_t s);
};

template <class T, void (*K)(unsigned long long*, T)>
void launch_t(int blocks, unsigned long long* out, hipStream_t s) {
    hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, s, out, (T)1);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_out;
    const int max_waves = cus * 4 * 8;
    CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * 4 * max_waves));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
#define ROW(name, klass, n, extra, T, K) Row{name, klass, n, extra, &launch_t<T, K>}
    std::vector<Row> rows = {
        ROW("v_fma_f64", "fp64", 16, 0, double, k_fma64),
        ROW("v_add_f64", "fp64", 16, 0, double, k_add64),
        ROW("v_mul_f64", "fp64", 16, 0, double, k_mul64),
        ROW("v_max_f64", "fp64", 16, 0, double, k_max64),
        ROW("v_cmp_lt_f64", "fp64", 16, 0, double, k_cmp64),
        ROW("v_rcp_f64", "trans64", 16, 0, double, k_rcp64),
        ROW("v_sqrt_f64", "trans64", 16, 0, double, k_sqrt64),
        ROW("v_lshlrev_b64", "int64", 16, 0, unsigned long long, k_lshl64),
        ROW("v_mov_b64", "mov64", 16, 0, unsigned long long, k_mov64),
        ROW("v_cvt_f64_f32", "cvt", 16, 0, double, k_cvt_f64_f32),
        ROW("v_cvt_f32_f64", "cvt", 16, 0, double, k_cvt_f32_f64),
        ROW("v_cvt_i32_f64", "cvt", 16, 0, double, k_cvt_i32_f64),
        ROW("v_fma_f32", "fp32", 16, 0, float, k_fma32),
        ROW("v_add_f32", "fp32", 16, 0, float, k_add32),
        ROW("v_mul_f32", "fp32", 16, 0, float, k_mul32),
        ROW("v_max_f32", "fp32", 16, 0, float, k_max32),
        ROW("v_max3_f32", "fp32", 16, 0, float, k_max3_32),
        ROW("v_cmp_lt_f32", "fp32", 16, 0, float, k_cmp32),
        ROW("v_rcp_f32", "trans32", 16, 0, float, k_rcp32),
        ROW("v_cndmask_b32", "b32", 16, 0, unsigned, k_cndmask),
        ROW("v_cndmask_b32 (vcc set before the loop)", "b32", 16, 0, unsigned, k_cndmask_vccset),
        ROW("v_cndmask_b32_e64 (sgpr-pair mask)", "b32", 16, 0, unsigned, k_cndmask_s),
        ROW("v_cmp_lt_u32 vcc + v_cndmask_b32", "cmp+select", 32, 0, unsigned, k_cmp_cndmask),
        ROW("v_cmp_lt_u32_e64 sgpr + v_cndmask_b32_e64", "cmp+select", 32, 0, unsigned, k_cmp_cndmask_s),
        ROW("v_cmp_lt_u32 vcc + 2 v_cndmask_b32 (one mask, two selects: an fp64 select)", "cmp+2 selects", 48, 0, unsigned, k_cmp_2cnd),
        ROW("v_cmp_lt_u32_e64 sgpr + 2 v_cndmask_b32_e64", "cmp+2 selects", 48, 0, unsigned, k_cmp_2cnd_s),
        ROW("v_cmp_lt_u32_e64 sgpr", "int32", 16, 0, unsigned, k_cmp_s),
        ROW("v_min_u32", "int32", 16, 0, unsigned, k_minu32),
        ROW("v_mad_u32_u24", "int32", 16, 0, unsigned, k_mad_u32),
        ROW("v_or3_b32", "int32", 16, 0, unsigned, k_or3),
        ROW("v_lshl_add_u32", "int32", 16, 0, unsigned, k_lshladd),
        ROW("v_bfe_u32", "int32", 16, 0, unsigned, k_bfe),
        ROW("v_readlane_b32", "lane", 16, 0, unsigned, k_readlane),
        ROW("v_writelane_b32", "lane", 16, 0, unsigned, k_writelane),
        ROW("v_mov_b32", "b32", 16, 0, unsigned, k_mov32),
        ROW("v_add_u32", "int32", 16, 0, unsigned, k_addu32),
        ROW("v_and_b32", "int32", 16, 0, unsigned, k_and32),
        ROW("v_lshlrev_b32", "int32", 16, 0, unsigned, k_lshl32),
        ROW("v_mul_lo_u32", "int32_mul", 16, 0, unsigned, k_mullo),
        ROW("v_addc_co_u32", "int32", 16, 0, unsigned, k_addc),
        ROW("v_pk_fma_f32", "pk32", 16, 0, double, k_pkfma),
        ROW("v_pk_add_f32", "pk32", 16, 0, double, k_pkadd),
        ROW("v_pk_mul_f32", "pk32", 16, 0, double, k_pkmul),
        ROW("mix: v_fma_f64 + s_add_u32", "mix", 16, 16, double, k_mix_fma64_salu),
        ROW("mix: v_fma_f64 + 2 salu", "mix", 16, 32, double, k_mix_fma64_2salu),
        ROW("mix: v_fma_f64 + v_add_u32", "mix", 16, 16, double, k_mix_fma64_valu32),
        ROW("mix: v_fma_f64 + s_nop", "mix", 16, 16, double, k_mix_fma64_nop),
        ROW("mix: v_fma_f64 + ds_read_b32", "mix", 16, 16, double, k_mix_fma64_lds),
        ROW("s_add_u32 alone", "salu", 16, 0, double, k_salu_only),
        ROW("dep: v_fma_f64", "dep", 16, 0, double, k_dep_fma64),
        ROW("dep: v_add_f64", "dep", 16, 0, double, k_dep_add64),
        ROW("dep: v_fma_f32", "dep", 16, 0, float, k_dep_fma32),
        ROW("dep: v_add_u32", "dep", 16, 0, unsigned, k_dep_addu32),
        ROW("dep: v_rcp_f64", "dep", 16, 0, double, k_dep_rcp64),
    };
    // clock ramp: half a second of fp64 FMAs
    for (int i = 0; i < 400; ++i) rows[0].launch(cus * 4, d_out, s);
    CHECK(hipStreamSynchronize(s));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"iters\": %d,\n \"what\": \"cycles_per_inst_simd = median over SIMDs of (last end - first start of the SIMD's waves, s_memtime) / (instructions its waves issued); "
           "waves are assigned to SIMDs by HW_ID; simds_by_wave_count[i] = SIMDs that held i + 1 waves; launch_us = HIP-event time per launch\",\n \"rows\": [\n",
           prop.name, cus, prop.clockRate, kIters);
    std::vector<unsigned long long> h(4 * (size_t)max_waves);
    bool first = true;
    for (const Row& r : rows) {
        for (int k : {1, 2, 4}) {
            const int blocks = cus * k, waves = blocks * 4;
            r.launch(blocks, d_out, s);   // untimed (code object, instruction cache)
            CHECK(hipStreamSynchronize(s));
            CHECK(hipEventRecord(e0, s));
            const int reps = 20;
            for (int i = 0; i < reps; ++i) r.launch(blocks, d_out, s);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            CHECK(hipGetLastError());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 4 * waves, hipMemcpyDeviceToHost));
            // per SIMD (XCC, SE, SH, CU, SIMD of HW_ID): instructions issued by its waves / (last end - first start)
            std::map<unsigned long long, std::array<double, 4>> simd;   // first start, last end, waves, sum of wave spans
            std::vector<double> per_wave;
            const double n_class = (double)kIters * r.per_block;
            for (int w = 0; w < waves; ++w) {
                const unsigned long long t0 = h[4 * w], t1 = h[4 * w + 1], hw = h[4 * w + 2], xcc = h[4 * w + 3];
                const unsigned long long key = (xcc << 32) | (hw & 0xfff0u);   // (drop the wave slot)
                auto it = simd.find(key);
                if (it == simd.end()) simd[key] = {(double)t0, (double)t1, 1.0, (double)(t1 - t0)};
                else {
                    it->second[0] = std::min(it->second[0], (double)t0);
                    it->second[1] = std::max(it->second[1], (double)t1);
                    it->second[2] += 1.0;
                    it->second[3] += (double)(t1 - t0);
                }
                per_wave.push_back((double)(t1 - t0) / n_class);
            }
            std::vector<double> thr;
            int hist[10] = {0};
            for (auto& kv : simd) {
                thr.push_back((kv.second[1] - kv.second[0]) / (kv.second[2] * n_class));
                hist[std::min(9, (int)kv.second[2])]++;
            }
            std::sort(thr.begin(), thr.end());
            std::sort(per_wave.begin(), per_wave.end());
            printf("%s  {\"inst\": \"%s\", \"class\": \"%s\", \"waves_per_simd_target\": %d, \"cycles_per_inst_simd\": %.3f, "
                   "\"cycles_per_inst_simd_min\": %.3f, \"cycles_per_inst_simd_max\": %.3f, \"cycles_per_inst_wave_median\": %.3f, "
                   "\"simds\": %d, \"simds_by_wave_count\": [%d, %d, %d, %d, %d, %d, %d, %d, %d], \"other_insts_per_inst\": %.2f, \"launch_us\": %.2f}",
                   first ? "" : ",\n", r.name.c_str(), r.klass.c_str(), k, thr[thr.size() / 2], thr.front(), thr.back(),
                   per_wave[per_wave.size() / 2], (int)simd.size(), hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8], hist[9],
                   (double)r.extra / r.per_block, 1e3 * ms / reps);
            first = false;
        }
    }
    printf("\n ]}\n");
    return 0;
}

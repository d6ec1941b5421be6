// Instruction issue-rate microbenchmark for one gfx950 CU (is a wave64 fp32 VALU instruction 2 or 4 cycles?).
//
// One workgroup per CU (a 128 KiB LDS allocation keeps a second one out), W waves per SIMD (block = 256 * W threads), every
// wave runs ITER x 64 copies of one instruction as inline asm - a DEPENDENT chain (each result feeds the next) or 8
// INDEPENDENT chains - and reads the shader clock (s_memtime) and the 100 MHz real-time clock around the loop.
// Reported per case: cycles per instruction seen by one wave, and instructions issued per cycle and SIMD
// (= W * instructions / cycles). A 4-cycle wave64 op saturates at 0.25 / SIMD, a 2-cycle one at 0.5, a 1-cycle one at 1.0.
//
// build: hipcc --offload-arch=gfx950 -O2 -o issue_rate tools/issue_rate.hip ; run: ./issue_rate > profiles/issue_rate.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

typedef float v2f __attribute__((ext_vector_type(2)));

enum Op { FMA, PK_FMA, MUL, ADD, CNDMASK, EXP, RCP, MAD_U32_U24, ADD_U32, LSHL_ADD, CMP_BALLOT, DS_ADD_F32, DS_ADD_F32_SAME, DS_READ_B64, DS_READ_B128, DS_WRITE_B64, S_ADD, S_AND_B64, MIX_V_S, DPP_ADD, READLANE, CNDMASK_SGPR, CMP_E64, CMP_CNDMASK, DS_ADD_U32, DS_ADD_RTN_U32, DS_BPERMUTE, PERMLANE32_SWAP, GLOBAL_ATOMIC_ADD, SALU_VCC_CND, SALU_SGPR_CND, VCMP_SGPR_CND, VCMP_SXOR_CND, V_MOV, V_MIN, V_AND, V_CMP_E32, CND_VALU_VCC, V_MAC, MIN_F64, CMP_U64, CMP_U64_CND, N_OPS };
static const char *kNames[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_add_f32", "v_cndmask_b32", "v_exp_f32", "v_rcp_f32",
                                    "v_mad_u32_u24", "v_add_u32", "v_lshl_add_u32", "v_cmp+s_and(ballot)", "ds_add_f32(distinct)",
                                    "ds_add_f32(same addr)", "ds_read_b64", "ds_read_b128", "ds_write_b64", "s_add_u32", "s_and_b64",
                                    "v_fma_f32+s_add_u32", "v_add_f32_dpp", "v_readlane_b32", "v_cndmask_b32_e64(sgpr pair)", "v_cmp_lt_f32_e64->sgpr",
                                    "v_cmp(vcc)+v_cndmask(vcc)", "ds_add_u32(distinct)", "ds_add_rtn_u32(distinct)", "ds_bpermute_b32", "v_permlane32_swap_b32",
                                    "global_atomic_add_u32(no return, distinct dwords)", "s_and_b64(vcc)+v_cndmask(vcc)", "s_and_b64(sgpr)+v_cndmask_e64(sgpr)",
                                    "v_cmp_e64(sgpr)+v_cndmask_e64(sgpr)", "v_cmp_e64(sgpr)+s_xor_b64+v_cndmask_e64(sgpr)",
                                    "v_mov_b32", "v_min_f32", "v_and_b32", "v_cmp_lt_f32_e32(vcc)", "v_cndmask_b32(vcc written by one v_cmp)", "v_fmac_f32_e32",
                                    "v_min_f64", "v_cmp_gt_u64_e64->sgpr", "v_cmp_gt_u64(vcc)+2 v_cndmask(vcc)"};

// One asm statement holds the whole 64-instruction block: the compiler cannot see into it, so it neither reorders it nor
// pads it with s_nop (it does pad BETWEEN separate asm statements, which would be measured as issue slots).
#define X8(a) a a a a a a a a
#define DEP64(I) X8(X8(I(0)))
#define IND64(I) X8(I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7))
#define VREGS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
#define UREGS "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
#define PREGS "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
#define QREGS "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
#define BLOCK(I, REGS, ...)                                                    \
    do {                                                                       \
        if (DEP) asm volatile(DEP64(I) : REGS : __VA_ARGS__);                  \
        else asm volatile(IND64(I) : REGS : __VA_ARGS__);                      \
    } while (0)
// operands: %0..%7 the eight chain registers, %8 / %9 / ... the inputs
#define I_FMA(r) "v_fma_f32 %" #r ", %" #r ", %8, %9\n"
#define I_PKFMA(r) "v_pk_fma_f32 %" #r ", %" #r ", %8, %9\n"
#define I_MUL(r) "v_mul_f32 %" #r ", %" #r ", %8\n"
#define I_ADD(r) "v_add_f32 %" #r ", %" #r ", %9\n"
#define I_CND(r) "v_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define I_EXP(r) "v_exp_f32 %" #r ", %" #r "\n"
#define I_RCP(r) "v_rcp_f32 %" #r ", %" #r "\n"
#define I_MAD24(r) "v_mad_u32_u24 %" #r ", %" #r ", %8, %8\n"
#define I_ADDU(r) "v_add_u32 %" #r ", %" #r ", %8\n"
#define I_LSHLADD(r) "v_lshl_add_u32 %" #r ", %" #r ", 1, %8\n"
#define I_CMPB(r) "v_cmp_lt_f32 vcc, %" #r ", %8\n s_and_b64 %10, %10, vcc\n"
#define I_DSADD(r) "ds_add_f32 %10, %9\n"
#define I_DSR64(r) "ds_read_b64 %" #r ", %8\n"
#define I_DSR128(r) "ds_read_b128 %" #r ", %8\n"
#define I_DSW64(r) "ds_write_b64 %8, %" #r "\n"
#define I_SADD(r) "s_add_u32 %10, %10, %11\n"
#define I_SADD2(r) "s_add_u32 %10, %10, 1\n s_add_u32 %11, %11, 1\n"
#define I_SAND64(r) "s_and_b64 %10, %10, %10\n"
#define I_MIX(r) "v_fma_f32 %" #r ", %" #r ", %8, %9\n s_add_u32 %10, %10, 1\n"
#define I_DPP(r) "v_add_f32_dpp %" #r ", %" #r ", %" #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_RDL(r) "v_readlane_b32 %10, %" #r ", 3\n"
#define I_CNDS(r) "v_cndmask_b32_e64 %" #r ", %" #r ", %8, %10\n"
#define I_CMP64(r) "v_cmp_lt_f32_e64 %10, %" #r ", %8\n"
#define I_CMPCND(r) "v_cmp_lt_f32 vcc, %" #r ", %8\n v_cndmask_b32 %" #r ", %" #r ", %9, vcc\n"
#define I_DSADDU(r) "ds_add_u32 %10, %9\n"
#define I_DSADDRTN(r) "ds_add_rtn_u32 %" #r ", %10, %9\n"
#define I_BPERM(r) "ds_bpermute_b32 %" #r ", %10, %" #r "\n"
#define I_PL32(r) "v_permlane32_swap_b32 %" #r ", %8\n"
#define I_GATOM(r) "global_atomic_add %10, %9, off\n"
#define I_SVCC(r) "s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define I_SSGPR(r) "s_and_b64 %10, %10, exec\n v_cndmask_b32_e64 %" #r ", %" #r ", %8, %10\n"
#define I_VSGPR(r) "v_cmp_lt_f32_e64 %10, %" #r ", %8\n v_cndmask_b32_e64 %" #r ", %" #r ", %9, %10\n"
#define I_VXOR(r) "v_cmp_lt_f32_e64 %10, %" #r ", %8\n s_xor_b64 %10, %10, exec\n v_cndmask_b32_e64 %" #r ", %" #r ", %9, %10\n"

#define I_MOV(r) "v_mov_b32 %" #r ", %8\n"
#define I_MIN(r) "v_min_f32 %" #r ", %" #r ", %8\n"
#define I_AND(r) "v_and_b32 %" #r ", %" #r ", %8\n"
#define I_CMP32(r) "v_cmp_lt_f32 vcc, %" #r ", %8\n"
#define I_MAC(r) "v_fmac_f32 %" #r ", %8, %9\n"
#define I_MINF64(r) "v_min_f64 %" #r ", %" #r ", %8\n"
#define I_CMPU64(r) "v_cmp_gt_u64_e64 %10, %" #r ", %8\n"
#define I_CMPU64CND(r) "v_cmp_gt_u64 vcc, %" #r ", %8\n v_cndmask_b32 %11, %11, %9, vcc\n v_cndmask_b32 %12, %12, %9, vcc\n"

typedef float v4f __attribute__((ext_vector_type(4)));

template <int OP, bool DEP>
__global__ void __launch_bounds__(1024) rate_kernel(int iters, uint64_t *cycles, uint64_t *real, float *sink, uint32_t *gbuf)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    float x[8];
    v2f p[8];
    v4f q[8];
    uint32_t u[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = 1.0f + lane * 1e-3f + i; p[i] = v2f{x[i], x[i] + 0.5f}; u[i] = lane + i;
        q[i] = v4f{x[i], x[i], x[i], x[i]};
    }
    float a = 1.0000001f, b = 1e-9f;
    v2f a2 = {a, a}, b2 = {b, b};
    uint32_t c = (uint32_t)lane | 1u;
    uint32_t saddr = (OP == DS_ADD_F32_SAME) ? 0u : (uint32_t)(threadIdx.x * 16u);
    uint32_t s0 = 1, s1 = 2;
    uint64_t sm = ~0ull;
    uint32_t *gaddr = gbuf + (size_t)blockIdx.x * 1024 + threadIdx.x; // one dword per lane: a wave adds into 2 cache lines
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
    lds[threadIdx.x * 4] = 0.f; lds[threadIdx.x * 4 + 1] = 0.f; lds[threadIdx.x * 4 + 2] = 0.f; lds[threadIdx.x * 4 + 3] = 0.f;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) BLOCK(I_FMA, VREGS, "v"(a), "v"(b));
        if (OP == PK_FMA) BLOCK(I_PKFMA, PREGS, "v"(a2), "v"(b2));
        if (OP == MUL) BLOCK(I_MUL, VREGS, "v"(a), "v"(b));
        if (OP == ADD) BLOCK(I_ADD, VREGS, "v"(a), "v"(b));
        if (OP == CNDMASK) BLOCK(I_CND, VREGS, "v"(a), "v"(b));
        if (OP == EXP) BLOCK(I_EXP, VREGS, "v"(a), "v"(b));
        if (OP == RCP) BLOCK(I_RCP, VREGS, "v"(a), "v"(b));
        if (OP == MAD_U32_U24) BLOCK(I_MAD24, UREGS, "v"(c));
        if (OP == ADD_U32) BLOCK(I_ADDU, UREGS, "v"(c));
        if (OP == LSHL_ADD) BLOCK(I_LSHLADD, UREGS, "v"(c));
        if (OP == CMP_BALLOT) asm volatile(DEP64(I_CMPB) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : "vcc", "scc");
        if (OP == DS_ADD_F32 || OP == DS_ADD_F32_SAME) asm volatile(DEP64(I_DSADD) "s_waitcnt lgkmcnt(0)\n" : VREGS, "+v"(a), "+v"(b), "+v"(saddr) : : "memory");
        if (OP == DS_READ_B64) { if (DEP) asm volatile(DEP64(I_DSR64) "s_waitcnt lgkmcnt(0)\n" : PREGS : "v"(saddr) : "memory"); else asm volatile(IND64(I_DSR64) "s_waitcnt lgkmcnt(0)\n" : PREGS : "v"(saddr) : "memory"); }
        if (OP == DS_READ_B128) { if (DEP) asm volatile(DEP64(I_DSR128) "s_waitcnt lgkmcnt(0)\n" : QREGS : "v"(saddr) : "memory"); else asm volatile(IND64(I_DSR128) "s_waitcnt lgkmcnt(0)\n" : QREGS : "v"(saddr) : "memory"); }
        if (OP == DS_WRITE_B64) asm volatile(IND64(I_DSW64) "s_waitcnt lgkmcnt(0)\n" : PREGS : "v"(saddr) : "memory");
        if (OP == S_ADD) { if (DEP) asm volatile(DEP64(I_SADD) : VREGS, "+v"(a), "+v"(b), "+s"(s0), "+s"(s1) : : "scc"); else asm volatile(X8(X8(I_SADD2(0))) : VREGS, "+v"(a), "+v"(b), "+s"(s0), "+s"(s1) : : "scc"); }
        if (OP == S_AND_B64) asm volatile(DEP64(I_SAND64) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : "scc");
        if (OP == MIX_V_S) { if (DEP) asm volatile(DEP64(I_MIX) : VREGS, "+v"(a), "+v"(b), "+s"(s0) : : "scc"); else asm volatile(IND64(I_MIX) : VREGS, "+v"(a), "+v"(b), "+s"(s0) : : "scc"); }
        if (OP == DPP_ADD) BLOCK(I_DPP, VREGS, "v"(a), "v"(b));
        if (OP == READLANE) asm volatile(IND64(I_RDL) : UREGS, "+v"(a), "+v"(b), "+s"(s0) : : );
        if (OP == CNDMASK_SGPR) { if (DEP) asm volatile(DEP64(I_CNDS) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : ); else asm volatile(IND64(I_CNDS) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : ); }
        if (OP == CMP_E64) asm volatile(IND64(I_CMP64) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : );
        if (OP == CMP_CNDMASK) { if (DEP) asm volatile(DEP64(I_CMPCND) : VREGS, "+v"(a), "+v"(b) : : "vcc"); else asm volatile(IND64(I_CMPCND) : VREGS, "+v"(a), "+v"(b) : : "vcc"); }
        if (OP == DS_ADD_U32) asm volatile(DEP64(I_DSADDU) "s_waitcnt lgkmcnt(0)\n" : UREGS, "+v"(a), "+v"(c), "+v"(saddr) : : "memory");
        if (OP == DS_ADD_RTN_U32) asm volatile(IND64(I_DSADDRTN) "s_waitcnt lgkmcnt(0)\n" : UREGS, "+v"(a), "+v"(c), "+v"(saddr) : : "memory");
        if (OP == DS_BPERMUTE) asm volatile(IND64(I_BPERM) "s_waitcnt lgkmcnt(0)\n" : UREGS, "+v"(a), "+v"(c), "+v"(saddr) : : "memory");
        if (OP == PERMLANE32_SWAP) { if (DEP) asm volatile(DEP64(I_PL32) : UREGS, "+v"(c) : : ); else asm volatile(IND64(I_PL32) : UREGS, "+v"(c) : : ); }
        if (OP == SALU_VCC_CND) { if (DEP) asm volatile(DEP64(I_SVCC) : VREGS, "+v"(a), "+v"(b) : : "vcc", "scc"); else asm volatile(IND64(I_SVCC) : VREGS, "+v"(a), "+v"(b) : : "vcc", "scc"); }
        if (OP == SALU_SGPR_CND) { if (DEP) asm volatile(DEP64(I_SSGPR) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : "scc"); else asm volatile(IND64(I_SSGPR) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : "scc"); }
        if (OP == VCMP_SGPR_CND) { if (DEP) asm volatile(DEP64(I_VSGPR) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : ); else asm volatile(IND64(I_VSGPR) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : ); }
        if (OP == VCMP_SXOR_CND) { if (DEP) asm volatile(DEP64(I_VXOR) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : "scc"); else asm volatile(IND64(I_VXOR) : VREGS, "+v"(a), "+v"(b), "+s"(sm) : : "scc"); }
        if (OP == V_MOV) BLOCK(I_MOV, VREGS, "v"(a), "v"(b));
        if (OP == V_MIN) BLOCK(I_MIN, VREGS, "v"(a), "v"(b));
        if (OP == V_AND) BLOCK(I_AND, UREGS, "v"(c));
        if (OP == V_CMP_E32) { if (DEP) asm volatile(DEP64(I_CMP32) : VREGS, "+v"(a), "+v"(b) : : "vcc"); else asm volatile(IND64(I_CMP32) : VREGS, "+v"(a), "+v"(b) : : "vcc"); }
        if (OP == CND_VALU_VCC) {
            if (it == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n" : : "v"(a), "v"(x[0]) : "vcc");
            BLOCK(I_CND, VREGS, "v"(a), "v"(b));
        }
        if (OP == V_MAC) BLOCK(I_MAC, VREGS, "v"(a), "v"(b));
        if (OP == MIN_F64) BLOCK(I_MINF64, PREGS, "v"(a2), "v"(b2));
        if (OP == CMP_U64) asm volatile(IND64(I_CMPU64) : PREGS, "+v"(a2), "+v"(b2), "+s"(sm) : : );
        if (OP == CMP_U64_CND) { float c2 = b; if (DEP) asm volatile(DEP64(I_CMPU64CND) : PREGS, "+v"(a2), "+v"(b), "+s"(sm), "+v"(a), "+v"(c2) : : "vcc"); else asm volatile(IND64(I_CMPU64CND) : PREGS, "+v"(a2), "+v"(b), "+s"(sm), "+v"(a), "+v"(c2) : : "vcc"); a += c2; }
        if (OP == GLOBAL_ATOMIC_ADD) asm volatile(DEP64(I_GATOM) "s_waitcnt vmcnt(0)\n" : UREGS, "+v"(a), "+v"(c), "+v"(gaddr) : : "memory");
    }
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float acc = a + b + (float)c;
    for (int i = 0; i < 8; ++i) acc += x[i] + p[i].x + p[i].y + (float)u[i] + q[i].x + q[i].w;
    acc += (float)s0 + (float)s1 + (float)(uint32_t)sm;
    if (acc == 12345.678f) sink[0] = acc + lds[lane];
    if (lane == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cycles[w] = t1 - t0;
        real[w]   = r1 - r0;
    }
}

template <int OP>
static void run_op(std::string &out, bool &first)
{
    const int n_cu = 256;
    uint64_t *cyc, *real;
    float *sink;
    uint32_t *gbuf;
    hipMalloc(&cyc, n_cu * 32 * 8); hipMalloc(&real, n_cu * 32 * 8); hipMalloc(&sink, 64); hipMalloc(&gbuf, 512 * 1024 * 4);
    hipMemset(gbuf, 0, 512 * 1024 * 4);
    const bool lds_op = OP == DS_ADD_F32 || OP == DS_ADD_F32_SAME || OP == DS_READ_B64 || OP == DS_READ_B128 || OP == DS_WRITE_B64
                        || OP == DS_ADD_U32 || OP == DS_ADD_RTN_U32 || OP == DS_BPERMUTE || OP == GLOBAL_ATOMIC_ADD;
    const int iters = (OP == EXP || OP == RCP || lds_op) ? 2000 : 8000;
    for (int dep = 1; dep >= 0; --dep) {
        if (!dep && (OP == DS_ADD_F32 || OP == DS_ADD_F32_SAME || OP == S_AND_B64 || OP == CMP_BALLOT || OP == DS_ADD_U32 || OP == GLOBAL_ATOMIC_ADD)) continue;
        if (dep && (OP == READLANE || OP == DS_WRITE_B64 || OP == CMP_E64 || OP == CMP_U64 || OP == DS_ADD_RTN_U32 || OP == DS_BPERMUTE)) continue;
        for (int w : {1, 2, 4, 8}) { // waves per SIMD; one block per CU (LDS-limited), two blocks of 1024 threads for w = 8
            const int blocks_per_cu = w == 8 ? 2 : 1;
            const int threads = 256 * (w / blocks_per_cu);
            const size_t lds = blocks_per_cu == 2 ? 72 * 1024 : 128 * 1024;
            auto k = dep ? rate_kernel<OP, true> : rate_kernel<OP, false>;
            hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            hipLaunchKernelGGL(k, dim3(n_cu * blocks_per_cu), dim3(threads), lds, 0, 50, cyc, real, sink, gbuf); // warm-up
            hipEvent_t e0, e1; // wall clock of the whole launch: an independent check of the in-kernel counters
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(n_cu * blocks_per_cu), dim3(threads), lds, 0, iters, cyc, real, sink, gbuf);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float wall_ms = 0.0f;
            hipEventElapsedTime(&wall_ms, e0, e1);
            hipEventDestroy(e0); hipEventDestroy(e1);
            const int n_waves = n_cu * 4 * w;
            std::vector<uint64_t> hc(n_waves), hr(n_waves);
            hipMemcpy(hc.data(), cyc, n_waves * 8, hipMemcpyDeviceToHost);
            hipMemcpy(hr.data(), real, n_waves * 8, hipMemcpyDeviceToHost);
            double mc = 0, mr = 0;
            for (int i = 0; i < n_waves; ++i) { mc += (double)hc[i]; mr += (double)hr[i]; }
            mc /= n_waves; mr /= n_waves;
            const double instr = (double)iters * 64 * ((OP == VCMP_SXOR_CND || OP == CMP_U64_CND) ? 3 : (OP == MIX_V_S || OP == CMP_BALLOT || OP == CMP_CNDMASK || OP == SALU_VCC_CND || OP == SALU_SGPR_CND || OP == VCMP_SGPR_CND || (OP == S_ADD && !dep)) ? 2 : 1);
            const double ns = mr * 10.0; // 100 MHz real-time counter
            char buf[512];
            snprintf(buf, sizeof buf,
                     "%s\n  {\"op\": \"%s\", \"chain\": \"%s\", \"waves_per_simd\": %d, \"counter_ticks_per_instr\": %.3f, \"ns_per_instr_per_wave\": %.4f, "
                     "\"instr_per_ns_per_simd\": %.4f, \"launch_wall_us\": %.1f, \"instr_per_ns_per_simd_by_wall_clock\": %.4f}",
                     first ? "" : ",", kNames[OP], dep ? "dependent" : "independent x8", w, mc / instr, ns / instr, w * instr / ns,
                     wall_ms * 1e3, (double)n_waves * instr / (wall_ms * 1e6) / (n_cu * 4));
            first = false;
            out += buf;
        }
    }
    hipFree(cyc); hipFree(real); hipFree(sink); hipFree(gbuf);
}

static int g_first_op = 0;
template <int OP>
static void run_all(std::string &out, bool &first)
{
    if (OP >= g_first_op) run_op<OP>(out, first);
    if constexpr (OP + 1 < N_OPS) run_all<OP + 1>(out, first);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_first_op = atoi(argv[1]);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    std::string out;
    bool first = true;
    run_all<0>(out, first);
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d,\n \"note\": \"instr_per_ns_per_simd / (shader clock in GHz) = wave-instructions issued per cycle and SIMD; "
           "counter_ticks = s_memtime ticks (not necessarily shader cycles)\",\n \"cases\": [%s\n]}\n",
           prop.name, prop.gcnArchName, prop.multiProcessorCount, clk_khz, out.c_str());
    return 0;
}

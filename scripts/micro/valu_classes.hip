// Micro-benchmark: which wave64 VALU instruction forms run at the "fast" FP32 rate on gfx950 (~2.25 SIMD cycles per
// wave-instruction, valu_issue.hip) and which at the ~4.5-cycle rate?
//
// Why (round 3): rocprof's VALUBusy = SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE comes out at ~100 % for the three
// tri compositing kernels, while DESIGN.md priced every SQ_INSTS_VALU at 2.25 cycles ("40-47 % of issue").  Both are
// right only if most of the kernels' instructions are NOT the fast kind.  This table says which forms are which, so
// that the kernels' inner loops can be rewritten towards the fast ones.
//
// Method: 4 waves per SIMD (four 256-thread blocks per CU, forced by the dynamic LDS request), 4 independent chains
// per wave, 64 instructions per loop iteration, cycles from the kernel's wall time x the clock the chip held
// (s_memtime / s_memrealtime inside the kernel), as in valu_issue.hip.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o valu_classes valu_classes.hip && ./valu_classes [out.txt]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Stamp { unsigned long long cyc, real; };

#define R4(I) I(a0) I(a1) I(a2) I(a3)
#define R16(I) R4(I) R4(I) R4(I) R4(I)
#define R64(I) R16(I) R16(I) R16(I) R16(I)

// every form: destination = its own chain register, sources = chain register + loop-invariant operands
#define F_FMA(r) "v_fma_f32 %[" #r "], %[" #r "], %[m], %[c]\n\t"
#define F_FMAC(r) "v_fmac_f32 %[" #r "], %[m], %[c]\n\t"
#define F_ADD(r) "v_add_f32 %[" #r "], %[" #r "], %[c]\n\t"
#define F_SUB(r) "v_sub_f32 %[" #r "], %[" #r "], %[c]\n\t"
#define F_MUL(r) "v_mul_f32 %[" #r "], %[" #r "], %[m]\n\t"
#define F_MULNEG(r) "v_mul_f32 %[" #r "], -%[" #r "], |%[m]|\n\t"
#define F_MAX(r) "v_max_f32 %[" #r "], %[" #r "], %[c]\n\t"
#define F_MIN(r) "v_min_f32 %[" #r "], %[" #r "], %[m]\n\t"
#define F_MED3(r) "v_med3_f32 %[" #r "], %[" #r "], %[c], %[m]\n\t"
#define F_MOV(r) "v_mov_b32 %[" #r "], %[c]\n\t"
#define F_ADDU(r) "v_add_u32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_SUBU(r) "v_sub_u32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_ADDCO(r) "v_add_co_u32 %[" #r "], vcc, %[" #r "], %[ci]\n\t"
#define F_ADD3(r) "v_add3_u32 %[" #r "], %[" #r "], %[ci], %[mi]\n\t"
#define F_LSHLADD(r) "v_lshl_add_u32 %[" #r "], %[" #r "], 1, %[ci]\n\t"
#define F_AND(r) "v_and_b32 %[" #r "], %[" #r "], %[mi]\n\t"
#define F_OR(r) "v_or_b32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_XOR(r) "v_xor_b32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_ANDOR(r) "v_and_or_b32 %[" #r "], %[" #r "], %[mi], %[ci]\n\t"
#define F_OR3(r) "v_or3_b32 %[" #r "], %[" #r "], %[mi], %[ci]\n\t"
#define F_BFE(r) "v_bfe_u32 %[" #r "], %[" #r "], 3, 9\n\t"
#define F_BFI(r) "v_bfi_b32 %[" #r "], %[mi], %[" #r "], %[ci]\n\t"
#define F_SHL(r) "v_lshlrev_b32 %[" #r "], 1, %[" #r "]\n\t"
#define F_SHR(r) "v_lshrrev_b32 %[" #r "], 1, %[" #r "]\n\t"
#define F_SHLV(r) "v_lshlrev_b32 %[" #r "], %[ci], %[" #r "]\n\t"
#define F_ASHR(r) "v_ashrrev_i32 %[" #r "], 1, %[" #r "]\n\t"
#define F_FFBL(r) "v_ffbl_b32 %[" #r "], %[" #r "]\n\t"
#define F_FFBH(r) "v_ffbh_u32 %[" #r "], %[" #r "]\n\t"
#define F_BCNT(r) "v_bcnt_u32_b32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_MBCNT(r) "v_mbcnt_lo_u32_b32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_MULLO(r) "v_mul_lo_u32 %[" #r "], %[" #r "], %[mi]\n\t"
#define F_MULHI(r) "v_mul_hi_u32 %[" #r "], %[" #r "], %[mi]\n\t"
#define F_MUL24(r) "v_mul_u32_u24 %[" #r "], %[" #r "], %[mi]\n\t"
#define F_MAD24(r) "v_mad_u32_u24 %[" #r "], %[" #r "], %[mi], %[ci]\n\t"
#define F_MADI24(r) "v_mad_i32_i24 %[" #r "], %[" #r "], %[mi], %[ci]\n\t"
#define F_MINU(r) "v_min_u32 %[" #r "], %[" #r "], %[mi]\n\t"
#define F_MAXI(r) "v_max_i32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_MIN3(r) "v_min3_u32 %[" #r "], %[" #r "], %[mi], %[ci]\n\t"
#define F_CNDV(r) "v_cndmask_b32 %[" #r "], %[" #r "], %[c], vcc\n\t"
#define F_CMPF(r) "v_cmp_lt_f32 vcc, %[" #r "], %[c]\n\t"
#define F_CMPFS(r) "v_cmp_lt_f32 s[10:11], %[" #r "], %[c]\n\t"
#define F_CMPU(r) "v_cmp_lt_u32 vcc, %[" #r "], %[ci]\n\t"
#define F_CMPXCHAIN(r) "v_cmp_lt_f32 vcc, %[" #r "], %[c]\n\tv_cndmask_b32 %[" #r "], %[" #r "], %[m], vcc\n\t"
#define F_CVTFU(r) "v_cvt_f32_u32 %[" #r "], %[" #r "]\n\t"
#define F_CVTIF(r) "v_cvt_i32_f32 %[" #r "], %[" #r "]\n\t"
#define F_RCP(r) "v_rcp_f32 %[" #r "], %[" #r "]\n\t"
#define F_RSQ(r) "v_rsq_f32 %[" #r "], %[" #r "]\n\t"
#define F_EXP(r) "v_exp_f32 %[" #r "], %[" #r "]\n\t"
#define F_FRACT(r) "v_fract_f32 %[" #r "], %[" #r "]\n\t"
#define F_FLOOR(r) "v_floor_f32 %[" #r "], %[" #r "]\n\t"
#define F_LDEXP(r) "v_ldexp_f32 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_DIVFIX(r) "v_div_fixup_f32 %[" #r "], %[" #r "], %[m], %[c]\n\t"
#define F_DIVFMAS(r) "v_div_fmas_f32 %[" #r "], %[" #r "], %[m], %[c]\n\t"
#define F_DIVSCALE(r) "v_div_scale_f32 %[" #r "], vcc, %[" #r "], %[m], %[c]\n\t"
#define F_DPPADD(r) "v_add_f32_dpp %[" #r "], %[" #r "], %[c] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define F_DPPMOV(r) "v_mov_b32_dpp %[" #r "], %[" #r "] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define F_SDWA(r) "v_add_u32_sdwa %[" #r "], %[" #r "], %[ci] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
#define F_PKFMA(r) "v_pk_fma_f32 %[" #r "], %[" #r "], %[pm], %[pc]\n\t"
#define F_PKADD(r) "v_pk_add_f32 %[" #r "], %[" #r "], %[pc]\n\t"
#define F_PKMUL(r) "v_pk_mul_f32 %[" #r "], %[" #r "], %[pm]\n\t"
#define F_PKADDU16(r) "v_pk_add_u16 %[" #r "], %[" #r "], %[ci]\n\t"
#define F_ADD64(r) "v_lshlrev_b64 %[" #r "], 1, %[" #r "]\n\t"
#define F_MAD64(r) "v_mad_u64_u32 %[" #r "], vcc, %[mi], %[ci], %[" #r "]\n\t"
#define F_FMA64(r) "v_fma_f64 %[" #r "], %[" #r "], %[dm], %[dc]\n\t"
#define F_ADD64F(r) "v_add_f64 %[" #r "], %[" #r "], %[dc]\n\t"
#define F_READLANE(r) "v_readfirstlane_b32 s10, %[" #r "]\n\t"
#define F_PERM(r) "v_perm_b32 %[" #r "], %[" #r "], %[ci], %[mi]\n\t"
#define F_ALIGNBIT(r) "v_alignbit_b32 %[" #r "], %[" #r "], %[ci], 7\n\t"
#define F_SAD(r) "v_sad_u32 %[" #r "], %[" #r "], %[mi], %[ci]\n\t"
#define F_FMAMIX(r) "v_fma_mix_f32 %[" #r "], %[" #r "], %[m], %[c]\n\t"
#define F_CUBE(r) "v_mul_legacy_f32 %[" #r "], %[" #r "], %[m]\n\t"

typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind { K_F32, K_PK, K_B64, K_F64 };

#define KERNEL(NAME, FORM, KIND)                                                                                  \
    __global__ void __launch_bounds__(256) k_##NAME(Stamp* out, int iters) {                                      \
        extern __shared__ char lds_pad[];                                                                         \
        float a[4]; f2 p[4]; unsigned long long q[4]; double d[4];                                                \
        for (int i = 0; i < 4; i++) {                                                                             \
            a[i] = threadIdx.x * 0.001f + i + 1.0f; p[i] = f2{a[i], a[i] + 0.5f};                                 \
            q[i] = threadIdx.x * 77u + i; d[i] = a[i];                                                            \
        }                                                                                                         \
        const float m = 0.999f, c = 0.001f; const f2 pm = {0.999f, 0.998f}, pc = {0.001f, 0.002f};                \
        const unsigned mi = 0x00ff37u, ci = 5u; const double dm = 0.999, dc = 0.001;                              \
        __syncthreads();                                                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();        \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                       \
        for (int it = 0; it < iters; it++) {                                                                      \
            if (KIND == K_F32)                                                                                    \
                asm volatile(R64(FORM) : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3])       \
                             : [m] "v"(m), [c] "v"(c), [mi] "v"(mi), [ci] "v"(ci), [pm] "v"(pm), [pc] "v"(pc),    \
                               [dm] "v"(dm), [dc] "v"(dc) : "vcc", "s10", "s11");                                 \
            else if (KIND == K_PK)                                                                                \
                asm volatile(R64(FORM) : [a0] "+v"(p[0]), [a1] "+v"(p[1]), [a2] "+v"(p[2]), [a3] "+v"(p[3])       \
                             : [m] "v"(m), [c] "v"(c), [mi] "v"(mi), [ci] "v"(ci), [pm] "v"(pm), [pc] "v"(pc),    \
                               [dm] "v"(dm), [dc] "v"(dc) : "vcc", "s10", "s11");                                 \
            else if (KIND == K_B64)                                                                               \
                asm volatile(R64(FORM) : [a0] "+v"(q[0]), [a1] "+v"(q[1]), [a2] "+v"(q[2]), [a3] "+v"(q[3])       \
                             : [m] "v"(m), [c] "v"(c), [mi] "v"(mi), [ci] "v"(ci), [pm] "v"(pm), [pc] "v"(pc),    \
                               [dm] "v"(dm), [dc] "v"(dc) : "vcc", "s10", "s11");                                 \
            else                                                                                                  \
                asm volatile(R64(FORM) : [a0] "+v"(d[0]), [a1] "+v"(d[1]), [a2] "+v"(d[2]), [a3] "+v"(d[3])       \
                             : [m] "v"(m), [c] "v"(c), [mi] "v"(mi), [ci] "v"(ci), [pm] "v"(pm), [pc] "v"(pc),    \
                               [dm] "v"(dm), [dc] "v"(dc) : "vcc", "s10", "s11");                                 \
        }                                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();        \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                       \
        float s = 0.f;                                                                                            \
        for (int i = 0; i < 4; i++) s += a[i] + p[i].x + p[i].y + (float)q[i] + (float)d[i];                      \
        if (s == 12345.678f) out[0].cyc = (unsigned long long)s;                                                  \
        if ((threadIdx.x & 63) == 0) {                                                                            \
            const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);                                                 \
            out[wave].cyc = t1 - t0; out[wave].real = r1 - r0;                                                    \
        }                                                                                                         \
    }

#define FORMS(X)                                                                                                   \
    X(fma_f32, F_FMA, K_F32, 1) X(fmac_f32, F_FMAC, K_F32, 1) X(add_f32, F_ADD, K_F32, 1) X(sub_f32, F_SUB, K_F32, 1) \
    X(mul_f32, F_MUL, K_F32, 1) X(mul_f32_neg_abs, F_MULNEG, K_F32, 1) X(max_f32, F_MAX, K_F32, 1)                  \
    X(min_f32, F_MIN, K_F32, 1) X(med3_f32, F_MED3, K_F32, 1) X(mov_b32, F_MOV, K_F32, 1)                           \
    X(add_u32, F_ADDU, K_F32, 1) X(sub_u32, F_SUBU, K_F32, 1) X(add_co_u32, F_ADDCO, K_F32, 1)                      \
    X(add3_u32, F_ADD3, K_F32, 1) X(lshl_add_u32, F_LSHLADD, K_F32, 1) X(and_b32, F_AND, K_F32, 1)                  \
    X(or_b32, F_OR, K_F32, 1) X(xor_b32, F_XOR, K_F32, 1) X(and_or_b32, F_ANDOR, K_F32, 1) X(or3_b32, F_OR3, K_F32, 1) \
    X(bfe_u32, F_BFE, K_F32, 1) X(bfi_b32, F_BFI, K_F32, 1) X(lshlrev_b32_imm, F_SHL, K_F32, 1)                     \
    X(lshrrev_b32_imm, F_SHR, K_F32, 1) X(lshlrev_b32_vgpr, F_SHLV, K_F32, 1) X(ashrrev_i32, F_ASHR, K_F32, 1)      \
    X(ffbl_b32, F_FFBL, K_F32, 1) X(ffbh_u32, F_FFBH, K_F32, 1) X(bcnt_u32_b32, F_BCNT, K_F32, 1)                   \
    X(mbcnt_lo, F_MBCNT, K_F32, 1) X(mul_lo_u32, F_MULLO, K_F32, 1) X(mul_hi_u32, F_MULHI, K_F32, 1)                \
    X(mul_u32_u24, F_MUL24, K_F32, 1) X(mad_u32_u24, F_MAD24, K_F32, 1) X(mad_i32_i24, F_MADI24, K_F32, 1)          \
    X(min_u32, F_MINU, K_F32, 1) X(max_i32, F_MAXI, K_F32, 1) X(min3_u32, F_MIN3, K_F32, 1)                         \
    X(cndmask_b32_vcc, F_CNDV, K_F32, 1) X(cmp_lt_f32_vcc, F_CMPF, K_F32, 1) X(cmp_lt_f32_sgpr, F_CMPFS, K_F32, 1)  \
    X(cmp_lt_u32_vcc, F_CMPU, K_F32, 1) X(cmp_then_cndmask_pair, F_CMPXCHAIN, K_F32, 2)                             \
    X(cvt_f32_u32, F_CVTFU, K_F32, 1) X(cvt_i32_f32, F_CVTIF, K_F32, 1) X(rcp_f32, F_RCP, K_F32, 1)                 \
    X(rsq_f32, F_RSQ, K_F32, 1) X(exp_f32, F_EXP, K_F32, 1) X(fract_f32, F_FRACT, K_F32, 1)                         \
    X(floor_f32, F_FLOOR, K_F32, 1) X(ldexp_f32, F_LDEXP, K_F32, 1) X(div_fixup_f32, F_DIVFIX, K_F32, 1)            \
    X(div_fmas_f32, F_DIVFMAS, K_F32, 1) X(div_scale_f32, F_DIVSCALE, K_F32, 1) X(add_f32_dpp, F_DPPADD, K_F32, 1)  \
    X(mov_b32_dpp_quad, F_DPPMOV, K_F32, 1) X(add_u32_sdwa, F_SDWA, K_F32, 1) X(perm_b32, F_PERM, K_F32, 1)         \
    X(alignbit_b32, F_ALIGNBIT, K_F32, 1) X(sad_u32, F_SAD, K_F32, 1) X(fma_mix_f32, F_FMAMIX, K_F32, 1)            \
    X(mul_legacy_f32, F_CUBE, K_F32, 1) X(readfirstlane, F_READLANE, K_F32, 1)                                      \
    X(pk_fma_f32, F_PKFMA, K_PK, 1) X(pk_add_f32, F_PKADD, K_PK, 1) X(pk_mul_f32, F_PKMUL, K_PK, 1)                 \
    X(lshlrev_b64, F_ADD64, K_B64, 1) X(mad_u64_u32, F_MAD64, K_B64, 1) X(fma_f64, F_FMA64, K_F64, 1)               \
    X(add_f64, F_ADD64F, K_F64, 1)

#define DEF(NAME, FORM, KIND, N) KERNEL(NAME, FORM, KIND)
FORMS(DEF)

typedef void (*kern_t)(Stamp*, int);

static void run(const char* name, kern_t k, int per_form, int wps, FILE* f) {
    const int blocks = 256 * wps, iters = 1000, per_iter = 64 * per_form;
    const size_t lds = (160 * 1024) / wps - 512;
    Stamp* out; hipMalloc(&out, sizeof(Stamp) * blocks * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks * 4);
    hipMemcpy(h.data(), out, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (auto& s : h) clk.push_back(s.real ? (double)s.cyc / (double)s.real * 100.0 : 0.0);
    std::sort(clk.begin(), clk.end());
    const double n = (double)iters * per_iter;
    const double wall_cyc = ms * 1e-3 * clk[clk.size() / 2] * 1e6 / (n * wps);
    fprintf(f, "%-24s waves/SIMD %d : %6.2f SIMD cycles per wave-instruction (wall time x clock %4.0f MHz, kernel %7.3f ms)\n",
            name, wps, wall_cyc, clk[clk.size() / 2], ms);
    fflush(f);
    hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    FILE* f = argc > 1 ? fopen(argv[1], "w") : stdout;
    if (!f) return 1;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    fprintf(f, "# %s, %d CUs; 4 chains per wave, 64 instructions per iteration, 1000 iterations, 256-thread blocks\n",
            pr.gcnArchName, pr.multiProcessorCount);
#define RUN(NAME, FORM, KIND, N) run(#NAME, k_##NAME, N, 4, f);
    FORMS(RUN)
    fprintf(f, "\n# the same at 6 waves per SIMD (the forward's occupancy) for four forms\n");
    run("fma_f32", k_fma_f32, 1, 6, f); run("add_u32", k_add_u32, 1, 6, f);
    run("cndmask_b32_vcc", k_cndmask_b32_vcc, 1, 6, f); run("and_b32", k_and_b32, 1, 6, f);
    if (f != stdout) fclose(f);
    return 0;
}

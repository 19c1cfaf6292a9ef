// Micro-benchmark: VALU issue rates on gfx950 that decide how the hit-parallel backward should be written:
// plain v_fma_f32, 32/64-bit integer compare + add-with-carry (the rank sort's inner loop), packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, DPP forms (v_fmac_f32_dpp, v_add_f32_dpp,
// v_cndmask_b32_dpp + v_add_f32) and v_rcp_f32.  Reports SIMD cycles per wave-instruction with the SIMDs full.
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define FMA(i) "v_fma_f32 %[a" #i "], %[a" #i "], %[m], %[c]\n\t"
#define PKFMA(i) "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[pm], %[pc]\n\t"
#define PKMUL(i) "v_pk_mul_f32 %[p" #i "], %[p" #i "], %[pm]\n\t"
#define PKADD(i) "v_pk_add_f32 %[p" #i "], %[p" #i "], %[pc]\n\t"
#define FMACDPP(i) "v_fmac_f32_dpp %[a" #i "], %[a" #i "], %[m] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define ADDDPP(i) "v_add_f32_dpp %[a" #i "], %[a" #i "], %[a" #i "] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define CNDADD(i) "v_cndmask_b32_dpp %[t], %[a" #i "], %[c], vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_add_f32 %[a" #i "], %[a" #i "], %[t]\n\t"
#define RCP(i) "v_rcp_f32 %[a" #i "], %[a" #i "]\n\t"
#define MADI24(i) "v_mad_i32_i24 %[a" #i "], %[a" #i "], %[m], %[c]\n\t"
#define CMPU64(i) "v_cmp_lt_u64 vcc, %[p" #i "], %[pm]\n\tv_addc_co_u32 %[a" #i "], vcc, 0, %[a" #i "], vcc\n\t"
#define CMPU32(i) "v_cmp_lt_u32 vcc, %[a" #i "], %[m]\n\tv_addc_co_u32 %[a" #i "], vcc, 0, %[a" #i "], vcc\n\t"

#define A_OPS [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7])
#define P_OPS [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7])

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float m = 0.999f, c = 0.001f; const f2 pm = {0.999f, 0.998f}, pc = {0.001f, 0.002f};
    float t = 0.f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(REP8(FMA) REP8(FMA) : A_OPS : [m] "v"(m), [c] "v"(c));
        else if (MODE == 1) asm volatile(REP8(PKFMA) REP8(PKFMA) : P_OPS : [pm] "v"(pm), [pc] "v"(pc));
        else if (MODE == 2) asm volatile(REP8(PKMUL) REP8(PKMUL) : P_OPS : [pm] "v"(pm));
        else if (MODE == 3) asm volatile(REP8(PKADD) REP8(PKADD) : P_OPS : [pc] "v"(pc));
        else if (MODE == 4) asm volatile(REP8(FMACDPP) REP8(FMACDPP) : A_OPS : [m] "v"(m));
        else if (MODE == 5) asm volatile(REP8(ADDDPP) REP8(ADDDPP) : A_OPS :);
        else if (MODE == 6) asm volatile(REP8(CNDADD) REP8(CNDADD) : [t] "+v"(t), A_OPS : [c] "v"(c) : "vcc");
        else if (MODE == 7) asm volatile(REP8(RCP) REP8(RCP) : A_OPS :);
        else if (MODE == 8) asm volatile(REP8(MADI24) REP8(MADI24) : A_OPS : [m] "v"(m), [c] "v"(c));
        else if (MODE == 9) asm volatile(REP8(CMPU64) REP8(CMPU64) : A_OPS : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]), [pm] "v"(pm) : "vcc");
        else if (MODE == 10) asm volatile(REP8(CMPU32) REP8(CMPU32) : A_OPS : [m] "v"(m) : "vcc");
    }
    float s = t;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, int per_iter) {
    float* out; hipMalloc(&out, 64);
    const int blocks = 256 * 8, iters = 4000;  // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * per_iter;
    const double cyc = ms * 1e-3 * 2.4e9 / (wave_instr / 1024.0);  // SIMD cycles per wave-instruction (1024 SIMDs, 2.4 GHz)
    printf("%-44s %8.3f ms  %5.2f SIMD cycles per wave-instruction\n", name, ms, cyc);
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32 (2 fma per lane)", 16);
    run<2>("v_pk_mul_f32", 16);
    run<3>("v_pk_add_f32", 16);
    run<4>("v_fmac_f32_dpp row_shr:1", 16);
    run<5>("v_add_f32_dpp row_shr:1", 16);
    run<6>("v_cndmask_b32_dpp + v_add_f32 (pair = 2)", 32);
    run<7>("v_rcp_f32", 16);
    run<8>("v_mad_i32_i24", 16);
    run<9>("v_cmp_lt_u64 + v_addc_co_u32 (pair = 2)", 32);
    run<10>("v_cmp_lt_u32 + v_addc_co_u32 (pair = 2)", 32);
    return 0;
}

// Micro-benchmark: what does ONE SIMD of gfx950 sustain in non-packed wave64 VALU instructions per cycle?
//
// The question ("What's weak" 4): DESIGN.md priced SQ_INSTS_VALU at 4 SIMD cycles per wave-instruction,
// MI355X_MICROARCH.md says "v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4".  round 1's valu_rates.hip divided
// wall time by an ASSUMED 2.4 GHz; under an all-CU VALU loop the chip may hold a lower clock, which would read as more
// cycles.  Here every figure is in shader cycles read inside the kernel (s_memtime), so the clock does not enter, and
// the clock the chip holds is reported beside it (s_memtime / s_memrealtime, the latter ticks at 100 MHz).
//
// Sweep: waves per SIMD 1 / 2 / 4 / 8 (256-thread blocks = one wave per SIMD each; k blocks per CU are forced by a
// dynamic LDS request of 160 KiB / k and a grid of 256 * k blocks) x independent chains per wave 1 / 2 / 4 / 8
// (chains = 1: every instruction depends on the previous one) x instruction (v_fma_f32, v_add_f32, v_mul_f32,
// v_mad_u32_u24 (integer), v_pk_fma_f32, v_fmac_f32_dpp).
//
// Per (wave): cycles = s_memtime(after) - s_memtime(before), N = instructions issued in between.  A SIMD hosting W
// such waves that all run concurrently issued W * N instructions in ~cycles, so
//     cycles per wave-instruction per SIMD = median(cycles) / (W * N).
// build + run:  hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define I_FMA(r) "v_fma_f32 %[" #r "], %[" #r "], %[m], %[c]\n\t"
#define I_ADD(r) "v_add_f32 %[" #r "], %[" #r "], %[c]\n\t"
#define I_MUL(r) "v_mul_f32 %[" #r "], %[" #r "], %[m]\n\t"
#define I_MAD24(r) "v_mad_u32_u24 %[" #r "], %[" #r "], %[m], %[c]\n\t"
#define I_DPP(r) "v_fmac_f32_dpp %[" #r "], %[" #r "], %[m] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define I_PK(r) "v_pk_fma_f32 %[" #r "], %[" #r "], %[pm], %[pc]\n\t"

// 16 instructions per block over CH independent chains (a0..a7)
#define B1(I) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0) I(a0)
#define B2(I) I(a0) I(a1) I(a0) I(a1) I(a0) I(a1) I(a0) I(a1) I(a0) I(a1) I(a0) I(a1) I(a0) I(a1) I(a0) I(a1)
#define B4(I) I(a0) I(a1) I(a2) I(a3) I(a0) I(a1) I(a2) I(a3) I(a0) I(a1) I(a2) I(a3) I(a0) I(a1) I(a2) I(a3)
#define B8(I) I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7) I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)

#define A_OPS [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7])
#define P_OPS [a0] "+v"(p[0]), [a1] "+v"(p[1]), [a2] "+v"(p[2]), [a3] "+v"(p[3]), [a4] "+v"(p[4]), [a5] "+v"(p[5]), [a6] "+v"(p[6]), [a7] "+v"(p[7])

enum { OP_FMA = 0, OP_ADD, OP_MUL, OP_MAD24, OP_DPP, OP_PK, N_OPS };
static const char* op_name[N_OPS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_mad_u32_u24", "v_fmac_f32_dpp", "v_pk_fma_f32"};

struct Stamp { unsigned long long cyc, real; };

template <int OP, int CH>
__global__ void __launch_bounds__(256) k(Stamp* out, int iters) {
    extern __shared__ char lds_pad[];  // its size fixes the blocks per CU
    float a[8]; f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    const float m = 0.999f, c = 0.001f; const f2 pm = {0.999f, 0.998f}, pc = {0.001f, 0.002f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the stamps have landed
    for (int it = 0; it < iters; it++) {
#define BODY(I, OPS, EXTRA)                                                        \
        if (CH == 1) asm volatile(B1(I) B1(I) B1(I) B1(I) : OPS : EXTRA);          \
        else if (CH == 2) asm volatile(B2(I) B2(I) B2(I) B2(I) : OPS : EXTRA);     \
        else if (CH == 4) asm volatile(B4(I) B4(I) B4(I) B4(I) : OPS : EXTRA);     \
        else asm volatile(B8(I) B8(I) B8(I) B8(I) : OPS : EXTRA);
#define EX_MC [m] "v"(m), [c] "v"(c)
#define EX_PK [pm] "v"(pm), [pc] "v"(pc)
        if (OP == OP_FMA) { BODY(I_FMA, A_OPS, EX_MC) }
        else if (OP == OP_ADD) { BODY(I_ADD, A_OPS, EX_MC) }
        else if (OP == OP_MUL) { BODY(I_MUL, A_OPS, EX_MC) }
        else if (OP == OP_MAD24) { BODY(I_MAD24, A_OPS, EX_MC) }
        else if (OP == OP_DPP) { BODY(I_DPP, A_OPS, EX_MC) }
        else { BODY(I_PK, P_OPS, EX_PK) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0].cyc = (unsigned long long)s;  // keeps the chains alive
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[wave].cyc = t1 - t0; out[wave].real = r1 - r0;
    }
}

template <int OP, int CH>
void run(int wps, FILE* f) {
    const int blocks = 256 * wps, iters = 2000, per_iter = 64;
    const size_t lds = (160 * 1024) / wps - 512;  // at most `wps` blocks fit a CU; the grid makes it exactly that many
    Stamp* out; hipMalloc(&out, sizeof(Stamp) * blocks * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, CH><<<blocks, 256, lds>>>(out, 50);  // warm-up
    hipEventRecord(e0);
    k<OP, CH><<<blocks, 256, lds>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks * 4);
    hipMemcpy(h.data(), out, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto& s : h) { cyc.push_back((double)s.cyc); clk.push_back(s.real ? (double)s.cyc / (double)s.real * 100.0 : 0.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double n = (double)iters * per_iter;
    const double med = cyc[cyc.size() / 2], mx = cyc.back();
    // wall-clock view of the same: the kernel's duration x the median clock / instructions per SIMD
    const double wall_cyc = ms * 1e-3 * clk[clk.size() / 2] * 1e6 / (n * wps);
    fprintf(f, "%-16s waves/SIMD %d  chains %d : %6.2f cycles per wave-instruction per SIMD (median wave; slowest wave %6.2f; from wall time %6.2f)"
               "  one wave's own rate %6.2f cyc/inst  clock %4.0f MHz  kernel %7.3f ms\n",
            op_name[OP], wps, CH, med / (n * wps), mx / (n * wps), wall_cyc, med / n, clk[clk.size() / 2], ms);
    fflush(f);
    hipFree(out);
}

template <int OP>
void sweep(FILE* f) {
    for (int wps : {1, 2, 4, 8}) {
        run<OP, 1>(wps, f); run<OP, 2>(wps, f); run<OP, 4>(wps, f); run<OP, 8>(wps, f);
    }
    fprintf(f, "\n");
}

int main(int argc, char** argv) {
    FILE* f = argc > 1 ? fopen(argv[1], "w") : stdout;
    if (!f) return 1;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    fprintf(f, "# %s, %d CUs, clockRate %d kHz; 64 instructions per loop iteration, 2000 iterations, 256-thread blocks\n",
            pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    sweep<OP_FMA>(f); sweep<OP_ADD>(f); sweep<OP_MUL>(f); sweep<OP_MAD24>(f); sweep<OP_DPP>(f); sweep<OP_PK>(f);
    if (f != stdout) fclose(f);
    return 0;
}

// Micro-benchmark: LDS atomic throughput on gfx950 (per-CU), distinct vs same addresses, float vs int.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int PATTERN>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ float sf[4096];
    unsigned* su = reinterpret_cast<unsigned*>(sf);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) sf[i] = 0.f;
    __syncthreads();
    // PATTERN 0: every lane its own address (conflict-free); 1: groups of 8 lanes share an address; 2: all 64 lanes share
    int idx = PATTERN == 0 ? tid : (PATTERN == 1 ? (tid >> 3) * 9 : (tid >> 6) * 33);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int a = (idx + u * 257) & 4095;
            if (MODE == 0) atomicAdd(&sf[a], 1.0f);
            else if (MODE == 1) atomicAdd(&su[a], 1u);
            else if (MODE == 2) atomicOr(&su[a], 1u << (lane & 31));
            else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(sf) + (a >> 1), 1ull);
            else if (MODE == 5) atomicAdd(reinterpret_cast<double*>(sf) + (a >> 1), 1.0);
            else sf[a] = (float)it;  // plain store for reference
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = sf[0];
}
template <int MODE, int PATTERN>
void run(const char* name) {
    float* out; hipMalloc(&out, 4096 * sizeof(float));
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, PATTERN><<<blocks, 256>>>(out, 10);
    hipEventRecord(a);
    k<MODE, PATTERN><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * 4 * iters * 8;          // wave-level atomic instructions
    const double per_cu_cycles = ms * 1e-3 * 2.4e9 / (wave_instr / 256.0);  // cycles per wave-instruction per CU
    printf("%-34s %8.3f ms  %6.1f cycles per wave-instruction per CU (%.2f per lane)\n", name, ms, per_cu_cycles, per_cu_cycles / 64.0);
    hipFree(out);
}
int main() {
    run<0, 0>("ds_add_f32 distinct");  run<0, 1>("ds_add_f32 8 lanes/address");  run<0, 2>("ds_add_f32 64 lanes/address");
    run<1, 0>("ds_add_u32 distinct");  run<1, 1>("ds_add_u32 8 lanes/address");  run<1, 2>("ds_add_u32 64 lanes/address");
    run<2, 0>("ds_or_b32  distinct");  run<2, 1>("ds_or_b32  8 lanes/address");  run<2, 2>("ds_or_b32  64 lanes/address");
    run<3, 0>("ds_write_b32 distinct");
    run<4, 0>("ds_add_u64 distinct");  run<4, 1>("ds_add_u64 8 lanes/address");  run<4, 2>("ds_add_u64 64 lanes/address");
    run<5, 0>("ds_add_f64 distinct");  run<5, 1>("ds_add_f64 8 lanes/address");
    return 0;
}

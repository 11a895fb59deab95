// Micro-benchmark: issue cost of a few VALU instructions on gfx950 (cycles per wave-instruction on one SIMD with W waves).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define ITER 2000
template <int OP> __global__ void k(uint32_t* out, uint64_t* cyc, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
    uint32_t b = seed | 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t*)&a[i & 6]) : "v"(b), "v"(b) : "vcc");
                if (OP == 6) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(a[i]));
                if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(uint64_t*)&a[i & 6]));
                if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(uint64_t*)&a[i & 6]));
                if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(uint64_t*)&a[i & 6]));
                if (OP == 10) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 11) asm volatile("v_lshl_add_u32 %0, %0, 6, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 12) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 13) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 15) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == 16) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 17) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 18) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 20) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 21) asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 22) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
                if (OP == 23) asm volatile("s_nop 0");
                if (OP == 24) asm volatile("v_pk_mov_b32 %0, %0, %0" : "+v"(*(uint64_t*)&a[i & 6]));
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name)
{
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    for (int waves = 1; waves <= 8; waves *= 2) { // waves per SIMD: block = waves * 4 wavefronts on one CU
        int threads = 64 * 4 * waves; if (threads > 1024) break;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<256, threads>>>(out, cyc, 12345);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<256, threads>>>(out, cyc, 12345);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double insts = (double)ITER * REP;
        // s_memtime-style counter runs at a fixed 100 MHz on some parts: report time-based cycles too (2.4 GHz assumed)
        printf("%-14s waves/SIMD %d: %.2f ns/inst/wave  => %.2f cyc@2.4GHz per wave-inst, SIMD throughput %.2f cyc/inst (counter %.1f/inst)\n",
               name, waves, ms * 1e6 / insts, ms * 1e6 / insts * 2.4, ms * 1e6 / insts * 2.4 / waves, (double)c / insts);
    }
}
int main()
{
    run<0>("v_mul_lo_u32"); run<1>("v_mul_u32_u24"); run<2>("v_fma_f32"); run<3>("v_rcp_f32"); run<4>("v_xor_b32");
    run<5>("v_mad_u64_u32"); run<6>("v_sqrt_f32"); run<7>("v_pk_mul_f32");
    run<8>("v_pk_fma_f32"); run<9>("v_pk_add_f32"); run<10>("v_add_u32"); run<11>("v_lshl_add_u32"); run<12>("v_max3_f32"); run<13>("v_min_f32");
    run<14>("v_cndmask_b32"); run<15>("v_cmp_le_f32"); run<16>("v_mul_f32"); run<17>("v_add_f32"); run<18>("v_mov_b32"); run<19>("v_fmac_f32");
    run<20>("v_and_b32"); run<21>("v_lshlrev_b32"); run<22>("v_readlane_b32"); run<23>("s_nop"); run<24>("v_pk_mov_b32");
    return 0;
}

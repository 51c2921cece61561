// Microbenchmark: issue cost (cycles per wavefront instruction) of the VALU instructions that make up the scale
// kernel's sweeps on gfx950, measured at 1 / 2 / 4 / 8 wavefronts per SIMD.  Eight independent register chains per
// wavefront, so that a result's latency is never waited for; the loop is timed with s_memtime (shader clock).
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIter = 2048;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void probe(unsigned long long *cycles, double *sink, double seed, int iseed) {
    double a[8], b = seed, c = seed * 0.5;
    int ia[8], ib = iseed;
    unsigned long long la[8];
    for (int j = 0; j < 8; ++j) { a[j] = seed + j + threadIdx.x; ia[j] = iseed + j + threadIdx.x; la[j] = (unsigned long long)(iseed + j) * 77ull + threadIdx.x; }
    unsigned long long sm = (unsigned long long)iseed * 0x0101010101010101ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < kIter; ++it) {
#define DO(j) \
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b)); \
        else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b)); \
        else if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); \
        else if (OP == 3) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc"); \
        else if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(la[j]) : "v"(la[(j + 1) & 7])); \
        else if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[j]) : "v"(ib) : ); \
        else if (OP == 6) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 8) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j])); \
        else if (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 10) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[j]) : "v"(b) : "vcc"); \
        else if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 12) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(la[j])); \
        else if (OP == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(ia[j]) : "v"(ib), "s"(sm)); \
        else if (OP == 15) asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ia[j]) : "s"(sm)); \
        else if (OP == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 17) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 18) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(ia[j]), "v"(ib) : "vcc"); \
        else if (OP == 19) asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(sm) : "v"(a[j]), "v"(b)); \
        else if (OP == 20) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(ia[j])); \
        else if (OP == 21) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(ia[j]) : "v"(ib)); \
        else if (OP == 22) asm volatile("v_mov_b64 %0, %1" : "+v"(a[j]) : "v"(b)); \
        else if (OP == 23) asm volatile("v_add_f64 %0, |%0|, -%1" : "+v"(a[j]) : "v"(b)); \
        else if (OP == 14) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
        REP8(DO) REP8(DO) REP8(DO) REP8(DO)
#undef DO
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0.0;
    for (int j = 0; j < 8; ++j) r += a[j] + ia[j] + (double)la[j];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r + (double)sm;
}

template <int OP>
static void run(const char *name) {
    unsigned long long *cyc;
    double *sink;
    hipMalloc(&cyc, sizeof(unsigned long long) * 4096);
    hipMalloc(&sink, sizeof(double) * 4096 * 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%-16s", name);
    for (int wps : {1, 2, 4}) {                      // wavefronts per SIMD: one workgroup of 4*wps wavefronts per CU... on one CU only
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;
        const int blocks = (64 * 4 * wps) / threads;    // all on few CUs: what matters is per-SIMD sharing, checked via the slowest wave
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, cyc, sink, 1.000001, 3);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        // ticks of s_memtime per instruction of ONE wave (32 instructions per trip); divide by waves sharing the SIMD for issue cost
        printf("  w/SIMD=%d: %7.2f ticks/instr/wave", wps, (double)mx / (kIter * 32.0));
    }
    printf("\n");
    hipFree(cyc); hipFree(sink);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<0>("v_add_f64"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<14>("v_max_f64"); run<3>("v_cmp_lt_f64");
    run<8>("v_rcp_f64"); run<10>("v_div_scale_f64");
    run<4>("v_lshl_add_u64"); run<12>("v_lshlrev_b64"); run<5>("v_cndmask_b32"); run<6>("v_max3_u32"); run<7>("v_lshl_add_u32");
    run<9>("v_add_u32"); run<11>("v_mov_b32"); run<13>("v_cndmask_e64 sgpr"); run<15>("v_cndmask 0,1,sgpr");
    run<16>("v_and_b32"); run<17>("v_and_or_b32"); run<18>("v_cmp_gt_u32 vcc"); run<19>("v_cmp_lt_f64 sgpr"); run<20>("v_lshlrev_b32 imm");
    run<21>("v_lshlrev_b32 v"); run<22>("v_mov_b64"); run<23>("v_add_f64 abs/neg");
    return 0;
}

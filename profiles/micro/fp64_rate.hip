// Microbenchmark: what the WHOLE chip sustains on fp64 vector FMAs / fp64 matrix FMAs, and at which shader clock
// (clock64 = shader clock counter, wall_clock64 = the constant 100 MHz counter).
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/fp64_rate.hip -o /tmp/fp64_rate && /tmp/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(double *sink, long long *clk, int iters, double seed) {
    double a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + j + threadIdx.x * 1e-3;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double b = seed * 0.5, c = 1e-9;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], b, c);
        } else if (MODE == 1) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b, acc1, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = a[j] * b + c;       // mul + add (two roundings)
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    s += acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
static void run(const char *name, int blocks, int iters, double flop_per_thread_iter) {
    double *sink; long long *clk;
    hipMalloc(&sink, sizeof(double) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    burn<MODE><<<blocks, 256>>>(sink, clk, iters / 10, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burn<MODE><<<blocks, 256>>>(sink, clk, iters, 1.0000001);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double sc = 0, wc = 0;
    for (int i = 0; i < blocks; ++i) { sc += h[2 * i]; wc += h[2 * i + 1]; }
    const double flops = flop_per_thread_iter * (double)iters * blocks * 256;
    printf("%-28s blocks %5d  %.3f ms  %.1f TFLOP/s   shader clock %.2f GHz (clock64 / wall_clock64 x 100 MHz)\n", name, blocks, ms,
           flops / ms * 1e-9, sc / wc * 0.1);
    hipFree(sink); hipFree(clk);
}

int main() {
    for (int blocks : {256, 2048, 8192}) {
        run<0>("v_fma_f64 (8 chains)", blocks, 20000, 16.0);
        run<2>("v_mul_f64 + v_add_f64", blocks, 20000, 16.0);
        run<1>("v_mfma_f64_16x16x4 (2 chains)", blocks, 20000, 2.0 * 2.0 * 16 * 16 * 4 / 64.0);
    }
    return 0;
}

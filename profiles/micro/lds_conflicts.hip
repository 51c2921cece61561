// Microbenchmark: cost of a wavefront's LDS gather (ds_read_b128 / ds_read_b64 / ds_add_u32) as a function of
// the lanes' address pattern, to learn which lanes share a conflict group on gfx950.
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/lds_conflicts.hip -o /tmp/lds_conflicts && /tmp/lds_conflicts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kN = 2048;          // records in LDS
constexpr int kIter = 4096;

template <int MODE>
__global__ void probe(const int *idx, unsigned long long *cycles, double *sink) {
    __shared__ double2 P[kN];
    __shared__ unsigned C[kN];
    for (int i = threadIdx.x; i < kN; i += blockDim.x) { double2 v; v.x = i; v.y = 2 * i; P[i] = v; C[i] = 0; }
    __syncthreads();
    int id = (idx[threadIdx.x & 63] + 64 * (threadIdx.x >> 6)) & (kN - 1);
    double acc = 0.0, acc2 = 0.0, acc3 = 0.0, acc4 = 0.0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; it += 4) {          // four independent accesses per trip: throughput, not latency
        const int i1 = (id + 512) & (kN - 1), i2 = (id + 1024) & (kN - 1), i3 = (id + 1536) & (kN - 1);
        if (MODE == 0) { const double2 a = P[id], b = P[i1], c = P[i2], d = P[i3]; acc += a.x + a.y; acc2 += b.x + b.y; acc3 += c.x + c.y; acc4 += d.x + d.y; }
        else if (MODE == 1) { const double *Q = reinterpret_cast<const double *>(P); acc += Q[id]; acc2 += Q[i1]; acc3 += Q[i2]; acc4 += Q[i3]; }
        else { atomicAdd(&C[id], 1u); atomicAdd(&C[i1], 1u); atomicAdd(&C[i2], 1u); atomicAdd(&C[i3], 1u); }
        id = (id + 64 * 8) & (kN - 1);          // same pattern class every iteration (mod 8 / 16 / 32 / 64 preserved)
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    acc += acc2 + acc3 + acc4;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + C[id & (kN - 1)];
}

int main() {
    struct Pat { const char *name; std::vector<int> idx; };
    std::vector<Pat> pats;
    auto mk = [&](const char *n, auto f) { Pat p; p.name = n; p.idx.resize(64); for (int l = 0; l < 64; ++l) p.idx[l] = f(l) & (kN - 1); pats.push_back(p); };
    srand(1);
    mk("sequential (lane)", [](int l) { return l; });
    mk("random", [](int) { return rand(); });
    mk("all same record", [](int) { return 5; });
    mk("stride 8 records (same 128-B column)", [](int l) { return l * 8; });
    mk("stride 16 records", [](int l) { return l * 16; });
    mk("stride 2 records", [](int l) { return l * 2; });
    mk("stride 4 records", [](int l) { return l * 4; });
    mk("random, distinct mod 8 within 8 consecutive lanes", [](int l) { return (rand() & ~7) | (l & 7); });
    mk("random, distinct mod 16 within 16 consecutive lanes", [](int l) { return (rand() & ~15) | (l & 15); });
    mk("random, distinct mod 32 within 32 consecutive lanes", [](int l) { return (rand() & ~31) | (l & 31); });
    mk("random, distinct mod 8 within lanes {l, l+8, ..}", [](int l) { return (rand() & ~7) | ((l >> 3) & 7); });
    mk("random, distinct mod 4 within 4 consecutive lanes", [](int l) { return (rand() & ~3) | (l & 3); });
    int *d_idx; unsigned long long *d_cyc; double *d_sink;
    hipMalloc(&d_idx, 64 * sizeof(int)); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 1024 * 8);
    const char *modes[3] = {"ds_read_b128", "ds_read_b64", "ds_add_u32"};
    for (int m = 0; m < 3; ++m) {
        printf("== %s: s_memtime ticks per wave-instruction, 16 wavefronts on one CU issuing independent accesses\n", modes[m]);
        for (auto &p : pats) {
            hipMemcpy(d_idx, p.idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 3; ++rep) {
                if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(1024), 0, 0, d_idx, d_cyc, d_sink);
                if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(1024), 0, 0, d_idx, d_cyc, d_sink);
                if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(1024), 0, 0, d_idx, d_cyc, d_sink);
                unsigned long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
                if (c < best) best = c;
            }
            printf("  %-56s %8.3f\n", p.name, (double)best / kIter / 16.0);
        }
    }
    return 0;
}

// Device check of asl_common.h's div_by(a, recip_of(d)) against a / d: log-uniform magnitudes within 2^+-LIM, both signs
// for a, positive and negative d, plus a = 0.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/div_test tools/div_test.hip
#include "../aprilslam_amd/csrc/asl_common.h"
#include <cstdio>
#include <cstdlib>

__device__ unsigned long long rng(unsigned long long &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

__global__ void k_div_test(unsigned long long seed, int per_thread, int lim, unsigned long long *bad, double *first)
{
    unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
    for (int i = 0; i < 8; i++) rng(s);
    unsigned long long nbad = 0;
    for (int i = 0; i < per_thread; i++) {
        const unsigned long long u = rng(s), v = rng(s);
        const int ea = (int)(u % (unsigned)(2 * lim + 1)) - lim, ed = (int)((u >> 20) % (unsigned)(2 * lim + 1)) - lim;
        double a = ldexp(1.0 + (double)(v & 0xFFFFFFFFFFFFFull) * 0x1p-52, ea);
        double d = ldexp(1.0 + (double)(rng(s) & 0xFFFFFFFFFFFFFull) * 0x1p-52, ed);
        if (u & (1ull << 60)) a = -a;
        if (u & (1ull << 61)) d = -d;
        if ((u >> 40) % 257 == 0) a = 0.0;
        const double want = a / d, got = div_by(a, recip_of(d));
        if (__double_as_longlong(want) != __double_as_longlong(got)) {
            if (nbad == 0 && atomicAdd(bad, 1ull) == 0) { first[0] = a; first[1] = d; first[2] = want; first[3] = got; }
            else if (nbad) atomicAdd(bad, 1ull);
            nbad++;
        }
    }
}

int main(int argc, char **argv)
{
    const int lim = argc > 1 ? atoi(argv[1]) : 100;
    unsigned long long *bad, hbad;
    double *first, hf[4];
    hipMalloc(&bad, 8); hipMalloc(&first, 32); hipMemset(bad, 0, 8);
    const int blocks = 4096, threads = 256, per = 4096;
    hipLaunchKernelGGL(k_div_test, dim3(blocks), dim3(threads), 0, 0, 12345ull, per, lim, bad, first);
    hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
    printf("div_by vs '/': %lld pairs with exponents within +-%d, %llu differ\n", (long long)blocks * threads * per, lim, hbad);
    if (hbad) printf("first: a=%a d=%a want=%a got=%a\n", hf[0], hf[1], hf[2], hf[3]);
    return hbad ? 1 : 0;
}

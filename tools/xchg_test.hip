// Diagnostic: the lane-exchange forms used by the in-register bitonic network (DPP quad_perm / row shifts / row_ror,
// v_permlane16_swap, v_permlane32_swap) against __shfl_xor, on the device.  hipcc --offload-arch=gfx950 tools/xchg_test.hip -o build/xchg_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../aprilslam_amd/csrc/k_xchg.inc"

__global__ void k(unsigned long long *out, const unsigned long long *in)
{
    const unsigned long long v = in[threadIdx.x];
    out[0 * 64 + threadIdx.x] = lane_xor<1>(v);
    out[1 * 64 + threadIdx.x] = lane_xor<2>(v);
    out[2 * 64 + threadIdx.x] = lane_xor<4>(v);
    out[3 * 64 + threadIdx.x] = lane_xor<8>(v);
    out[4 * 64 + threadIdx.x] = lane_xor<16>(v);
    out[5 * 64 + threadIdx.x] = lane_xor<32>(v);
}

int main()
{
    unsigned long long h[64], o[6 * 64], *di, *dout;
    for (int i = 0; i < 64; i++) h[i] = 0x0123456700000000ull * (unsigned long long)(i + 1) + (unsigned long long)i * 0x10001ull + 7;
    hipMalloc((void **)&di, sizeof h); hipMalloc((void **)&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < 6; s++)
        for (int i = 0; i < 64; i++)
            if (o[s * 64 + i] != h[i ^ (1 << s)]) { if (bad < 8) printf("xor %d lane %d: got %llx want %llx\n", 1 << s, i, o[s * 64 + i], h[i ^ (1 << s)]); bad++; }
    printf(bad ? "FAILED (%d)\n" : "lane_xor ok%.0d\n", bad);
    return bad != 0;
}

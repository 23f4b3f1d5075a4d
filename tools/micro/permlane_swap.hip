// v_permlane32_swap_b32 as the attention kernels use it (adf_kernels.hip): max(sw[0], sw[1]) of swap(v, v) must be max(v[lane], v[lane ^ 32]) in every lane.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/permlane_swap.hip -o tools/micro/bin/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out_swap, float* out_shfl) {
    const float v = in[threadIdx.x];
    const unsigned u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out_swap[threadIdx.x] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    out_shfl[threadIdx.x] = fmaxf(v, __shfl_xor(v, 32, 64));
}
int main() {
    float h[64], a[64], b[64], *din, *da, *db;
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37 + 11) % 64) - 20.5f;
    hipMalloc(&din, 256); hipMalloc(&da, 256); hipMalloc(&db, 256);
    hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, da, db);
    hipMemcpy(a, da, 256, hipMemcpyDeviceToHost); hipMemcpy(b, db, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { const float want = h[i] > h[i ^ 32] ? h[i] : h[i ^ 32]; if (a[i] != want || b[i] != want) ++bad; }
    printf("permlane32_swap max across lane halves: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad != 0;
}

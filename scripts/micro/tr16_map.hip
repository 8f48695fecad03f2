// ds_read_b64_tr_b16 lane mapping on gfx950 (the ISA text is not on this box): every lane reads 4 u16 at element offset
// 4 * lane of an LDS array holding lds[e] = e; the values a lane receives name their source (lane = v / 4, element = v % 4).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/tr16_map.hip -o scripts/micro/tr16_map
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int e = threadIdx.x; e < 1024; e += 64) lds[e] = (unsigned short)e;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (lane %2d, elem %d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
        printf("\n");
    }
    return 0;
}

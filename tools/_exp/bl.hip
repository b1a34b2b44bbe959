#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* x, unsigned bytes, float* y, const unsigned* voffs, unsigned soff) {
    __shared__ float lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voffs[threadIdx.x], soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) y[i] = lds[i];
}
int main() {
    float *x, *y; unsigned* o;
    float hx[64]; for (int i = 0; i < 64; ++i) hx[i] = 100 + i;
    unsigned ho[64];
    for (int i = 0; i < 64; ++i) ho[i] = 0xFFFFFFF0u;
    ho[0] = 4; ho[1] = 20; ho[2] = 252; ho[3] = 0xFFFFFFFCu; ho[4] = 248;
    (void)hipMalloc(&x, 4096); (void)hipMalloc(&y, 1024); (void)hipMalloc(&o, 256);
    (void)hipMemcpy(x + 16, hx, 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(o, ho, 256, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 8u, 240u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x + 16, 256u, y, o, soff);
        float hy[256]; (void)hipMemcpy(hy, y, 1024, hipMemcpyDeviceToHost);
        printf("soffset %u (buffer = 64 floats 100..163, 256 B):\n", soff);
        for (int t = 0; t < 6; ++t) printf("  lane %d voff %u: %g %g %g %g\n", t, ho[t], hy[4*t], hy[4*t+1], hy[4*t+2], hy[4*t+3]);
    }
    return 0;
}

// Microbenchmark (gfx950): cost of LDS float / integer atomic adds (no return) per wave instruction per CU, against
// ds_write_b32, for the address patterns of the cost-volume source-tile sweep (cv_src_grad_kernel):
//   linear   lane l -> dword l (conflict-free)
//   quad     lane 4 j + c -> c * 262 * 4 + j      (the sweep's (pixel, quarter) lanes on 262-float accumulator rows)
//   same     every lane the same dword
//   pair     lanes 2 i, 2 i + 1 the same dword (two pixels sharing a texel)
#include <hip/hip_runtime.h>
#include <stdio.h>
enum { kF32Add, kU32Add, kWrite, kRmw };
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, int pattern)
{
    __shared__ float lds[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int a0;
    if (pattern == 0) a0 = lane;
    else if (pattern == 1) a0 = (lane & 3) * 262 * 4 + (lane >> 2);
    else if (pattern == 2) a0 = 5;
    else a0 = lane >> 1;
    a0 += wave * 64;   // (waves on different rows: as in the sweep, where they walk different pixels)
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            float* p = lds + a0 + j * 262;
            if (OP == kF32Add) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == kU32Add) __hip_atomic_fetch_add((unsigned*)p, (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == kWrite) *(volatile float*)p = v;
            if (OP == kRmw) { const float o = *(volatile float*)p; *(volatile float*)p = o + v; }
        }
        v += 1.0f;
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
template <int OP>
void run(const char* name, float* d)
{
    const char* pat[4] = {"linear", "quad", "same", "pair"};
    for (int p = 0; p < 4; ++p)
        for (int w = 1; w <= 3; w += 2) {   // workgroups (of 4 wavefronts) per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 2000, blocks = 256 * w;
            k<OP><<<blocks, 256>>>(d, 10, p); hipDeviceSynchronize();
            hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, p); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-10s %-7s workgroups/CU=%d: %.3f ms -> %.1f cycles (2.4 GHz) per wave instruction per CU\n", name, pat[p], w, ms,
                   ms * 2.4e6 / (iters * 12.0 * 4 * w));
        }
}
int main()
{
    float* d; hipMalloc(&d, 256 * 3 * 256 * 4);
    run<kF32Add>("ds_add_f32", d); run<kU32Add>("ds_add_u32", d); run<kWrite>("ds_write", d); run<kRmw>("read+write", d);
    return 0;
}

// Microbenchmark (gfx950): cost of wave-uniform (broadcast) ds_read_b128 / b64 / b32, per CU, with 1..8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int W>  // dwords per read
__global__ __launch_bounds__(64) void k(float* out, int iters, int stride)
{
    __shared__ float4 lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    float acc = 0;
    int idx = (blockIdx.x * 7) & 255;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int a = (idx + j * stride) & 511;       // wave-uniform address: every lane reads the same 16 bytes
            if (W == 4) { const float4 v = lds[a]; acc += v.x + v.w; }
            if (W == 2) { const float2 v = ((const float2*)lds)[a]; acc += v.x + v.y; }
            if (W == 1) { acc += ((const float*)lds)[a]; }
        }
        idx = (idx + 17) & 255;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int W>
void run(const char* name, float* d)
{
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000, blocks = 256 * 4 * w;   // single-wave workgroups: 4 * w per CU
        k<W><<<blocks, 64>>>(d, 10, 3); hipDeviceSynchronize();
        hipEventRecord(e0); k<W><<<blocks, 64>>>(d, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s waves/SIMD=%d: %.3f ms -> %.2f cycles (2.4 GHz) per read per CU\n", name, w, ms, ms * 2.4e6 / (iters * 16.0 * 4 * w));
    }
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    run<4>("ds_read_b128", d); run<2>("ds_read_b64", d); run<1>("ds_read_b32", d);
    return 0;
}

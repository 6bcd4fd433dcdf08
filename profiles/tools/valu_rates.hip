// Microbenchmark (gfx950): issue cost of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_cndmask with 1..8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float v[16]; f32x2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = seed + i + threadIdx.x; p[i] = (f32x2){v[i], v[i] + 1.0f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (MODE == 0) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
            if (MODE == 1) p[j & 15] = __builtin_elementwise_fma(p[j & 15], (f32x2){1.0001f, 1.0002f}, (f32x2){0.5f, 0.25f});
            if (MODE == 2) v[j & 15] = __builtin_amdgcn_exp2f(v[j & 15]);
            if (MODE == 3) v[j & 15] = v[(j + 1) & 15] > 0.5f ? v[j & 15] : seed;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d)
{
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000;
        k<MODE><<<256 * w, 256>>>(d, 10, 1.0f); hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE><<<256 * w, 256>>>(d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s waves/SIMD=%d: %.3f ms -> %.2f cycles (2.4 GHz) per instruction per wave, %.2f per SIMD-instruction\n", name, w, ms,
               ms * 2.4e6 / (iters * 64.0), ms * 2.4e6 / (iters * 64.0 * w));
    }
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_exp_f32", d); run<3>("cmp+cndmask", d);
    return 0;
}

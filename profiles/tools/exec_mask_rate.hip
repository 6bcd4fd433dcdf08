// Microbenchmark (gfx950): does a wave64 VALU instruction get cheaper when part of EXEC is off?
//   full wave, lower 32 lanes, lower 16 lanes, even lanes only -- v_fma_f32 and v_pk_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/emr profiles/tools/exec_mask_rate.hip ; run: /tmp/emr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int MASK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float v[16]; f32x2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = seed + i + threadIdx.x; p[i] = (f32x2){v[i], v[i] + 1.0f}; }
    const int lane = threadIdx.x & 63;
    const bool on = MASK == 0 ? true : MASK == 1 ? lane < 32 : MASK == 2 ? lane < 16 : MASK == 3 ? (lane & 1) == 0 : lane >= 32;
    if (on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (MODE == 0) v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
                if (MODE == 1) p[j & 15] = __builtin_elementwise_fma(p[j & 15], (f32x2){1.0001f, 1.0002f}, (f32x2){0.5f, 0.25f});
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int MASK>
void run(const char* name, float* d)
{
    for (int w = 1; w <= 4; w *= 4) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000;
        k<MODE, MASK><<<256 * w, 256>>>(d, 10, 1.0f); hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE, MASK><<<256 * w, 256>>>(d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s waves/SIMD=%d: %.3f ms -> %.2f cycles (2.4 GHz) per SIMD-instruction\n", name, w, ms, ms * 2.4e6 / (iters * 64.0 * w));
    }
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0, 0>("v_fma_f32 full", d); run<0, 1>("v_fma_f32 lanes 0-31", d); run<0, 4>("v_fma_f32 lanes 32-63", d);
    run<0, 2>("v_fma_f32 lanes 0-15", d); run<0, 3>("v_fma_f32 even lanes", d);
    run<1, 0>("v_pk_fma_f32 full", d); run<1, 1>("v_pk_fma_f32 lanes 0-31", d); run<1, 2>("v_pk_fma_f32 lanes 0-15", d);
    run<1, 3>("v_pk_fma_f32 even lanes", d);
    return 0;
}

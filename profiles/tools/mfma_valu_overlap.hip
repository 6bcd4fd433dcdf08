// Microbenchmark (gfx950): do VALU instructions overlap a dependent v_mfma_f32_32x32x2_f32 chain
//   (a) inside ONE wavefront,  (b) between the 2..4 wavefronts of a SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo profiles/tools/mfma_valu_overlap.hip ; run: /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed * r;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);   // dependent chain
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);   // independent VALU
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, int NV>
float run(int waves_per_simd, int iters, float* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    k<NM, NV><<<blocks, 256>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, NV><<<blocks, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 2000;
    printf("per (MFMA + NV VALU) step, cycles at 2.4 GHz = ms*2.4e6/(iters*NM)\n");
#define R(NM, NV) for (int w = 1; w <= 4; w *= 2) { float ms = run<NM, NV>(w, iters, d); \
        printf("NM=%d NV=%2d waves/SIMD=%d: %.3f ms  -> %.1f cycles per step per wave-slot-set (MFMA alone = 64, VALU alone = %d)\n", NM, NV, w, ms, ms * 2.4e6 / (iters * NM) , 4 * NV); }
    R(16, 0) R(16, 4) R(16, 8) R(16, 12) R(16, 15) R(16, 24) R(16, 32)
    return 0;
}

// Microbenchmark (gfx950): what does one scattered dwordx4 wave load of the plane sweep cost per CU, as a function of how
// the 64 lanes' 16-byte chunks fall into 128-byte lines?  A source texel is a 192-byte record (48 fp32 channels); a wave
// gathers the records of 32 neighbouring pixels (texel index = pixel + a slowly moving offset, as along a sweep).
//   A  lane = (pixel p, parity hf), the record stored [parity][24]: instruction s reads bytes [96 hf + 16 s, +16) of
//      record(p) -- the two lanes of a pixel sit 96 bytes apart (today's layout): ~56 lines per instruction
//   B  same lanes, record stored [6][parity][4]: bytes [32 s + 16 hf, +16): the two lanes of a pixel are adjacent: 32 lines
//   C  lane = (pixel j of 16, quarter c): bytes [64 s + 16 c, +16) of record(j), 3 instructions per 16 pixels: 16 lines
//      per instruction, 64 contiguous bytes per pixel (the 16x16x4 MFMA operand layout)
//   D  today's lanes (pixel p of 32, parity hf) on a CHUNK-PLANAR map [12 chunks][texel][4 floats]: instruction s reads
//      chunk 6 hf + s of texel(p) -- neighbouring lanes = neighbouring pixels read neighbouring 16-byte chunks when they
//      sample neighbouring texels (STEP/4 texels per pixel: 4 = the source at the pixel grid's resolution)
// Per (pixel, tap) A, B and D issue 6/32 instructions, C 3/16: the same.  Output: ns per wave instruction per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/tpr profiles/tools/tap_pattern_rate.hip ; run: /tmp/tpr
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int PAT, int STEP = 4>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, int nrec, int iters, float* out)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned pos = (unsigned)wave * 977u;
    for (int it = 0; it < iters; ++it) {
        pos += (it & 3) == 3 ? 61u : 1u;                     // next tap: one texel on; every fourth: another row
        if (PAT == 2) {
            const int j = lane >> 2, c = lane & 3;
#pragma unroll
            for (int half = 0; half < 2; ++half) {           // two groups of 16 pixels = the 32 pixels of A / B
                const unsigned rec = (pos + 16 * half + j) % (unsigned)nrec;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float4 v = src[(size_t)rec * 12 + 4 * s + c];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        } else if (PAT == 3) {
            const int p = lane & 31, hf = lane >> 5;
            const unsigned rec = (pos + (unsigned)(p * STEP) / 4u) % (unsigned)nrec;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float4 v = src[(size_t)(6 * hf + s) * nrec + rec];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        } else {
            const int p = lane & 31, hf = lane >> 5;
            const unsigned rec = (pos + (unsigned)(p * STEP) / 4u) % (unsigned)nrec;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float4 v = src[(size_t)rec * 12 + (PAT == 0 ? 6 * hf + s : 2 * s + hf)];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int PAT, int STEP = 4>
void run(const char* name, const float4* src, int nrec, float* out, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 4096;  // 4 waves per workgroup: one per SIMD
    k<PAT, STEP><<<blocks, 256>>>(src, nrec, 64, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<PAT, STEP><<<blocks, 256>>>(src, nrec, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)blocks * 4 * iters * 6 / 256.0;
    printf("%-28s waves/SIMD %d  footprint %6.1f MB  %7.3f ms  %6.2f ns per wave instruction per CU\n", name, waves_per_simd,
           nrec * 192.0 / 1e6, ms, ms * 1e6 / instr_per_cu);
}

int main()
{
    float* out; hipMalloc(&out, 4);
    for (int nrec : {96 * 128, 242 * 324 * 2}) {             // one native source map (2.4 MB); config-3 scale, K = 2 (30 MB)
        float4* src; hipMalloc(&src, (size_t)nrec * 192);
        hipMemset(src, 0, (size_t)nrec * 192);
        for (int w : {2}) {
            run<0>("A [parity][24] (today)", src, nrec, out, w);
            run<1>("B [6][parity][4]", src, nrec, out, w);
            run<2>("C 16 px x 4 quarters", src, nrec, out, w);
            run<3, 4>("D planar, 1 texel/pixel", src, nrec, out, w);
            run<3, 3>("D planar, 0.75 texel/pixel", src, nrec, out, w);
            run<3, 8>("D planar, 2 texels/pixel", src, nrec, out, w);
            run<3, 16>("D planar, 4 texels/pixel", src, nrec, out, w);
            run<0, 16>("A today, 4 texels/pixel", src, nrec, out, w);
        }
        hipFree(src);
    }
    return 0;
}

// Microbenchmark (gfx950): throughput of no-return fp32 global atomics at agent scope (atomicAdd: executed memory-side,
// the line leaves the XCD's L2) vs workgroup scope into a buffer PRIVATE to the issuing XCD (blockIdx.x % 8): do they
// stay in that XCD's L2, and are the sums still exact when many workgroups of the XCD hit the same words?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/asr profiles/tools/atomic_scope_rate.hip ; run: /tmp/asr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int SCOPE>   // 0: agent (atomicAdd), 1: workgroup scope
__global__ __launch_bounds__(256) void k(float* buf, int words_per_copy, int iters, int private_copies)
{
    const int xcd = blockIdx.x & 7;
    float* base = buf + (private_copies ? (size_t)xcd * words_per_copy : 0);
    // lane = channel of a 48-float record (like the cost-volume scatter): 64 lanes, 48 active, record index pseudo-random
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
    const int nrec = words_per_copy / 48;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const int rec = (int)((s >> 8) % (unsigned)nrec);
        if (lane < 48) {
            float* p = base + (size_t)rec * 48 + lane;
            if (SCOPE == 0) atomicAdd(p, 1.0f);
            else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <int SCOPE>
void run(const char* name, int private_copies)
{
    const int words = 96 * 128 * 48 * 2;          // one native source-gradient map pair: 4.7 MB
    const int copies = private_copies ? 8 : 1;
    float* d; hipMalloc(&d, (size_t)words * copies * 4);
    hipMemset(d, 0, (size_t)words * copies * 4);
    const int blocks = 1024, iters = 2000;
    k<SCOPE><<<blocks, 256>>>(d, words, 10, private_copies);
    hipDeviceSynchronize();
    hipMemset(d, 0, (size_t)words * copies * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<SCOPE><<<blocks, 256>>>(d, words, iters, private_copies);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h((size_t)words * copies);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (float v : h) sum += v;
    const double expect = (double)blocks * 4 * iters * 48;
    printf("%-44s %.3f ms  %.2f G record-atomics/s (192 B each)   sum %.0f expected %.0f %s\n", name, ms,
           blocks * 4.0 * iters / ms / 1e6, sum, expect, sum == expect ? "OK" : "MISMATCH");
    hipFree(d);
}
int main()
{
    run<0>("agent scope, one shared buffer", 0);
    run<0>("agent scope, per-XCD private buffers", 1);
    run<1>("workgroup scope, per-XCD private buffers", 1);
    run<1>("workgroup scope, one shared buffer (WRONG?)", 0);
    return 0;
}

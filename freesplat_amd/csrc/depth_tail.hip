// depth_tail.hip -- the depth-regression tail of FreeSplat's DepthDecoder, fused.
//
// Replaces src/model/encoder/modules/networks.py:130-152: per scale
//     p = softmax(logits, dim=1);  E = sum_d cand_d p_d  (expected log-depth / inverse depth);
//     depth = exp(E)  (log planes)  |  1 / E,
// and for the finest scale additionally
//     fine = bilinear x2 (align_corners=True) of E;  depth_map = exp(fine) | 1 / fine;
//     depth_weights = max_d bilinear x2 of p_d.
// The reference materialises the softmax ([B,D,h/2,w/2]) and its x2-upsampled copy ([B,D,h,w], 200 MB at
// the native 2 x 128 x 384 x 512) just to take a max; here the logits are read once for the expectation
// (online softmax) and the upsampled probabilities are formed on the fly from the 4 neighbours' logits
// and softmax statistics (per tile of fine pixels, through LDS).  HBM-bound: algorithmic bytes = 4 * B*D*h2*w2 (logits) + O(B*h*w).
#include "fs_common.h"

namespace fs {

// one thread per coarse pixel: online softmax statistics (m, s) and expectation
__global__ __launch_bounds__(256) void depth_expect_kernel(int B, int D, int hw, const float* __restrict__ logits,
                                                           const float* __restrict__ cand, int log_planes,
                                                           float* __restrict__ stats, float* __restrict__ coarse,
                                                           float* __restrict__ depth)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * hw) return;
    const int b = (int)(e / hw), p = (int)(e % hw);
    const float* l = logits + (size_t)b * D * hw + p;
    float m = -3.0e38f, s = 0.0f, acc = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float v = l[(size_t)d * hw];
        if (v > m) {
            const float r = expf(m - v);
            s *= r; acc *= r; m = v;
        }
        const float ex = expf(v - m);
        s += ex;
        acc += cand[d] * ex;
    }
    const float E = acc / s;
    stats[(size_t)b * 2 * hw + p] = m;
    stats[(size_t)b * 2 * hw + hw + p] = s;
    coarse[e] = E;
    depth[e] = log_planes ? expf(E) : 1.0f / E;
}

struct Bilin { int i00, i01, i10, i11; float w00, w01, w10, w11; };
__device__ __forceinline__ Bilin bilin_x2(int y, int x, int h2, int w2)
{
    // F.interpolate(scale_factor=2, mode="bilinear", align_corners=True): src = dst * (in-1)/(out-1)
    const float sy = h2 > 1 ? (float)y * ((float)(h2 - 1) / (float)(2 * h2 - 1)) : 0.0f;
    const float sx = w2 > 1 ? (float)x * ((float)(w2 - 1) / (float)(2 * w2 - 1)) : 0.0f;
    const int y0 = min((int)sy, h2 - 1), x0 = min((int)sx, w2 - 1);
    const int y1 = min(y0 + 1, h2 - 1), x1 = min(x0 + 1, w2 - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    Bilin o;
    o.i00 = y0 * w2 + x0; o.i01 = y0 * w2 + x1; o.i10 = y1 * w2 + x0; o.i11 = y1 * w2 + x1;
    o.w00 = (1.0f - fy) * (1.0f - fx); o.w01 = (1.0f - fy) * fx; o.w10 = fy * (1.0f - fx); o.w11 = fy * fx;
    return o;
}

// One workgroup per 32 x 8 tile of fine pixels.  The tile's coarse footprint (<= 18 x 6 pixels) has its
// probabilities exp(l - m) / s formed ONCE per plane in LDS, 16 planes at a time, and every fine pixel then takes its
// bilinear combination from LDS: 0.4 exponentials per fine pixel and plane instead of 4, and the logits are read
// once per tile footprint instead of four times per fine pixel.
constexpr int kUpW = 32, kUpH = 8, kUpPlanes = 16, kUpPatch = 18 * 6;
__global__ __launch_bounds__(256) void depth_upsample_kernel(int B, int D, int h2, int w2,
                                                             const float* __restrict__ logits,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ coarse, int log_planes,
                                                             float* __restrict__ depth_map,
                                                             float* __restrict__ depth_w, int32_t* __restrict__ argmax)
{
    __shared__ float s_p[kUpPlanes][kUpPatch];
    __shared__ float s_m[kUpPatch], s_rs[kUpPatch];
    const int H = 2 * h2, W = 2 * w2, hw = h2 * w2;
    const int b = blockIdx.z, t = threadIdx.x;
    const int X0 = blockIdx.x * kUpW, Y0 = blockIdx.y * kUpH;
    const int x = X0 + (t & 31), y = Y0 + (t >> 5);
    const bool in = x < W && y < H;
    // coarse footprint of the tile (same source-coordinate arithmetic as bilin_x2)
    const float ry = h2 > 1 ? (float)(h2 - 1) / (float)(2 * h2 - 1) : 0.0f, rx = w2 > 1 ? (float)(w2 - 1) / (float)(2 * w2 - 1) : 0.0f;
    const int cy0 = min((int)((float)Y0 * ry), h2 - 1), cx0 = min((int)((float)X0 * rx), w2 - 1);
    const int cy1 = min(min((int)((float)min(Y0 + kUpH - 1, H - 1) * ry), h2 - 1) + 1, h2 - 1);
    const int cx1 = min(min((int)((float)min(X0 + kUpW - 1, W - 1) * rx), w2 - 1) + 1, w2 - 1);
    const int pw = cx1 - cx0 + 1, ph = cy1 - cy0 + 1, np = pw * ph;  // <= 18 x 6
    const float* m = stats + (size_t)b * 2 * hw;
    const float* s = m + hw;
    for (int k = t; k < np; k += 256) {
        const int ci = (cy0 + k / pw) * w2 + cx0 + k % pw;
        s_m[k] = m[ci];
        s_rs[k] = 1.0f / s[ci];
    }
    // (threads past the image edge take the taps of the nearest inside pixel of THIS tile: their patch-local
    // indices must stay inside the LDS patch; their results are never stored)
    const Bilin q = bilin_x2(min(y, H - 1), min(x, W - 1), h2, w2);
    // patch-local indices of the four taps
    const int l00 = (q.i00 / w2 - cy0) * pw + (q.i00 % w2 - cx0), l01 = (q.i01 / w2 - cy0) * pw + (q.i01 % w2 - cx0);
    const int l10 = (q.i10 / w2 - cy0) * pw + (q.i10 % w2 - cx0), l11 = (q.i11 / w2 - cy0) * pw + (q.i11 % w2 - cx0);
    const float* l = logits + (size_t)b * D * hw;
    float best = -1.0f;
    int bi = 0;
    for (int d0 = 0; d0 < D; d0 += kUpPlanes) {
        __syncthreads();  // (also orders the s_m / s_rs fill before their first use)
        const int nd = min(kUpPlanes, D - d0);
        for (int k = t; k < nd * np; k += 256) {
            const int dd = k / np, c = k % np;
            const int ci = (cy0 + c / pw) * w2 + cx0 + c % pw;
            s_p[dd][c] = expf(l[(size_t)(d0 + dd) * hw + ci] - s_m[c]) * s_rs[c];
        }
        __syncthreads();
        for (int dd = 0; dd < nd; ++dd) {
            const float v = q.w00 * s_p[dd][l00] + q.w01 * s_p[dd][l01] + q.w10 * s_p[dd][l10] + q.w11 * s_p[dd][l11];
            if (v > best) { best = v; bi = d0 + dd; }
        }
    }
    if (in) {
        const float* c = coarse + (size_t)b * hw;
        const float fine = q.w00 * c[q.i00] + q.w01 * c[q.i01] + q.w10 * c[q.i10] + q.w11 * c[q.i11];
        const size_t e = ((size_t)b * H + y) * W + x;
        depth_map[e] = log_planes ? expf(fine) : 1.0f / fine;
        depth_w[e] = best;
        argmax[e] = bi;
    }
}

// ---- backward ----
// fine pixels: d depth_map -> d fine -> 4 coarse neighbours; d depth_weights -> probabilities (arg max plane)
__global__ __launch_bounds__(256) void depth_upsample_bwd_kernel(int B, int D, int h2, int w2,
                                                                 const float* __restrict__ depth_map,
                                                                 const int32_t* __restrict__ argmax, int log_planes,
                                                                 const float* __restrict__ g_map,
                                                                 const float* __restrict__ g_w,
                                                                 float* __restrict__ gE, float* __restrict__ g_prob)
{
    const int H = 2 * h2, W = 2 * w2, hw = h2 * w2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * H * W) return;
    const int b = (int)(e / ((long long)H * W)), r = (int)(e % ((long long)H * W));
    const Bilin q = bilin_x2(r / W, r % W, h2, w2);
    if (g_map) {
        const float dm = depth_map[e];
        const float gf = log_planes ? g_map[e] * dm : -g_map[e] * dm * dm;  // d exp(f) = exp(f); d(1/f) = -1/f^2
        float* g = gE + (size_t)b * hw;
        atomicAdd(g + q.i00, q.w00 * gf); atomicAdd(g + q.i01, q.w01 * gf);
        atomicAdd(g + q.i10, q.w10 * gf); atomicAdd(g + q.i11, q.w11 * gf);
    }
    if (g_w) {
        float* gp = g_prob + ((size_t)b * D + argmax[e]) * hw;
        const float gw = g_w[e];
        atomicAdd(gp + q.i00, q.w00 * gw); atomicAdd(gp + q.i01, q.w01 * gw);
        atomicAdd(gp + q.i10, q.w10 * gw); atomicAdd(gp + q.i11, q.w11 * gw);
    }
}

// coarse pixels: softmax-expectation backward, dl_d = p_d [ (cand_d - E) gE + g_prob_d - sum_j p_j g_prob_j ]
__global__ __launch_bounds__(256) void depth_expect_bwd_kernel(int B, int D, int hw, const float* __restrict__ logits,
                                                               const float* __restrict__ cand, int log_planes,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ coarse,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ g_coarse,
                                                               const float* __restrict__ g_depth,
                                                               const float* __restrict__ gE_up,
                                                               const float* __restrict__ g_prob,
                                                               float* __restrict__ g_logits)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * hw) return;
    const int b = (int)(e / hw), p = (int)(e % hw);
    const float m = stats[(size_t)b * 2 * hw + p], s = stats[(size_t)b * 2 * hw + hw + p];
    const float E = coarse[e];
    float gE = gE_up ? gE_up[e] : 0.0f;
    if (g_coarse) gE += g_coarse[e];
    if (g_depth) gE += log_planes ? g_depth[e] * depth[e] : -g_depth[e] * depth[e] * depth[e];
    const float* l = logits + (size_t)b * D * hw + p;
    const float* gp = g_prob ? g_prob + (size_t)b * D * hw + p : nullptr;
    float dotp = 0.0f;
    if (gp)
        for (int d = 0; d < D; ++d) dotp += expf(l[(size_t)d * hw] - m) / s * gp[(size_t)d * hw];
    float* go = g_logits + (size_t)b * D * hw + p;
    for (int d = 0; d < D; ++d) {
        const float pd = expf(l[(size_t)d * hw] - m) / s;
        const float gpd = gp ? gp[(size_t)d * hw] : 0.0f;
        go[(size_t)d * hw] = pd * ((cand[d] - E) * gE + gpd - dotp);
    }
}

}  // namespace fs

using namespace fs;

FS_API int fs_depth_tail_forward(int32_t B, int32_t D, int32_t h2, int32_t w2, const float* logits,
                                 const float* candidates, int32_t log_planes, float* stats, float* coarse,
                                 float* depth, float* depth_map, float* depth_weights, int32_t* argmax, void* stream_)
{
    if (B <= 0 || D <= 0 || h2 <= 0 || w2 <= 0 || !logits || !candidates || !stats || !coarse || !depth)
        return FS_ERR_INVALID_ARG;
    if ((depth_map == nullptr) != (depth_weights == nullptr) || (depth_map == nullptr) != (argmax == nullptr))
        return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    const long long n = (long long)B * h2 * w2;
    hipLaunchKernelGGL(depth_expect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B, D, h2 * w2, logits,
                       candidates, log_planes, stats, coarse, depth);
    if (depth_map)
        hipLaunchKernelGGL(depth_upsample_kernel, dim3((2 * w2 + kUpW - 1) / kUpW, (2 * h2 + kUpH - 1) / kUpH, B), dim3(256),
                           0, st, B, D, h2, w2, logits, stats, coarse, log_planes, depth_map, depth_weights, argmax);
    FS_CHECK_LAUNCH("depth_tail_forward");
    return FS_OK;
}

FS_API int fs_depth_tail_backward(int32_t B, int32_t D, int32_t h2, int32_t w2, const float* logits,
                                  const float* candidates, int32_t log_planes, const float* stats,
                                  const float* coarse, const float* depth, const float* depth_map,
                                  const int32_t* argmax, const float* g_coarse, const float* g_depth,
                                  const float* g_map, const float* g_weights, float* scratch_gE,
                                  float* scratch_gprob, float* g_logits, void* stream_)
{
    if (B <= 0 || D <= 0 || h2 <= 0 || w2 <= 0 || !logits || !candidates || !stats || !coarse || !depth || !g_logits)
        return FS_ERR_INVALID_ARG;
    if ((g_map || g_weights) && (!depth_map || !argmax || !scratch_gE)) return FS_ERR_INVALID_ARG;
    if (g_weights && !scratch_gprob) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    const long long n = (long long)B * h2 * w2;
    const bool up = g_map || g_weights;
    if (up) {
        bool ok = hipMemsetAsync(scratch_gE, 0, (size_t)n * 4, st) == hipSuccess;
        if (g_weights) ok = ok && hipMemsetAsync(scratch_gprob, 0, (size_t)n * D * 4, st) == hipSuccess;
        if (!ok) { set_last_error("depth tail memset", hipGetLastError()); return FS_ERR_LAUNCH; }
        hipLaunchKernelGGL(depth_upsample_bwd_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, st, B, D, h2,
                           w2, depth_map, argmax, log_planes, g_map, g_weights, scratch_gE, scratch_gprob);
    }
    hipLaunchKernelGGL(depth_expect_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B, D, h2 * w2,
                       logits, candidates, log_planes, stats, coarse, depth, g_coarse, g_depth,
                       up ? scratch_gE : nullptr, g_weights ? scratch_gprob : nullptr, g_logits);
    FS_CHECK_LAUNCH("depth_tail_backward");
    return FS_OK;
}

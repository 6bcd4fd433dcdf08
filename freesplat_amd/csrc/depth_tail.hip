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

// Softmax statistics (m, s) and the expectation of 64 consecutive coarse pixels per workgroup: lane = pixel, wavefront w
// takes the planes d = w (mod 4) in batches of 8 independent loads (one rescaling exponential per batch), the four partial
// (m, s, acc) triples meet in LDS.  (Rounds 2 - 3: one thread per pixel walking all D planes through a branchy online
// update -- 384 workgroups, one dependent load at a time: 63 us for 50 MB, 0.8 TB/s.)
__global__ __launch_bounds__(256) void depth_expect_kernel(int B, int D, int hw, const float* __restrict__ logits,
                                                           const float* __restrict__ cand, int log_planes,
                                                           float* __restrict__ stats, float* __restrict__ coarse,
                                                           float* __restrict__ depth)
{
    __shared__ float s_m[4][64], s_s[4][64], s_a[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    const bool live = e < (long long)B * hw;
    const long long ee = live ? e : 0;           // (a dead lane works on pixel 0 and stores nothing)
    const int b = (int)(ee / hw), p = (int)(ee % hw);
    const float* l = logits + (size_t)b * D * hw + p;
    float m = -3.0e38f, s = 0.0f, acc = 0.0f;
    auto batch = [&](const float (&v)[8], int d0) __attribute__((always_inline)) {
        float mb = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) mb = fmaxf(mb, v[i]);
        const float mn = fmaxf(m, mb);
        const float r = expf(m - mn);
        s *= r; acc *= r; m = mn;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = d0 + 4 * i;
            if (d < D) {                          // (wave-uniform)
                const float ex = expf(v[i] - m);
                s += ex;
                acc += cand[d] * ex;
            }
        }
    };
    if (D <= 128) {
        // the wavefront's 32 planes of its 64 pixels: 32 independent loads in flight, then four rescaling batches (round 6; before:
        // each batch of 8 waited for its own loads -- four exposed memory latencies per workgroup)
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = w + 32 * k + 4 * i;                  // (past D: the last plane's address, the value replaced)
                const float x = l[(size_t)min(d, D - 1) * hw];
                v[k][i] = d < D ? x : -3.0e38f;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (w + 32 * k < D) batch(v[k], w + 32 * k);
    } else {
        for (int d0 = w; d0 < D; d0 += 32) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = d0 + 4 * i;
                v[i] = d < D ? l[(size_t)d * hw] : -3.0e38f;
            }
            batch(v, d0);
        }
    }
    s_m[w][lane] = m; s_s[w][lane] = s; s_a[w][lane] = acc;
    __syncthreads();
    if (w != 0 || !live) return;
    const float M = fmaxf(fmaxf(s_m[0][lane], s_m[1][lane]), fmaxf(s_m[2][lane], s_m[3][lane]));
    float S = 0.0f, A = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = expf(s_m[k][lane] - M);   // (a wavefront without planes: exp(-huge) = 0 times its zeros)
        S += s_s[k][lane] * r;
        A += s_a[k][lane] * r;
    }
    const float E = A / S;
    stats[(size_t)b * 2 * hw + p] = M;
    stats[(size_t)b * 2 * hw + hw + p] = S;
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
// probabilities exp(l - m) / s formed ONCE per plane in LDS, 32 planes at a time, and every fine pixel then takes its
// bilinear combination from LDS: 0.4 exponentials per fine pixel and plane instead of 4, and the logits are read
// once per tile footprint instead of four times per fine pixel.  The patch is stored PIXEL-major ([pixel][plane], rows of
// 32 + 4 words): a thread fills four consecutive planes of its patch pixel with one ds_write_b128 and reads four planes of
// a tap with one ds_read_b128 -- 8 instead of 11 instructions per plane and fine pixel in the loop that bounds the kernel
// (VALU issue: 128 planes x 393 k fine pixels).  Probabilities through v_exp_f32: |error| <= 2e-8 (the argument
// (l - m) log2 e <= 0 carries a relative rounding of 2^-24, which matters only where e^t is tiny).
constexpr int kUpW = 32, kUpH = 8, kUpPlanes = 32, kUpRow = kUpPlanes + 4, kUpPatch = 18 * 6;
__global__ __launch_bounds__(256) void depth_upsample_kernel(int B, int D, int h2, int w2,
                                                             const float* __restrict__ logits,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ coarse, int log_planes,
                                                             float* __restrict__ depth_map,
                                                             float* __restrict__ depth_w, int32_t* __restrict__ argmax)
{
    __shared__ __attribute__((aligned(16))) float s_p[kUpPatch * kUpRow];
    const int H = 2 * h2, W = 2 * w2, hw = h2 * w2;
    const int t = threadIdx.x;
    // XCD-banded tile order: workgroup n runs on XCD n % 8; neighbouring tiles share cache lines of every plane (footprints
    // overlap, 18-float rows straddle 128-byte lines), so each XCD gets a band of consecutive tile rows and finds its
    // neighbours' lines in ITS L2 instead of fetching them again
    const int tiles_x = (W + kUpW - 1) / kUpW, tiles_y = (H + kUpH - 1) / kUpH;
    const int rows_total = tiles_y * B, rows_per_xcd = (rows_total + 7) / 8;
    const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3;
    const int trow = xcd * rows_per_xcd + in_xcd / tiles_x;
    if (trow >= rows_total || in_xcd / tiles_x >= rows_per_xcd) return;      // (workgroup-uniform)
    const int b = trow / tiles_y;
    const int X0 = (in_xcd % tiles_x) * kUpW, Y0 = (trow % tiles_y) * kUpH;
    const int x = X0 + (t & 31), y = Y0 + (t >> 5);
    const bool in = x < W && y < H;
    // coarse footprint of the tile (same source-coordinate arithmetic as bilin_x2)
    const float ry = h2 > 1 ? (float)(h2 - 1) / (float)(2 * h2 - 1) : 0.0f, rx = w2 > 1 ? (float)(w2 - 1) / (float)(2 * w2 - 1) : 0.0f;
    const int cy0 = min((int)((float)Y0 * ry), h2 - 1), cx0 = min((int)((float)X0 * rx), w2 - 1);
    const int cy1 = min(min((int)((float)min(Y0 + kUpH - 1, H - 1) * ry), h2 - 1) + 1, h2 - 1);
    const int cx1 = min(min((int)((float)min(X0 + kUpW - 1, W - 1) * rx), w2 - 1) + 1, w2 - 1);
    const int pw = cx1 - cx0 + 1, ph = cy1 - cy0 + 1, np = pw * ph;  // <= 18 x 6
    // patch fill: thread t forms the probabilities of patch pixel t & 127 (< np <= 108) for the plane quads of parity
    // t >> 7 -- its coarse index and softmax statistics are found ONCE (the first version recomputed k / np, c / pw, c % pw
    // for each of its 7 elements of every 16-plane chunk: ~170 integer divisions per thread, most of the kernel's 67 us)
    const int pc = t & 127, par = t >> 7;
    const bool fill = pc < np;
    const int pci = fill ? (cy0 + pc / pw) * w2 + cx0 + pc % pw : 0;
    constexpr float kLog2e = 1.44269504088896341f;
    const float pm = stats[(size_t)b * 2 * hw + pci], prs = 1.0f / stats[(size_t)b * 2 * hw + hw + pci];
    // (threads past the image edge take the taps of the nearest inside pixel of THIS tile: their patch-local
    // indices must stay inside the LDS patch; their results are never stored)
    const Bilin q = bilin_x2(min(y, H - 1), min(x, W - 1), h2, w2);
    // patch-local rows of the four taps
    const float* t00 = s_p + ((q.i00 / w2 - cy0) * pw + (q.i00 % w2 - cx0)) * kUpRow;
    const float* t01 = s_p + ((q.i01 / w2 - cy0) * pw + (q.i01 % w2 - cx0)) * kUpRow;
    const float* t10 = s_p + ((q.i10 / w2 - cy0) * pw + (q.i10 % w2 - cx0)) * kUpRow;
    const float* t11 = s_p + ((q.i11 / w2 - cy0) * pw + (q.i11 % w2 - cx0)) * kUpRow;
    const float* l = logits + (size_t)b * D * hw + pci;
    float best = -1.0f;
    int bi = 0;
    // the fill thread's 16 logits of the NEXT chunk are loaded while the workgroup blends the taps of this one (round 6; before: loads
    // issued after the barrier and waited for before the next -- one exposed memory latency per 32 planes and workgroup)
    float nv[kUpPlanes / 2];
    auto issue = [&](int d0) __attribute__((always_inline)) {
        // (guarded loads, measured: unconditional loads -- threads without a patch pixel reading pixel 0 of the image -- made one L2
        // channel the bottleneck, 31 -> 42 us at the native size; one branch around the batch with clamped planes: 35 us)
#pragma unroll
        for (int k = 0; k < kUpPlanes / 8; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = d0 + 4 * (2 * k + par) + e;
                nv[4 * k + e] = (fill && d < D) ? l[(size_t)d * hw] : -3.0e38f;     // (exp2(-huge) = 0: never a new maximum)
            }
    };
    issue(0);
    for (int d0 = 0; d0 < D; d0 += kUpPlanes) {
        __syncthreads();  // (the previous chunk's taps are read)
        if (fill) {
#pragma unroll
            for (int k = 0; k < kUpPlanes / 8; ++k) {
                const int dd = 4 * (2 * k + par);          // planes dd .. dd + 3 of the chunk
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_exp2f((nv[4 * k + e] - pm) * kLog2e) * prs;
                *(float4*)(s_p + pc * kUpRow + dd) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (d0 + kUpPlanes < D) issue(d0 + kUpPlanes);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const int nd = min(kUpPlanes, D - d0);
        for (int dd = 0; dd < nd; dd += 4) {           // (planes past nd hold zeros: never a new maximum)
            const float4 a = *(const float4*)(t00 + dd), bq = *(const float4*)(t01 + dd);
            const float4 c = *(const float4*)(t10 + dd), e = *(const float4*)(t11 + dd);
            const float v0 = q.w00 * a.x + q.w01 * bq.x + q.w10 * c.x + q.w11 * e.x;
            const float v1 = q.w00 * a.y + q.w01 * bq.y + q.w10 * c.y + q.w11 * e.y;
            const float v2 = q.w00 * a.z + q.w01 * bq.z + q.w10 * c.z + q.w11 * e.z;
            const float v3 = q.w00 * a.w + q.w01 * bq.w + q.w10 * c.w + q.w11 * e.w;
            if (v0 > best) { best = v0; bi = d0 + dd; }
            if (v1 > best) { best = v1; bi = d0 + dd + 1; }
            if (v2 > best) { best = v2; bi = d0 + dd + 2; }
            if (v3 > best) { best = v3; bi = d0 + dd + 3; }
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
// One kernel, 64 consecutive coarse pixels per workgroup (lane = pixel, wavefront w = planes d = w (mod 4), as the forward).
//   dl_d = p_d [ (cand_d - E) gE + g_prob_d - sum_j p_j g_prob_j ],
// gE = d coarse + d depth through exp / reciprocal + what the x2 map sends back through its bilinear taps; g_prob = what
// depth_weights sends back: every FINE pixel contributes to ONE plane (its arg max) at its four taps, so a coarse pixel
// collects at most 6 x 6 fine pixels (rows 2 Y - 2 .. 2 Y + 3: the align_corners source coordinate is fine * (n - 1) /
// (2 n - 1)).  Each workgroup GATHERS those into an LDS array g_prob[128 planes][64 pixels] (LDS float atomics: two fine
// pixels of a coarse pixel may share their plane; 36 of them per pixel, not per plane) -- the dense [B,D,h2,w2] gradient of
// the probabilities is never in HBM.  D > 128: the planes are walked in chunks of 128, gathering per chunk (twice: once
// for the dot product, once for the output).
// (Rounds 2 - 3: memset of a dense g_prob, a scatter kernel with 8 global atomics per fine pixel, then one thread per
// coarse pixel reading logits and g_prob twice: 12 + 89 + 103 us for 2 x 128 x 192 x 256.)
constexpr int kBwdPlanes = 128;
template <bool ONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void depth_tail_bwd_kernel(int B, int D, int h2, int w2, const float* __restrict__ logits,
                                                             const float* __restrict__ cand, int log_planes,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ coarse,
                                                             const float* __restrict__ depth,
                                                             const float* __restrict__ depth_map,
                                                             const int32_t* __restrict__ argmax,
                                                             const float* __restrict__ g_coarse,
                                                             const float* __restrict__ g_depth,
                                                             const float* __restrict__ g_map,
                                                             const float* __restrict__ g_w, float* __restrict__ g_logits)
{
    __shared__ __attribute__((aligned(16))) float s_gp[kBwdPlanes * 64];
    __shared__ float s_gE[64], s_dot[4][64];
    const int hw = h2 * w2, H = 2 * h2, W = 2 * w2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + lane;
    const bool live = e < (long long)B * hw;
    const long long ee = live ? e : 0;
    const int b = (int)(ee / hw), p = (int)(ee % hw);
    const int Y = p / w2, X = p % w2;
    if (threadIdx.x < 64) s_gE[threadIdx.x] = 0.0f;

    // the fine pixels that tap this coarse pixel: wavefront w looks at 9 of the 36 candidates
    auto gather = [&](int c0, bool with_map) {
        for (int k = threadIdx.x; k < kBwdPlanes * 16; k += 256) ((float4*)s_gp)[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();
        if (live) {
            // (round 6: the weights first, then every load of the nine candidates in flight together -- the loop used to take them one
            // dependent load at a time behind its `continue`s)
            float wt[9], gw[9], gm[9], dm[9];
            int am[9];
            int f[9];                              // (inside image b: H W < 2^31)
            const size_t fb = (size_t)b * H * W;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int idx = 9 * w + i;
                const int fy = 2 * Y + idx / 6 - 2, fx = 2 * X + idx % 6 - 2;
                const bool inside = fy >= 0 && fy < H && fx >= 0 && fx < W;
                const int cy = min(max(fy, 0), H - 1), cx = min(max(fx, 0), W - 1);
                const Bilin q = bilin_x2(cy, cx, h2, w2);
                wt[i] = inside ? (q.i00 == p ? q.w00 : 0.0f) + (q.i01 == p ? q.w01 : 0.0f) + (q.i10 == p ? q.w10 : 0.0f) +
                                     (q.i11 == p ? q.w11 : 0.0f)
                               : 0.0f;
                f[i] = cy * W + cx;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                am[i] = g_w ? argmax[fb + f[i]] : 0;
                gw[i] = g_w ? g_w[fb + f[i]] : 0.0f;
                dm[i] = (with_map && g_map) ? depth_map[fb + f[i]] : 0.0f;
                gm[i] = (with_map && g_map) ? g_map[fb + f[i]] : 0.0f;
            }
            float gEacc = 0.0f;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (wt[i] == 0.0f) continue;
                if (with_map && g_map) gEacc += wt[i] * (log_planes ? gm[i] * dm[i] : -gm[i] * dm[i] * dm[i]);   // d exp(f) = exp(f); d(1/f) = -1/f^2
                if (g_w) {
                    const int d = am[i] - c0;
                    if (d >= 0 && d < kBwdPlanes) atomicAdd(&s_gp[d * 64 + lane], wt[i] * gw[i]);
                }
            }
            if (with_map && g_map) atomicAdd(&s_gE[lane], gEacc);
        }
        __syncthreads();
    };

    const bool up = g_map || g_w;
    const float m = stats[(size_t)b * 2 * hw + p], rs = 1.0f / stats[(size_t)b * 2 * hw + hw + p];
    const float* l = logits + (size_t)b * D * hw + p;
    const int nchunks = (D + kBwdPlanes - 1) / kBwdPlanes;
    float dot = 0.0f;
    if constexpr (ONE) {
        // D <= 128: this wavefront's 32 planes of the pixel are loaded ONCE, all 32 loads in flight while the workgroup clears
        // and gathers g_prob, and the probabilities stay in registers between the dot product and the output (round 6; before:
        // two passes over the logits with an expf each, the first load issued only after the gather's barrier)
        float pd[kBwdPlanes / 4];
#pragma unroll
        for (int j = 0; j < kBwdPlanes / 4; ++j) {
            const int d = w + 4 * j;
            const float x = l[(size_t)min(d, D - 1) * hw];         // (unconditional: planes past D re-read the last one)
            pd[j] = d < D ? x : -3.0e38f;                          // (exp(-huge) = 0)
        }
        if (up) gather(0, true); else __syncthreads();             // (else: s_gE's zeros)
#pragma unroll
        for (int j = 0; j < kBwdPlanes / 4; ++j) pd[j] = expf(pd[j] - m) * rs;
        if (g_w) {
#pragma unroll
            for (int j = 0; j < kBwdPlanes / 4; ++j) dot += pd[j] * s_gp[(w + 4 * j) * 64 + lane];
            s_dot[w][lane] = dot;
            __syncthreads();
            dot = (s_dot[0][lane] + s_dot[1][lane]) + (s_dot[2][lane] + s_dot[3][lane]);
        }
        const float E = coarse[ee];
        float gE = s_gE[lane];
        if (g_coarse) gE += g_coarse[ee];
        if (g_depth) gE += log_planes ? g_depth[ee] * depth[ee] : -g_depth[ee] * depth[ee] * depth[ee];
        float* go = g_logits + (size_t)b * D * hw + p;
        if (live) {
#pragma unroll
            for (int j = 0; j < kBwdPlanes / 4; ++j) {
                const int d = w + 4 * j;
                if (d < D) {
                    const float gpd = g_w ? s_gp[d * 64 + lane] : 0.0f;
                    go[(size_t)d * hw] = pd[j] * ((cand[d] - E) * gE + gpd - dot);
                }
            }
        }
        return;
    }
    if (up) {
        for (int c = 0; c < nchunks; ++c) {
            gather(c * kBwdPlanes, c == 0);
            if (g_w) {
                const int dend = min(D, (c + 1) * kBwdPlanes);
#pragma unroll 8
                for (int d = c * kBwdPlanes + w; d < dend; d += 4)
                    dot += expf(l[(size_t)d * hw] - m) * rs * s_gp[(d - c * kBwdPlanes) * 64 + lane];
                if (c + 1 < nchunks) __syncthreads();              // (the next gather clears s_gp: every wavefront must have read it)
            }
        }
        if (g_w) {
            s_dot[w][lane] = dot;
            __syncthreads();
            dot = (s_dot[0][lane] + s_dot[1][lane]) + (s_dot[2][lane] + s_dot[3][lane]);
        }
    } else {
        __syncthreads();     // (s_gE's zeros)
    }
    const float E = coarse[ee];
    float gE = s_gE[lane];
    if (g_coarse) gE += g_coarse[ee];
    if (g_depth) gE += log_planes ? g_depth[ee] * depth[ee] : -g_depth[ee] * depth[ee] * depth[ee];
    float* go = g_logits + (size_t)b * D * hw + p;
    for (int c = 0; c < nchunks; ++c) {
        if (up && nchunks > 1) gather(c * kBwdPlanes, false);      // (one chunk: the gather above is still in LDS)
        const int dend = min(D, (c + 1) * kBwdPlanes);
        if (live) {
#pragma unroll 8
            for (int d = c * kBwdPlanes + w; d < dend; d += 4) {
                const float pd = expf(l[(size_t)d * hw] - m) * rs;
                const float gpd = g_w ? s_gp[(d - c * kBwdPlanes) * 64 + lane] : 0.0f;
                go[(size_t)d * hw] = pd * ((cand[d] - E) * gE + gpd - dot);
            }
        }
        if (up && nchunks > 1) __syncthreads();                    // (the next gather clears s_gp)
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
    ScopedStage prof_(kStEncoderTail, st);
    hipLaunchKernelGGL(depth_expect_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, B, D, h2 * w2, logits,
                       candidates, log_planes, stats, coarse, depth);
    if (depth_map)
    {
        const int tiles_x = (2 * w2 + kUpW - 1) / kUpW, tiles_y = (2 * h2 + kUpH - 1) / kUpH;
        const int rows_per_xcd = (tiles_y * B + 7) / 8;
        hipLaunchKernelGGL(depth_upsample_kernel, dim3((unsigned)(8 * rows_per_xcd * tiles_x)), dim3(256), 0, st, B, D, h2, w2,
                           logits, stats, coarse, log_planes, depth_map, depth_weights, argmax);
    }
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
    if ((g_map || g_weights) && (!depth_map || !argmax)) return FS_ERR_INVALID_ARG;
    (void)scratch_gE; (void)scratch_gprob;      // (ABI 1 - 3 scratch of the scatter form: unused since the gather form, may be NULL)
    hipStream_t st = (hipStream_t)stream_;
    const long long n = (long long)B * h2 * w2;
    ScopedStage prof_(kStEncoderTail, st);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, B, D, h2, w2, logits, candidates, log_planes,
                           stats, coarse, depth, depth_map, argmax, g_coarse, g_depth, g_map, g_weights, g_logits);
    };
    if (D <= kBwdPlanes) go(depth_tail_bwd_kernel<true>); else go(depth_tail_bwd_kernel<false>);
    FS_CHECK_LAUNCH("depth_tail_backward");
    return FS_OK;
}

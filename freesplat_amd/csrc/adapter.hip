// adapter.hip -- the two GaussianAdapter steps either side of PTF (8(f) N2).
//
//  a11  Create_from_depth_map.project + GaussianAdapter.forward(fusion=True)
//       (src/model/encoder/common/gaussian_adapter.py:36-79, 174-188): per-pixel depth -> world xyz.
//       The reference runs Python loops over batch and views with ~10 small kernels each.
//  a14  GaussianAdapter.forward(fusion=False, coords=...) (gaussian_adapter.py:151-172, 191-201) with
//       quaternion_to_matrix / build_covariance (common/gaussians.py:8-44): 34 raw channels + depth +
//       blended extrinsics -> world covariance, masked SH, scales, unit quaternion (xyzw).
// Both are streaming per-element kernels (HBM-bound: a14 moves 34+1+16 floats in and 43 out per
// Gaussian); forward AND backward are provided so the ops stay differentiable.
#include "fs_common.h"

namespace fs {

// ------------------------------------------------------------------ a11: unprojection
__global__ __launch_bounds__(256) void unproject_fwd_kernel(int V, int h, int w, const float* __restrict__ depths,
                                                            const float* __restrict__ E,
                                                            const float* __restrict__ k0, float* __restrict__ xyz)
{
    const size_t P = (size_t)h * w, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)V * P) return;
    const int v = (int)(e / P), pix = (int)(e % P);
    const float u = (float)(pix % w), r = (float)(pix / w);          // integer pixel coordinates (no +0.5)
    const float px = (u - k0[2]) / k0[0], py = (r - k0[3]) / k0[1];   // gaussian_adapter.py:45
    const float z = depths[e];
    const float x = px * z, y = py * z;                               // :62
    const float* m = E + (size_t)v * 16;
    xyz[3 * e + 0] = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];      // :68-70 (c2w @ [x y z 1])
    xyz[3 * e + 1] = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
    xyz[3 * e + 2] = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
}

__global__ __launch_bounds__(256) void unproject_bwd_kernel(int V, int h, int w, const float* __restrict__ E,
                                                            const float* __restrict__ k0,
                                                            const float* __restrict__ g_xyz,
                                                            float* __restrict__ g_depths)
{
    const size_t P = (size_t)h * w, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)V * P) return;
    const int v = (int)(e / P), pix = (int)(e % P);
    const float px = ((float)(pix % w) - k0[2]) / k0[0], py = ((float)(pix / w) - k0[3]) / k0[1];
    const float* m = E + (size_t)v * 16;
    const float gx = g_xyz[3 * e], gy = g_xyz[3 * e + 1], gz = g_xyz[3 * e + 2];
    g_depths[e] = gx * (m[0] * px + m[1] * py + m[2]) + gy * (m[4] * px + m[5] * py + m[6]) +
                  gz * (m[8] * px + m[9] * py + m[10]);
}

// ------------------------------------------------------------------ a14: latent channels -> Gaussian
struct HeadFwd {
    float sig[3], sc[3];  // sigmoid(raw scale), final scale
    float nq;             // |raw quaternion|
    float q[4];           // unit quaternion xyzw
    float two_s;
    float R[9], Mm[9], A[9];
};

__device__ __forceinline__ void head_forward(const float* raw, float depth, float mult, float smin, float smax,
                                             HeadFwd& f)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        f.sig[j] = 1.0f / (1.0f + expf(-raw[j]));
        f.sc[j] = (smin + (smax - smin) * f.sig[j]) * depth * mult;       // gaussian_adapter.py:155-160
    }
    f.nq = sqrtf(raw[3] * raw[3] + raw[4] * raw[4] + raw[5] * raw[5] + raw[6] * raw[6]);
    const float D = f.nq + 1e-8f;
#pragma unroll
    for (int a = 0; a < 4; ++a) f.q[a] = raw[3 + a] / D;                  // :163
    const float i = f.q[0], j = f.q[1], k = f.q[2], r = f.q[3];           // gaussians.py:15 (xyzw)
    f.two_s = 2.0f / (i * i + j * j + k * k + r * r + 1e-8f);
    const float t = f.two_s;
    f.R[0] = 1.0f - t * (j * j + k * k); f.R[1] = t * (i * j - k * r); f.R[2] = t * (i * k + j * r);
    f.R[3] = t * (i * j + k * r); f.R[4] = 1.0f - t * (i * i + k * k); f.R[5] = t * (j * k - i * r);
    f.R[6] = t * (i * k - j * r); f.R[7] = t * (j * k + i * r); f.R[8] = 1.0f - t * (i * i + j * j);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) f.Mm[3 * a + b] = f.R[3 * a + b] * f.sc[b];   // R diag(s)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            f.A[3 * a + b] = f.Mm[3 * a] * f.Mm[3 * b] + f.Mm[3 * a + 1] * f.Mm[3 * b + 1] +
                             f.Mm[3 * a + 2] * f.Mm[3 * b + 2];                    // R S S^T R^T
}

__device__ __forceinline__ void mat3_mul(const float* X, const float* Y, float* Z)  // Z = X Y
{
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Z[3 * a + b] = X[3 * a] * Y[b] + X[3 * a + 1] * Y[3 + b] + X[3 * a + 2] * Y[6 + b];
}
__device__ __forceinline__ void mat3_mul_bt(const float* X, const float* Y, float* Z)  // Z = X Y^T
{
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            Z[3 * a + b] = X[3 * a] * Y[3 * b] + X[3 * a + 1] * Y[3 * b + 1] + X[3 * a + 2] * Y[3 * b + 2];
}
__device__ __forceinline__ void mat3_mul_at(const float* X, const float* Y, float* Z)  // Z = X^T Y
{
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Z[3 * a + b] = X[a] * Y[b] + X[3 + a] * Y[3 + b] + X[6 + a] * Y[6 + b];
}

// The [M, PER] arrays of this step are arrays of structures: thread m touching raw[m * 34 + c] makes every load
// instruction of a wavefront span 64 x 136 bytes (2.5 - 3.5 TB/s of the algorithmic bytes in rounds 2 - 3).  A workgroup's 256
// consecutive rows are ONE contiguous slab, so they cross HBM as 16-byte coalesced accesses and meet their threads in LDS
// (row strides 34 / 27 / 9 words: at most 2-way bank conflicts).
// One 4-float row of a per-Gaussian 4x4: a 16-byte access when the array is 16-byte aligned (the rows are 16-byte multiples
// apart, so the base decides), four scalar ones otherwise -- in the inference fold the extrinsics are a view at float offset
// rows * 69 inside one buffer, 4-byte aligned when h * w is not a multiple of 4.
__device__ __forceinline__ float4 load_row4(const float* __restrict__ p)
{
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) return *(const float4*)p;
    return make_float4(p[0], p[1], p[2], p[3]);
}
__device__ __forceinline__ void store_row4(float* __restrict__ p, float4 v)
{
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) { *(float4*)p = v; return; }
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}

template <int PER>
__device__ __forceinline__ void rows_in(float* __restrict__ lds, const float* __restrict__ src, long long m0, int cnt)
{
    const float* g = src + m0 * PER;
    const int total = cnt * PER;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        for (int k = 4 * threadIdx.x; k + 3 < total; k += 1024) *(float4*)(lds + k) = *(const float4*)(g + k);
        for (int k = (total & ~3) + threadIdx.x; k < total; k += 256) lds[k] = g[k];
    } else {
        for (int k = threadIdx.x; k < total; k += 256) lds[k] = g[k];
    }
}
template <int PER>
__device__ __forceinline__ void rows_out(float* __restrict__ dst, const float* __restrict__ lds, long long m0, int cnt)
{
    float* g = dst + m0 * PER;
    const int total = cnt * PER;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        for (int k = 4 * threadIdx.x; k + 3 < total; k += 1024) *(float4*)(g + k) = *(const float4*)(lds + k);
        for (int k = (total & ~3) + threadIdx.x; k < total; k += 256) g[k] = lds[k];
    } else {
        for (int k = threadIdx.x; k < total; k += 256) g[k] = lds[k];
    }
}

__global__ __launch_bounds__(256) void head_fwd_kernel(long long M, const float* __restrict__ raw,
                                                       const float* __restrict__ depths,
                                                       const float* __restrict__ E,
                                                       const float* __restrict__ mult, long long mult_stride,
                                                       const float* __restrict__ sh_mask, float smin, float smax,
                                                       float* __restrict__ cov, float* __restrict__ sh,
                                                       float* __restrict__ scales, float* __restrict__ rot)
{
    __shared__ float s_rows[256 * 43];          // in: 34 raw floats per row; out: sh 27 | cov 9 | scales 3 | rotation 4
    const long long m0 = (long long)blockIdx.x * 256;
    const int cnt = (int)min((long long)256, M - m0), t = threadIdx.x;
    const long long m = m0 + t;
    const bool live = t < cnt;
    rows_in<34>(s_rows, raw, m0, cnt);
    __syncthreads();
    float rw[34];
#pragma unroll
    for (int c = 0; c < 34; ++c) rw[c] = live ? s_rows[t * 34 + c] : 1.0f;
    __syncthreads();                             // (the outputs below reuse the rows' LDS)
    if (live) {
        HeadFwd f;
        head_forward(rw, depths[m], mult[m * mult_stride], smin, smax, f);
        float Rc[9], B[9], S[9];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float4 er = load_row4(E + m * 16 + 4 * a);
            Rc[3 * a] = er.x; Rc[3 * a + 1] = er.y; Rc[3 * a + 2] = er.z;
        }
        mat3_mul(Rc, f.A, B);        // c2w @ cov
        mat3_mul_bt(B, Rc, S);       // ... @ c2w^T                 gaussian_adapter.py:171-172
        float* o_sh = s_rows + t * 27;
        float* o_cov = s_rows + 256 * 27 + t * 9;
        float* o_sc = s_rows + 256 * 36 + t * 3;
        float* o_rot = s_rows + 256 * 39 + t * 4;
#pragma unroll
        for (int c = 0; c < 27; ++c) o_sh[c] = rw[7 + c] * sh_mask[c % 9];   // "(xyz d_sh)" * mask :166-167
#pragma unroll
        for (int c = 0; c < 9; ++c) o_cov[c] = S[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) o_sc[c] = f.sc[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) o_rot[c] = f.q[c];
    }
    __syncthreads();
    rows_out<27>(sh, s_rows, m0, cnt);
    rows_out<9>(cov, s_rows + 256 * 27, m0, cnt);
    rows_out<3>(scales, s_rows + 256 * 36, m0, cnt);
    rows_out<4>(rot, s_rows + 256 * 39, m0, cnt);
}

__global__ __launch_bounds__(256) void head_bwd_kernel(long long M, const float* __restrict__ raw,
                                                       const float* __restrict__ depths,
                                                       const float* __restrict__ E,
                                                       const float* __restrict__ mult, long long mult_stride,
                                                       const float* __restrict__ sh_mask, float smin, float smax,
                                                       const float* __restrict__ g_cov,
                                                       const float* __restrict__ g_sh,
                                                       const float* __restrict__ g_scales,
                                                       const float* __restrict__ g_rot, float* __restrict__ g_raw,
                                                       float* __restrict__ g_depths, float* __restrict__ g_E)
{
    __shared__ float s_rows[256 * 34];          // raw rows in, then g_sh rows in, then g_raw rows out
    const long long m0 = (long long)blockIdx.x * 256;
    const int cnt = (int)min((long long)256, M - m0), t = threadIdx.x;
    const bool live = t < cnt;
    const long long m = live ? m0 + t : m0;     // (a dead thread works on the workgroup's first row and stores nothing)
    rows_in<34>(s_rows, raw, m0, cnt);
    __syncthreads();
    float rw[34];
#pragma unroll
    for (int c = 0; c < 34; ++c) rw[c] = s_rows[(live ? t : 0) * 34 + c];
    __syncthreads();
    float gsh[27];
    if (g_sh) {
        rows_in<27>(s_rows, g_sh, m0, cnt);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 27; ++c) gsh[c] = s_rows[(live ? t : 0) * 27 + c];
        __syncthreads();
    } else {
#pragma unroll
        for (int c = 0; c < 27; ++c) gsh[c] = 0.0f;
    }
    const float depth = depths[m], mu = mult[m * mult_stride];
    HeadFwd f;
    head_forward(rw, depth, mu, smin, smax, f);
    float Rc[9], G[9], B[9];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 er = load_row4(E + m * 16 + 4 * a);
        Rc[3 * a] = er.x; Rc[3 * a + 1] = er.y; Rc[3 * a + 2] = er.z;
#pragma unroll
        for (int b = 0; b < 3; ++b) G[3 * a + b] = g_cov ? g_cov[m * 9 + 3 * a + b] : 0.0f;
    }
    mat3_mul(Rc, f.A, B);
    // Sigma = B Rc^T, B = Rc A
    float dB[9], dRc[9], dA[9], t1[9];
    mat3_mul(G, Rc, dB);                       // dB = G Rc
    mat3_mul_at(G, B, dRc);                    // dRc = G^T B
    mat3_mul_bt(dB, f.A, t1);                  // + dB A^T
#pragma unroll
    for (int c = 0; c < 9; ++c) dRc[c] += t1[c];
    mat3_mul_at(Rc, dB, dA);                   // dA = Rc^T dB
    // A = Mm Mm^T  ->  dMm = (dA + dA^T) Mm
    float sym[9], dMm[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) sym[3 * a + b] = dA[3 * a + b] + dA[3 * b + a];
    mat3_mul(sym, f.Mm, dMm);
    float dR[9], dsc[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            dR[3 * a + b] = dMm[3 * a + b] * f.sc[b];
            dsc[b] += dMm[3 * a + b] * f.R[3 * a + b];
        }
    if (g_scales)
#pragma unroll
        for (int b = 0; b < 3; ++b) dsc[b] += g_scales[m * 3 + b];
    // R = I + two_s * T(q)
    const float i = f.q[0], j = f.q[1], k = f.q[2], r = f.q[3], ts = f.two_s;
    const float T[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k),
                        j * k - i * r,    i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float dts = 0.0f, dT[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) { dts += dR[c] * T[c]; dT[c] = dR[c] * ts; }
    float dq[4];
    dq[0] = dT[1] * j + dT[2] * k + dT[3] * j - 2.0f * i * dT[4] - dT[5] * r + dT[6] * k + dT[7] * r - 2.0f * i * dT[8];
    dq[1] = -2.0f * j * dT[0] + dT[1] * i + dT[2] * r + dT[3] * i + dT[5] * k - dT[6] * r + dT[7] * k - 2.0f * j * dT[8];
    dq[2] = -2.0f * k * dT[0] - dT[1] * r + dT[2] * i + dT[3] * r - 2.0f * k * dT[4] + dT[5] * j + dT[6] * i + dT[7] * j;
    dq[3] = -dT[1] * k + dT[2] * j + dT[3] * k - dT[5] * i - dT[6] * j + dT[7] * i;
    const float dn2 = -0.5f * ts * ts * dts;   // two_s = 2 / (n2 + eps)
#pragma unroll
    for (int a = 0; a < 4; ++a) dq[a] += 2.0f * f.q[a] * dn2;
    if (g_rot)
#pragma unroll
        for (int a = 0; a < 4; ++a) dq[a] += g_rot[m * 4 + a];
    // q = raw_q / (|raw_q| + eps)
    const float D = f.nq + 1e-8f;
    float dot = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) dot += dq[a] * rw[3 + a];
    const float cq = f.nq > 0.0f ? dot / (D * D * f.nq) : 0.0f;
    float gr[34];
#pragma unroll
    for (int a = 0; a < 4; ++a) gr[3 + a] = dq[a] / D - cq * rw[3 + a];
    float gdepth = 0.0f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        gr[b] = dsc[b] * (smax - smin) * depth * mu * f.sig[b] * (1.0f - f.sig[b]);
        gdepth += dsc[b] * (smin + (smax - smin) * f.sig[b]) * mu;
    }
#pragma unroll
    for (int c = 0; c < 27; ++c) gr[7 + c] = gsh[c] * sh_mask[c % 9];
    if (live) {
#pragma unroll
        for (int c = 0; c < 34; ++c) s_rows[t * 34 + c] = gr[c];
        g_depths[m] = gdepth;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            store_row4(g_E + m * 16 + 4 * a, a < 3 ? make_float4(dRc[3 * a], dRc[3 * a + 1], dRc[3 * a + 2], 0.0f)
                                                   : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    }
    __syncthreads();
    rows_out<34>(g_raw, s_rows, m0, cnt);
}


// ------------------------------------------------------------------ per-pixel latents: head + skip, channel-major -> pixel-major
// encoder_freesplat.py:311-316: latents = (head[:, 1:] + skip) rearranged "(b v) c h w -> b v (h w) c", densities from head[:, :1].
// In torch that is an add, then -- because the fold wants [V, P, 64] rows -- a transposing copy of 963 MB at config 3 (1.4 ms), and
// in the backward two more (4.5 ms each: the gradient back to channel-major for the head slice and for the skip convolution):
// 10.3 ms of a 70 ms training step (profiles/r5_c3_step_glue.json).  Here: one pass each way through a 64 x 64 LDS tile -- rows of
// 64 pixels (256 B) in, rows of 64 channels (256 B) out, the same single fp32 add.
constexpr int kLpC = 64, kLpT = 64;   // channels, pixels per tile
__global__ __launch_bounds__(256) void latents_pack_fwd_kernel(long long P, const float* __restrict__ head,
                                                               const float* __restrict__ skip, float* __restrict__ lat,
                                                               float* __restrict__ dens)
{
    __shared__ float tile[kLpC][kLpT + 1];
    const int v = blockIdx.y, t = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * kLpT;
    const int np = (int)min((long long)kLpT, P - p0);
    const float* hv = head + (size_t)v * (kLpC + 1) * P;
    const float* sv = skip + (size_t)v * kLpC * P;
    const int px = t & 63, c0 = t >> 6;
    if (px < np) {
#pragma unroll 4
        for (int c = c0; c < kLpC; c += 4) tile[c][px] = hv[(size_t)(1 + c) * P + p0 + px] + sv[(size_t)c * P + p0 + px];
        if (c0 == 0 && dens) dens[(size_t)v * P + p0 + px] = hv[p0 + px];
    }
    __syncthreads();
    float* o = lat + ((size_t)v * P + p0) * kLpC;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = k * 256 + t, q = e >> 4, c4 = (e & 15) * 4;      // pixel q of the tile, channels c4 .. c4 + 3
        if (q < np) *(float4*)(o + (size_t)q * kLpC + c4) = make_float4(tile[c4][q], tile[c4 + 1][q], tile[c4 + 2][q], tile[c4 + 3][q]);
    }
}

// g_head [V, C + 1, P] (channel 0 <- g_dens or 0, channel 1 + c <- g_lat[.., c]) and g_skip [V, C, P] (<- g_lat[.., c]); either
// output may be NULL (its input does not require a gradient), g_lat == NULL means a zero gradient.
__global__ __launch_bounds__(256) void latents_pack_bwd_kernel(long long P, const float* __restrict__ g_lat,
                                                               const float* __restrict__ g_dens, float* __restrict__ g_head,
                                                               float* __restrict__ g_skip)
{
    __shared__ float tile[kLpC][kLpT + 1];
    const int v = blockIdx.y, t = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * kLpT;
    const int np = (int)min((long long)kLpT, P - p0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = k * 256 + t, q = e >> 4, c4 = (e & 15) * 4;
        float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (g_lat && q < np) g = *(const float4*)(g_lat + ((size_t)v * P + p0 + q) * kLpC + c4);
        tile[c4][q] = g.x; tile[c4 + 1][q] = g.y; tile[c4 + 2][q] = g.z; tile[c4 + 3][q] = g.w;
    }
    __syncthreads();
    const int px = t & 63, c0 = t >> 6;
    if (px >= np) return;
    float* hv = g_head ? g_head + (size_t)v * (kLpC + 1) * P + p0 + px : nullptr;
    float* sv = g_skip ? g_skip + (size_t)v * kLpC * P + p0 + px : nullptr;
#pragma unroll 4
    for (int c = c0; c < kLpC; c += 4) {
        const float g = tile[c][px];
        if (hv) hv[(size_t)(1 + c) * P] = g;
        if (sv) sv[(size_t)c * P] = g;
    }
    if (c0 == 0 && hv) hv[0] = g_dens ? g_dens[(size_t)v * P + p0 + px] : 0.0f;
}

}  // namespace fs

using namespace fs;

FS_API int fs_unproject_forward(int32_t V, int32_t h, int32_t w, const float* depths, const float* extrinsics,
                                const float* k0_pix, float* xyz, void* stream_)
{
    if (V < 0 || h <= 0 || w <= 0 || !extrinsics || !k0_pix) return FS_ERR_INVALID_ARG;
    if (V == 0) return FS_OK;
    if (!depths || !xyz) return FS_ERR_INVALID_ARG;
    const size_t n = (size_t)V * h * w;
    ScopedStage prof_(kStEncoderTail, (hipStream_t)stream_);
    hipLaunchKernelGGL(unproject_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, V, h,
                       w, depths, extrinsics, k0_pix, xyz);
    FS_CHECK_LAUNCH("unproject_forward");
    return FS_OK;
}

FS_API int fs_unproject_backward(int32_t V, int32_t h, int32_t w, const float* extrinsics, const float* k0_pix,
                                 const float* g_xyz, float* g_depths, void* stream_)
{
    if (V < 0 || h <= 0 || w <= 0 || !extrinsics || !k0_pix) return FS_ERR_INVALID_ARG;
    if (V == 0) return FS_OK;
    if (!g_xyz || !g_depths) return FS_ERR_INVALID_ARG;
    const size_t n = (size_t)V * h * w;
    ScopedStage prof_(kStEncoderTail, (hipStream_t)stream_);
    hipLaunchKernelGGL(unproject_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, V, h,
                       w, extrinsics, k0_pix, g_xyz, g_depths);
    FS_CHECK_LAUNCH("unproject_backward");
    return FS_OK;
}

FS_API int fs_gaussian_head_forward(int64_t M, const float* raw, const float* depths, const float* extrinsics,
                                    const float* multiplier, int64_t mult_stride, const float* sh_mask,
                                    float scale_min, float scale_max, float* cov, float* harmonics, float* scales,
                                    float* rotations, void* stream_)
{
    if (M < 0 || !multiplier || !sh_mask) return FS_ERR_INVALID_ARG;
    if (M == 0) return FS_OK;
    if (!raw || !depths || !extrinsics || !cov || !harmonics || !scales || !rotations) return FS_ERR_INVALID_ARG;
    ScopedStage prof_(kStEncoderTail, (hipStream_t)stream_);
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       (long long)M, raw, depths, extrinsics, multiplier, (long long)mult_stride, sh_mask, scale_min,
                       scale_max, cov, harmonics, scales, rotations);
    FS_CHECK_LAUNCH("gaussian_head_forward");
    return FS_OK;
}

FS_API int fs_gaussian_head_backward(int64_t M, const float* raw, const float* depths, const float* extrinsics,
                                     const float* multiplier, int64_t mult_stride, const float* sh_mask,
                                     float scale_min, float scale_max, const float* g_cov, const float* g_harmonics,
                                     const float* g_scales, const float* g_rotations, float* g_raw, float* g_depths,
                                     float* g_extrinsics, void* stream_)
{
    if (M < 0 || !multiplier || !sh_mask) return FS_ERR_INVALID_ARG;
    if (M == 0) return FS_OK;
    if (!raw || !depths || !extrinsics || !g_raw || !g_depths || !g_extrinsics) return FS_ERR_INVALID_ARG;
    ScopedStage prof_(kStEncoderTail, (hipStream_t)stream_);
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       (long long)M, raw, depths, extrinsics, multiplier, (long long)mult_stride, sh_mask, scale_min,
                       scale_max, g_cov, g_harmonics, g_scales, g_rotations, g_raw, g_depths, g_extrinsics);
    FS_CHECK_LAUNCH("gaussian_head_backward");
    return FS_OK;
}

FS_API int fs_latents_pack_forward(int32_t V, int64_t P, int32_t C, const float* head, const float* skip, float* latents,
                                   float* dens, void* stream_)
{
    if (V < 0 || P < 0 || C != kLpC) return FS_ERR_INVALID_ARG;
    if (V == 0 || P == 0) return FS_OK;
    if (!head || !skip || !latents || V > 65535) return FS_ERR_INVALID_ARG;
    ScopedStage prof_(kStEncoderTail, (hipStream_t)stream_);
    hipLaunchKernelGGL(latents_pack_fwd_kernel, dim3((unsigned)((P + kLpT - 1) / kLpT), (unsigned)V), dim3(256), 0, (hipStream_t)stream_,
                       (long long)P, head, skip, latents, dens);
    FS_CHECK_LAUNCH("latents_pack_forward");
    return FS_OK;
}

FS_API int fs_latents_pack_backward(int32_t V, int64_t P, int32_t C, const float* g_latents, const float* g_dens, float* g_head,
                                    float* g_skip, void* stream_)
{
    if (V < 0 || P < 0 || C != kLpC) return FS_ERR_INVALID_ARG;
    if (V == 0 || P == 0 || (!g_head && !g_skip)) return FS_OK;
    if (V > 65535) return FS_ERR_INVALID_ARG;
    ScopedStage prof_(kStEncoderTail, (hipStream_t)stream_);
    hipLaunchKernelGGL(latents_pack_bwd_kernel, dim3((unsigned)((P + kLpT - 1) / kLpT), (unsigned)V), dim3(256), 0, (hipStream_t)stream_,
                       (long long)P, g_latents, g_dens, g_head, g_skip);
    FS_CHECK_LAUNCH("latents_pack_backward");
    return FS_OK;
}

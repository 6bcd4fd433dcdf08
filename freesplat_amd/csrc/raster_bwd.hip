// raster_bwd.hip -- backward of the tile rasterizer for MI355X (gfx950).
//
// Replaces the autograd backward of the CUDA extension FreeSplat uses
// (src/model/decoder/cuda_splatting.py:114-127; semantics: SURVEY.md Appendix A.5).
//
//   render_bwd      1 workgroup / tile, back-to-front over the same sorted list.  Per-pixel
//                   partials are summed across the 64 lanes of each wavefront in registers
//                   (cross-lane butterflies), then across the 4 wavefronts with LDS float
//                   atomics into a per-batch accumulator, and only then flushed to HBM:
//                   one global atomic per (tile, Gaussian, component) instead of one per
//                   (pixel, Gaussian, component).
//   preprocess_bwd  1 thread / Gaussian: conic -> cov2D -> Sigma & view position, mean2D ->
//                   mean through the perspective divide, RGB -> SH & view direction.
//                   SH gradients leave through LDS so the [N, M, 3] rows are written coalesced.
#include "fs_common.h"

namespace fs {

constexpr int kGradStride = 12;  // floats per Gaussian in the accumulation scratch
// layout: 0,1 mean2D | 2,3,4 conic (x, y(half), z) | 5 opacity | 6,7,8 rgb | 9 view z | 10,11 pad

// ---- cross-lane exchange with lane ^ M without touching LDS memory where the ISA allows ----
template <int M>
__device__ __forceinline__ float xchg(float v)
{
    const int i = __float_as_int(v);
    if constexpr (M == 1) return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    else if constexpr (M == 32) return __shfl_xor(v, 32, 64);
    else return __int_as_float(__builtin_amdgcn_ds_swizzle(i, (M << 10) | 0x1F));                          // bit-mode xor M
}

// Sum 10 per-lane values over the 64 lanes of a wavefront with a HALVING butterfly: at every level a
// lane keeps one half of its values and hands the other half to its partner, so 10 values cost
// 5+3+2+1+1+1 = 13 exchanges instead of 10 x 6.  Afterwards the 16 lanes with (lane & 3) == 0 each
// hold the complete sum of ONE value (or of a zero pad); wave_value_index() says which.
__device__ __forceinline__ int wave_value_index(int lane)
{
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
    // level 1 keeps [0..4] | [5..9]; level 2 keeps slots [0,1,2] | [3,4,-]; level 3 [0,1] | [2,-]; level 4 [0] | [1]
    const int s3 = b2;                      // slot within the level-3 survivors (2 slots)
    if (b3 && s3 == 1) return -1;           // pad
    const int s2 = b3 ? 2 : s3;             // slot within the level-2 survivors (3 slots)
    if (b4 && s2 == 2) return -1;           // pad
    const int s1 = b4 ? 3 + s2 : s2;        // slot within the level-1 survivors (5 slots)
    return b5 ? 5 + s1 : s1;
}
__device__ __forceinline__ float wave_sum10(const float (&v)[10], int lane)
{
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] = (b5 ? v[5 + k] : v[k]) + xchg<32>(b5 ? v[k] : v[5 + k]);
    float c[3];
    c[0] = (b4 ? a[3] : a[0]) + xchg<16>(b4 ? a[0] : a[3]);
    c[1] = (b4 ? a[4] : a[1]) + xchg<16>(b4 ? a[1] : a[4]);
    c[2] = (b4 ? 0.0f : a[2]) + xchg<16>(b4 ? a[2] : 0.0f);
    float e[2];
    e[0] = (b3 ? c[2] : c[0]) + xchg<8>(b3 ? c[0] : c[2]);
    e[1] = (b3 ? 0.0f : c[1]) + xchg<8>(b3 ? c[1] : 0.0f);
    float f = (b2 ? e[1] : e[0]) + xchg<4>(b2 ? e[0] : e[1]);
    f += xchg<2>(f);
    f += xchg<1>(f);
    return f;
}

__global__ __launch_bounds__(256) void render_bwd_kernel(
    int H, int W, int T, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
    const float* __restrict__ bg, const float* __restrict__ final_T,
    const int32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_ddepth, float* __restrict__ grad)
{
    __shared__ float4 s0[256], s1[256], s2[256];
    __shared__ uint32_t s_id[256];  // (id << 4) | quadrant mask
    __shared__ float s_acc[256 * 10];
    __shared__ int s_maxlast;

    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    const int tile = tile_for_block(blockIdx.x, gx, gy);  // same XCD-aware, balanced order as the forward
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);   // same 8x8 quadrant per wavefront
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);  // as the forward
    const uint32_t qbit = 1u << wave;
    const int my_slot = (lane & 3) == 0 ? wave_value_index(lane) : -1;  // which reduced value this lane owns
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;
    const uint32_t a = offsets[tile], b = offsets[tile + 1];
    const int n = (int)(b - a);
    if (n == 0) return;

    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
    const float Tf = inside ? final_T[pix] : 0.0f;
    const int last = inside ? n_contrib[pix] : 0;
    const float gc0 = inside ? dL_dcolor[pix] : 0.0f;
    const float gc1 = inside ? dL_dcolor[HW + pix] : 0.0f;
    const float gc2 = inside ? dL_dcolor[2 * HW + pix] : 0.0f;
    const float gd = (inside && dL_ddepth) ? dL_ddepth[pix] : 0.0f;
    const float bgdot = bg[0] * gc0 + bg[1] * gc1 + bg[2] * gc2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    if (tid == 0) s_maxlast = 0;
    __syncthreads();
    atomicMax(&s_maxlast, last);
    __syncthreads();
    const int maxlast = s_maxlast;
    if (maxlast == 0) return;

    float T_ = Tf;
    float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    float lc0 = 0, lc1 = 0, lc2 = 0, lc3 = 0, last_alpha = 0;

    for (int r = (maxlast - 1) >> 8; r >= 0; --r) {
        __syncthreads();  // previous batch fully flushed
        const int idx = (r << 8) + tid;
        uint32_t wq = 0;
        if (idx < n) {
            wq = point_list[a + idx];
            if (wq & 15u) {
                const float4* q = rec + 3 * (size_t)(wq >> 4);
                s0[tid] = q[0];
                s1[tid] = q[1];
                s2[tid] = q[2];
            }
        }
        s_id[tid] = wq;
#pragma unroll
        for (int k = 0; k < 10; ++k) s_acc[k * 256 + tid] = 0.0f;
        __syncthreads();
        const int m = min(256, min(n, maxlast) - (r << 8));
        for (int c = (m - 1) & ~63; c >= 0; c -= 64) {
          unsigned long long hits = __ballot((c + lane < m) && (s_id[c + lane] & qbit));
          if (!hits) continue;
          // software pipeline: fetch the next survivor's record from LDS while this one is processed
          int jn = c + 63 - __builtin_clzll(hits);
          float4 n0 = s0[jn], n1 = s1[jn], n2 = s2[jn];
          while (hits) {
            const int j = jn;  // back to front
            const float4 g0 = n0, g1 = n1, g2 = n2;
            hits &= ~(1ull << (j - c));
            if (hits) {
                jn = c + 63 - __builtin_clzll(hits);
                n0 = s0[jn]; n1 = s1[jn]; n2 = s2[jn];
            }
            const int contributor = (r << 8) + j + 1;
            const float dx = g0.x - pfx, dy = g0.y - pfy;
            const float power = fmaf(g0.z * dx, dx, fmaf(g0.w * dy, dy, (g1.x * dx) * dy));
            bool on = contributor <= last && power <= 0.0f && power >= g1.z;
            float G = 0.0f, alpha = 0.0f;
            if (on) {
                G = fs_exp(power);
                alpha = fminf(0.99f, g1.y * G);
                on = alpha >= 1.0f / 255.0f;
            }
            const unsigned long long onmask = __ballot(on);
            if (!onmask) continue;
            float v_mx = 0, v_my = 0, v_ca = 0, v_cb = 0, v_cc = 0, v_op = 0, v_r = 0, v_g = 0, v_b = 0, v_z = 0;
            if (on) {
                // 1 - alpha >= 0.01: one reciprocal (1 ulp) serves both divisions of the reference formula
                const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
                T_ = T_ * rinv;
                const float w = alpha * T_;
                float dL_dalpha = 0.0f;
                acc0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0; lc0 = g2.x; dL_dalpha += (g2.x - acc0) * gc0;
                acc1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1; lc1 = g2.y; dL_dalpha += (g2.y - acc1) * gc1;
                acc2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2; lc2 = g2.z; dL_dalpha += (g2.z - acc2) * gc2;
                acc3 = last_alpha * lc3 + (1.0f - last_alpha) * acc3; lc3 = g1.w; dL_dalpha += (g1.w - acc3) * gd;
                v_r = w * gc0; v_g = w * gc1; v_b = w * gc2; v_z = w * gd;
                dL_dalpha *= T_;
                last_alpha = alpha;
                dL_dalpha -= Tf * rinv * bgdot;
                const float dL_dG = g1.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float cA = -2.0f * g0.z, cC = -2.0f * g0.w, cB = -g1.x;
                const float dG_ddelx = -gdx * cA - gdy * cB;
                const float dG_ddely = -gdy * cC - gdx * cB;
                v_mx = dL_dG * dG_ddelx * ddelx_dx;
                v_my = dL_dG * dG_ddely * ddely_dy;
                v_ca = -0.5f * gdx * dx * dL_dG;
                v_cb = -0.5f * gdx * dy * dL_dG;
                v_cc = -0.5f * gdy * dy * dL_dG;
                v_op = G * dL_dalpha;
            }
            const float vals[10] = {v_mx, v_my, v_ca, v_cb, v_cc, v_op, v_r, v_g, v_b, v_z};
            const float tot = wave_sum10(vals, lane);
            if (my_slot >= 0) atomicAdd(&s_acc[my_slot * 256 + j], tot);
          }
        }
        __syncthreads();
        if (tid < m && (s_id[tid] & 15u)) {
            float* gout = grad + (size_t)(s_id[tid] >> 4) * kGradStride;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float v = s_acc[k * 256 + tid];
                if (v != 0.0f) atomicAdd(&gout[k], v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int DEG>
__device__ __forceinline__ void sh_backward(const float* __restrict__ sh, float x, float y, float z,
                                            const float* gr, float* gsh, float* gdir)
{
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float b[16];
    sh_basis<DEG>(x, y, z, b);
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) gsh[3 * k + c] = b[k] * gr[c];
    float dbx[16], dby[16], dbz[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dbx[k] = dby[k] = dbz[k] = 0.0f;
    if constexpr (DEG > 0) {
        dby[1] = -kSH1; dbz[2] = kSH1; dbx[3] = -kSH1;
    }
    if constexpr (DEG > 1) {
        dbx[4] = kSH2[0] * y; dby[4] = kSH2[0] * x;
        dby[5] = kSH2[1] * z; dbz[5] = kSH2[1] * y;
        dbx[6] = kSH2[2] * (-2.0f * x); dby[6] = kSH2[2] * (-2.0f * y); dbz[6] = kSH2[2] * (4.0f * z);
        dbx[7] = kSH2[3] * z; dbz[7] = kSH2[3] * x;
        dbx[8] = kSH2[4] * (2.0f * x); dby[8] = kSH2[4] * (-2.0f * y);
    }
    if constexpr (DEG > 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        dbx[9] = kSH3[0] * 6.0f * x * y; dby[9] = kSH3[0] * (3.0f * xx - 3.0f * yy);
        dbx[10] = kSH3[1] * y * z; dby[10] = kSH3[1] * x * z; dbz[10] = kSH3[1] * x * y;
        dbx[11] = kSH3[2] * (-2.0f * x * y); dby[11] = kSH3[2] * (4.0f * zz - xx - 3.0f * yy); dbz[11] = kSH3[2] * 8.0f * y * z;
        dbx[12] = kSH3[3] * (-6.0f * x * z); dby[12] = kSH3[3] * (-6.0f * y * z); dbz[12] = kSH3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
        dbx[13] = kSH3[4] * (4.0f * zz - 3.0f * xx - yy); dby[13] = kSH3[4] * (-2.0f * x * y); dbz[13] = kSH3[4] * 8.0f * x * z;
        dbx[14] = kSH3[5] * 2.0f * x * z; dby[14] = kSH3[5] * (-2.0f * y * z); dbz[14] = kSH3[5] * (xx - yy);
        dbx[15] = kSH3[6] * (3.0f * xx - 3.0f * yy); dby[15] = kSH3[6] * (-6.0f * x * y);
    }
    gdir[0] = gdir[1] = gdir[2] = 0.0f;
#pragma unroll
    for (int k = 1; k < NB; ++k) {
        const float s = sh[3 * k] * gr[0] + sh[3 * k + 1] * gr[1] + sh[3 * k + 2] * gr[2];
        gdir[0] += dbx[k] * s; gdir[1] += dby[k] * s; gdir[2] += dbz[k] * s;
    }
}

__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    fs_raster_dims d, const float* __restrict__ means3D, const float* __restrict__ cov3D,
    const float* __restrict__ shs, const float* __restrict__ view, const float* __restrict__ proj,
    const float* __restrict__ campos, const float* __restrict__ tanfov_dev,
    const float* __restrict__ scale_dev, GeomView g, const float* __restrict__ grad,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcov3D,
    float* __restrict__ dL_dshs, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopac,
    int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [256 * M*3] SH grads
    const int base = blockIdx.x * 256;
    const int cnt = min(256, d.N - base);
    const int t = threadIdx.x;
    const int i = base + t;
    const int per_sh = d.M * 3;
    const bool have_sh = shs != nullptr;
    const bool sh_cm = (d.flags & FS_RASTER_SH_CHANNEL_MAJOR) != 0, cov_full = (d.flags & FS_RASTER_COV_FULL) != 0;
    constexpr int kTriu[6] = {0, 1, 2, 4, 5, 8};
    const float tanfovx = tanfov_dev ? tanfov_dev[0] : d.tanfovx;
    const float tanfovy = tanfov_dev ? tanfov_dev[1] : d.tanfovy;
    const float wscale = scale_dev ? scale_dev[0] : 1.0f;

    if (t < cnt) {
        float gm[3] = {0, 0, 0}, gcov[6] = {0, 0, 0, 0, 0, 0}, gop = 0.0f;
        float gm2x = 0.0f, gm2y = 0.0f;
        float gsh[48];
        float gcol[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 48; ++k) gsh[k] = 0.0f;
        const ushort4 rc = g.rect[i];
        if (rc.z > rc.x && rc.w > rc.y) {
            const float* ga = grad + (size_t)i * kGradStride;
            float3 p = make_float3(means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]);
            if (scale_dev) { p.x = p.x * wscale; p.y = p.y * wscale; p.z = p.z * wscale; }
            gop = ga[5];
            gm2x = ga[0]; gm2y = ga[1];
            const float fx = (float)d.W / (2.0f * tanfovx), fy = (float)d.H / (2.0f * tanfovy);
            float c3[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) c3[k] = cov_full ? cov3D[9 * (size_t)i + kTriu[k]] : cov3D[6 * (size_t)i + k];
            if (scale_dev) {
                const float s2 = wscale * wscale;
#pragma unroll
                for (int k = 0; k < 6; ++k) c3[k] = c3[k] * s2;
            }
            const Cov2D cv = project_cov(view, p, c3, fx, fy, tanfovx, tanfovy);
            const float a = cv.a, b = cv.b, c = cv.c;
            const float denom = a * c - b * b;
            const float d2inv = 1.0f / (denom * denom + 0.0000001f);
            const float gcx = ga[2], gcy = ga[3], gcz = ga[4];
            float gt[3] = {0, 0, 0};
            if (d2inv != 0.0f) {
                const float dL_da = d2inv * (-c * c * gcx + 2.0f * b * c * gcy + (denom - a * c) * gcz);
                const float dL_dc = d2inv * (-a * a * gcz + 2.0f * a * b * gcy + (denom - a * c) * gcx);
                const float dL_db = d2inv * 2.0f * (b * c * gcx - (denom + 2.0f * b * b) * gcy + a * b * gcz);
                const float* m0 = cv.m0;
                const float* m1 = cv.m1;
                gcov[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
                gcov[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
                gcov[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
                gcov[1] = 2.0f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.0f * m1[0] * m1[1] * dL_dc;
                gcov[2] = 2.0f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.0f * m1[0] * m1[2] * dL_dc;
                gcov[4] = 2.0f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.0f * m1[1] * m1[2] * dL_dc;
                const float s00 = c3[0], s01 = c3[1], s02 = c3[2], s11 = c3[3], s12 = c3[4], s22 = c3[5];
                float u0[3], u1[3];
                u0[0] = m0[0] * s00 + m0[1] * s01 + m0[2] * s02;
                u0[1] = m0[0] * s01 + m0[1] * s11 + m0[2] * s12;
                u0[2] = m0[0] * s02 + m0[1] * s12 + m0[2] * s22;
                u1[0] = m1[0] * s00 + m1[1] * s01 + m1[2] * s02;
                u1[1] = m1[0] * s01 + m1[1] * s11 + m1[2] * s12;
                u1[2] = m1[0] * s02 + m1[1] * s12 + m1[2] * s22;
                float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float gM0 = 2.0f * dL_da * u0[k] + dL_db * u1[k];
                    const float gM1 = 2.0f * dL_dc * u1[k] + dL_db * u0[k];
                    gJ00 += gM0 * view[0 + 4 * k];
                    gJ02 += gM0 * view[2 + 4 * k];
                    gJ11 += gM1 * view[1 + 4 * k];
                    gJ12 += gM1 * view[2 + 4 * k];
                }
                const float tz = 1.0f / cv.tz, tz2 = tz * tz, tz3 = tz2 * tz;
                gt[0] = cv.gmx * (-fx * tz2) * gJ02;
                gt[1] = cv.gmy * (-fy * tz2) * gJ12;
                gt[2] = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2.0f * fx * cv.tx) * tz3 * gJ02 + (2.0f * fy * cv.ty) * tz3 * gJ12;
            }
            gt[2] += ga[9];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                gm[k] += view[0 + 4 * k] * gt[0] + view[1 + 4 * k] * gt[1] + view[2 + 4 * k] * gt[2];
            {
                const float4 mh = xform44(proj, p);
                const float mw = 1.0f / (mh.w + 0.0000001f);
                const float mul1 = mh.x * mw * mw, mul2 = mh.y * mw * mw;
                gm[0] += (proj[0] * mw - proj[3] * mul1) * gm2x + (proj[1] * mw - proj[3] * mul2) * gm2y;
                gm[1] += (proj[4] * mw - proj[7] * mul1) * gm2x + (proj[5] * mw - proj[7] * mul2) * gm2y;
                gm[2] += (proj[8] * mw - proj[11] * mul1) * gm2x + (proj[9] * mw - proj[11] * mul2) * gm2y;
            }
            if (!have_sh) {
                gcol[0] = ga[6]; gcol[1] = ga[7]; gcol[2] = ga[8];
            } else {
                const uint8_t cb = g.clamp[i];
                float gr[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) gr[ch] = ((cb >> ch) & 1) ? 0.0f : ga[6 + ch];
                const float dox = p.x - campos[0], doy = p.y - campos[1], doz = p.z - campos[2];
                const float len = sqrtf(dox * dox + doy * doy + doz * doz);
                const float x = dox / len, y = doy / len, z = doz / len;
                float shbuf[48];
                const float* sh = shbuf;
                if (d.flags & FS_RASTER_SH_FP16) {
                    const _Float16* hsrc = (const _Float16*)shs + (size_t)i * per_sh;
#pragma unroll
                    for (int k = 0; k < 48; ++k) shbuf[k] = k < per_sh ? (float)hsrc[sh_cm ? (k % 3) * d.M + k / 3 : k] : 0.0f;
                } else {
                    const float* fsrc = shs + (size_t)i * per_sh;
#pragma unroll
                    for (int k = 0; k < 48; ++k) shbuf[k] = k < per_sh ? fsrc[sh_cm ? (k % 3) * d.M + k / 3 : k] : 0.0f;
                }
                float gdv[3];
                switch (d.sh_degree) {
                    case 0: sh_backward<0>(sh, x, y, z, gr, gsh, gdv); break;
                    case 1: sh_backward<1>(sh, x, y, z, gr, gsh, gdv); break;
                    case 2: sh_backward<2>(sh, x, y, z, gr, gsh, gdv); break;
                    default: sh_backward<3>(sh, x, y, z, gr, gsh, gdv); break;
                }
                const float s2 = dox * dox + doy * doy + doz * doz;
                const float inv32 = 1.0f / sqrtf(s2 * s2 * s2);
                gm[0] += ((s2 - dox * dox) * gdv[0] - doy * dox * gdv[1] - doz * dox * gdv[2]) * inv32;
                gm[1] += (-dox * doy * gdv[0] + (s2 - doy * doy) * gdv[1] - doz * doy * gdv[2]) * inv32;
                gm[2] += (-dox * doz * gdv[0] - doy * doz * gdv[1] + (s2 - doz * doz) * gdv[2]) * inv32;
            }
        }
        if (scale_dev) {
            const float s2 = wscale * wscale;
#pragma unroll
            for (int k = 0; k < 3; ++k) gm[k] = gm[k] * wscale;
#pragma unroll
            for (int k = 0; k < 6; ++k) gcov[k] = gcov[k] * s2;
        }
        if (accumulate) {
#pragma unroll
            for (int k = 0; k < 3; ++k) gm[k] += dL_dmeans3D[3 * (size_t)i + k];
            gm2x += dL_dmeans2D[3 * (size_t)i];
            gm2y += dL_dmeans2D[3 * (size_t)i + 1];
#pragma unroll
            for (int k = 0; k < 6; ++k) gcov[k] += cov_full ? dL_dcov3D[9 * (size_t)i + kTriu[k]] : dL_dcov3D[6 * (size_t)i + k];
            gop += dL_dopac[i];
            if (!have_sh) {
#pragma unroll
                for (int k = 0; k < 3; ++k) gcol[k] += dL_dcolors[3 * (size_t)i + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * (size_t)i + k] = gm[k];
        dL_dmeans2D[3 * (size_t)i] = gm2x;
        dL_dmeans2D[3 * (size_t)i + 1] = gm2y;
        dL_dmeans2D[3 * (size_t)i + 2] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (cov_full) dL_dcov3D[9 * (size_t)i + kTriu[k]] = gcov[k];
            else dL_dcov3D[6 * (size_t)i + k] = gcov[k];
        }
        if (cov_full) {  // the reference reads the upper triangle only (cuda_splatting.py:126): no gradient below it
            dL_dcov3D[9 * (size_t)i + 3] = 0.0f; dL_dcov3D[9 * (size_t)i + 6] = 0.0f; dL_dcov3D[9 * (size_t)i + 7] = 0.0f;
        }
        dL_dopac[i] = gop;
        if (have_sh) {
            // gsh[] beyond the active degree is still zero; static indices keep it in registers
#pragma unroll
            for (int k = 0; k < 48; ++k)
                if (k < per_sh) lds[t * per_sh + (sh_cm ? (k % 3) * d.M + k / 3 : k)] = gsh[k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) dL_dcolors[3 * (size_t)i + k] = gcol[k];
        }
    }
    if (have_sh) {
        __syncthreads();
        float* dst = dL_dshs + (size_t)base * per_sh;
        const int total = cnt * per_sh;
        if (accumulate)
            for (int k = t; k < total; k += 256) dst[k] += lds[k];
        else
            for (int k = t; k < total; k += 256) dst[k] = lds[k];
    }
}

}  // namespace fs

using namespace fs;

FS_API int fs_raster_backward(const fs_raster_dims* dims, const float* means3D, const float* cov3D,
                              const float* shs, const float* colors_precomp, const float* bg,
                              const float* viewmatrix, const float* projmatrix, const float* campos,
                              const float* tanfov_dev, const float* scale_dev,
                              const void* geom, const void* binning, const void* image,
                              const float* dL_dcolor, const float* dL_ddepth, void* grad_scratch,
                              float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D,
                              float* dL_dshs, float* dL_dcolors, float* dL_dopacities, int accumulate,
                              void* stream_)
{
    if (!dims || !means3D || !cov3D || !bg || !viewmatrix || !projmatrix || !campos || !geom ||
        !binning || !image || !dL_dcolor || !grad_scratch || !dL_dmeans3D || !dL_dmeans2D ||
        !dL_dcov3D || !dL_dopacities)
        return FS_ERR_INVALID_ARG;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return FS_ERR_INVALID_ARG;
    if (shs ? !dL_dshs : !dL_dcolors) return FS_ERR_INVALID_ARG;
    const fs_raster_dims d = *dims;
    if (d.N <= 0) return FS_OK;
    if (shs && d.M * 3 > 48) return FS_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream_;
    const int T = num_tiles(d.H, d.W);
    const size_t P = (size_t)d.H * d.W;
    GeomView g = geom_view(const_cast<void*>(geom), d.N);
    const uint32_t* offsets = (const uint32_t*)binning;
    const uint32_t* point_list = (const uint32_t*)((const char*)binning + binning_offsets_bytes(d.H, d.W));
    const float* final_T = (const float*)image;
    const int32_t* n_contrib = (const int32_t*)((const char*)image + align_up(P * 4, 256));
    float* grad = (float*)grad_scratch;
    if (hipMemsetAsync(grad, 0, (size_t)d.N * kGradStride * 4, st) != hipSuccess) {
        set_last_error("memset grad scratch", hipGetLastError());
        return FS_ERR_LAUNCH;
    }
    const int nblk = tile_grid_blocks((d.W + kTile - 1) / kTile, (d.H + kTile - 1) / kTile);
    {
        ScopedStage prof_(kStRenderBwd, st);
        hipLaunchKernelGGL(render_bwd_kernel, dim3(nblk), dim3(256), 0, st, d.H, d.W, T, offsets,
                           point_list, g.rec, bg, final_T, n_contrib, dL_dcolor, dL_ddepth, grad);
    }
    FS_CHECK_LAUNCH("render_bwd");
    const size_t lds = shs ? (size_t)256 * d.M * 3 * sizeof(float) : 0;
    {
        ScopedStage prof_(kStPreprocessBwd, st);
        hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((d.N + 255) / 256), dim3(256), lds, st, d, means3D,
                           cov3D, shs, viewmatrix, projmatrix, campos, tanfov_dev, scale_dev, g, grad, dL_dmeans3D,
                           dL_dmeans2D,
                           dL_dcov3D, dL_dshs, dL_dcolors, dL_dopacities, accumulate);
    }
    FS_CHECK_LAUNCH("preprocess_bwd");
    return FS_OK;
}

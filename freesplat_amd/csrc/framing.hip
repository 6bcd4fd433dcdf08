// framing.hip -- per-view camera framing of the decoder in ONE launch.
//
// Replaces the ~40 small torch kernels per call that build the rasterizer's camera inputs
// (src/model/decoder/cuda_splatting.py:17-44 get_projection_matrix, :64-87 the scale-invariant rescale,
// view / full-projection matrices; src/geometry/projection.py:233-247 get_fov) for v views.  One thread per
// view, double precision inside (the work is ~300 flops per view; what matters is that no launch train
// sits in front of every render call).
#include "fs_common.h"

namespace fs {

__device__ static bool inv3(const double* m, double* o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0) return false;
    const double r = 1.0 / det;
    o[0] = c00 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c01 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c02 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
    return true;
}

// general 4x4 inverse by cofactors (the reference inverts the full matrix, not just a rigid one)
__device__ static bool inv4(const double* m, double* o)
{
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    if (det == 0.0) return false;
    const double r = 1.0 / det;
    o[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * r;   o[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * r;
    o[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * r; o[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * r;
    o[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * r;  o[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * r;
    o[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * r; o[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * r;
    o[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * r;   o[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * r;
    o[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * r; o[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * r;
    o[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * r; o[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * r;
    o[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * r; o[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * r;
    return true;
}

__device__ static double edge_angle(const double* ki, double ax, double ay, double bx, double by)
{
    // angle between the unit rays K^-1 (ax, ay, 1) and K^-1 (bx, by, 1)   (projection.py:233-247)
    double a[3], b[3];
    for (int r = 0; r < 3; ++r) {
        a[r] = ki[3 * r] * ax + ki[3 * r + 1] * ay + ki[3 * r + 2];
        b[r] = ki[3 * r] * bx + ki[3 * r + 1] * by + ki[3 * r + 2];
    }
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    double d = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) / (na * nb);
    return acos(fmin(1.0, fmax(-1.0, d)));
}

__global__ void frame_views_kernel(int v, const float* __restrict__ extrinsics, const float* __restrict__ intrinsics,
                                   const float* __restrict__ near, const float* __restrict__ far, int scale_invariant,
                                   float* __restrict__ view, float* __restrict__ full, float* __restrict__ campos,
                                   float* __restrict__ tanfov, float* __restrict__ scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v) return;
    double E[16], K[9], Ki[9], Ei[16];
    for (int k = 0; k < 16; ++k) E[k] = (double)extrinsics[16 * i + k];
    for (int k = 0; k < 9; ++k) K[k] = (double)intrinsics[9 * i + k];
    double n = (double)near[i], f = (double)far[i];
    float s = 1.0f;
    if (scale_invariant) {
        s = 1.0f / near[i];                       // fp32 like the reference: the kernels multiply by this value
        E[3] = (double)(extrinsics[16 * i + 3] * s);
        E[7] = (double)(extrinsics[16 * i + 7] * s);
        E[11] = (double)(extrinsics[16 * i + 11] * s);
        n = (double)(near[i] * s);
        f = (double)(far[i] * s);
    }
    scale[i] = s;
    campos[3 * i] = (float)E[3]; campos[3 * i + 1] = (float)E[7]; campos[3 * i + 2] = (float)E[11];
    const float nanv = __builtin_nanf("");
    double tx = nanv, ty = nanv;
    if (inv3(K, Ki)) {
        tx = tan(0.5 * edge_angle(Ki, 0.0, 0.5, 1.0, 0.5));
        ty = tan(0.5 * edge_angle(Ki, 0.5, 0.0, 0.5, 1.0));
    }
    tanfov[2 * i] = (float)tx; tanfov[2 * i + 1] = (float)ty;
    // projection (cuda_splatting.py:17-44): x/y -> (-1,1), z -> (0,1), w = z
    const double top = ty * n, right = tx * n;
    double P[16] = {0};
    P[0] = 2.0 * n / (2.0 * right); P[5] = 2.0 * n / (2.0 * top);
    P[10] = f / (f - n); P[11] = -(f * n) / (f - n); P[14] = 1.0;
    const bool ok = inv4(E, Ei);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            // view = (E^-1)^T ; full = view @ P^T = (P E^-1)^T      (row-major storage of the transposes)
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += P[4 * c + k] * Ei[4 * k + r];
            view[16 * i + 4 * r + c] = ok ? (float)Ei[4 * c + r] : nanv;
            full[16 * i + 4 * r + c] = ok ? (float)acc : nanv;
        }
}

// n row-major 4x4 matrices -> their inverses (double precision inside, rounded once; NaN rows for a singular matrix)
__global__ void invert4x4_kernel(int n, const float* __restrict__ src, float* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double m[16], o[16];
    for (int k = 0; k < 16; ++k) m[k] = (double)src[16 * i + k];
    const bool ok = inv4(m, o);
    for (int k = 0; k < 16; ++k) dst[16 * i + k] = ok ? (float)o[k] : __builtin_nanf("");
}

// generate_depth_planes (cost_volume.py:116-125) in one launch, op by op -- 1 / min, 1 / max, min^-1 + (max^-1 - min^-1) * ramp,
// 1 / that, every operation rounded as torch rounds it -- instead of the module's eight elementwise launches.  (It lives in THIS
// translation unit, built with -ffp-contract=off: cost_volume.hip is built with -ffp-contract=on, under which a * b + c written as
// one expression is fused -- one rounding less than torch's.)
__global__ void cv_depth_planes_kernel(int D, const float* __restrict__ min_depth, const float* __restrict__ max_depth,
                                       const float* __restrict__ ramp, float* __restrict__ out)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float inv_min = 1.0f / min_depth[0], inv_max = 1.0f / max_depth[0];
    const float step = (inv_max - inv_min) * ramp[d];
    out[d] = 1.0f / (inv_min + step);
}

}  // namespace fs

using namespace fs;

FS_API int fs_invert_4x4(int32_t n, const float* src, float* dst, void* stream_)
{
    if (n < 0) return FS_ERR_INVALID_ARG;
    if (n == 0) return FS_OK;
    if (!src || !dst) return FS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(invert4x4_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream_, n, src, dst);
    FS_CHECK_LAUNCH("invert_4x4");
    return FS_OK;
}

FS_API int fs_frame_views(int32_t v, const float* extrinsics, const float* intrinsics, const float* near,
                          const float* far, int32_t scale_invariant, float* view, float* full, float* campos,
                          float* tanfov, float* scale, void* stream_)
{
    if (v < 0) return FS_ERR_INVALID_ARG;
    if (v == 0) return FS_OK;
    if (!extrinsics || !intrinsics || !near || !far || !view || !full || !campos || !tanfov || !scale)
        return FS_ERR_INVALID_ARG;
    ScopedStage prof_(kStPreprocess, (hipStream_t)stream_, 0);   // (framing of a decoder call: counted with the projection stage, no launch units)
    hipLaunchKernelGGL(frame_views_kernel, dim3((v + 63) / 64), dim3(64), 0, (hipStream_t)stream_, v, extrinsics,
                       intrinsics, near, far, scale_invariant, view, full, campos, tanfov, scale);
    FS_CHECK_LAUNCH("frame_views");
    return FS_OK;
}

FS_API int fs_cost_volume_depth_planes(int32_t D, const float* min_depth, const float* max_depth, const float* ramp, float* planes,
                                       void* stream_)
{
    if (D <= 0 || !min_depth || !max_depth || !ramp || !planes) return FS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cv_depth_planes_kernel, dim3((D + 127) / 128), dim3(128), 0, (hipStream_t)stream_, D, min_depth, max_depth, ramp,
                       planes);
    FS_CHECK_LAUNCH("cost_volume_depth_planes");
    return FS_OK;
}

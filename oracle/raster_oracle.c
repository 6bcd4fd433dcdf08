/*
 * oracle/raster_oracle.c -- CPU restatement of the tile-based 3D-Gaussian rasterizer
 * (forward + backward) that FreeSplat calls through
 *   /root/reference/src/model/decoder/cuda_splatting.py:100-127
 * (module `diff_gaussian_rasterization_depth`, requirements.txt:17, NOT vendored in the
 * reference tree; third-party: JonathonLuiten/diff-gaussian-rasterization-w-depth, unpinned
 * git HEAD).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (freesplat_amd/) may import, link or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference holds no golden vectors / tests for this boundary and the
 * CUDA extension cannot be built or run here (no source, no nvcc).  The algorithm below is
 * the published 3DGS tile rasterizer (Kerbl et al. 2023, "3D Gaussian Splatting", sec. 6 +
 * appendix) with the depth-accumulating fork's extra output, restated from SURVEY.md
 * Appendix A, and anchored on the reference's call site contract and on known-answer tests
 * (tests/test_raster_oracle.py) + finite differences + an independent torch-autograd
 * restatement (oracle/raster_dense_torch.py).
 *
 * Arithmetic contract (shared with the HIP kernels so that the FORWARD is bit-reproducible):
 *   - compiled with -ffp-contract=off; every fused multiply-add is an explicit fmaf();
 *   - exp() is fso_exp() below: IEEE fma / add + ldexp only (no libm exp);
 *   - sqrt and division are IEEE correctly rounded.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FSO_TILE 16
#define FSO_API __attribute__((visibility("default")))

typedef struct {
    int N;          /* gaussians */
    int M;          /* SH coefficients per channel present in `shs` ([N, M, 3]) */
    int H, W;       /* image */
    int sh_degree;  /* active degree (<= 3) */
    float tanfovx, tanfovy;
    float bg[3];
    float view[16]; /* torch tensor bytes of `viewmatrix` (= world->cam, transposed; column-major for us) */
    float proj[16]; /* torch tensor bytes of `projmatrix` (full projection, transposed) */
    float campos[3];
} fso_params;

/* ---- deterministic exp, x <= 0 in practice -------------------------------------------
 * n = round(x log2 e) through the magic-number addition (ONE fused rounding: fmaf), reduction r = x - n ln2 in ONE fma
 * with the fp32 ln2 (the product n * fl(ln2) is exact inside the fma; fl(ln2) - ln2 = 1.9e-9, so for the |n| <= 8 of
 * every alpha that can reach 1/255 the reduced argument is off by <= 1.5e-8 = 0.25 ulp of the result -- the hi/lo split
 * of round 2 bought nothing there), degree-5 minimax polynomial in Horner form (max relative error 1.7e-7 = 2.8 ulp on
 * [-6, 0], measured in fp32; 3.7e-7 on [-80, 0]), scaling by 2^n.  Every operation is an IEEE fma / add / ldexp: the HIP kernels
 * evaluate exactly this sequence (fs_common.h: fs_exp, fs_exp2_of_neg). */
static inline float fso_exp(float x)
{
    if (x < -80.0f) return 0.0f;
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f);   /* 1.5 * 2^23: the sum's low bits are round(x log2 e) */
    const float n = t - 12582912.0f;
    const float r = fmaf(n, -0.693147182464599609375f, x);
    float p = fmaf(r, 0.008290314115583897f, 0.04189793020486832f);
    p = fmaf(r, p, 0.1666763573884964f);
    p = fmaf(r, p, 0.4999915063381195f);
    p = fmaf(r, p, 0.9999997019767761f);
    p = fmaf(r, p, 1.0f);
    return ldexpf(p, (int)n);
}
FSO_API float fso_exp_public(float x) { return fso_exp(x); }

/* Blend-loop exponential: 0 = the arithmetic contract above (what the HIP kernels implement, the default),
 * 1 = libm expf().  Mode 1 exists ONLY to quantify how much of the image depends on the private contract
 * (tests/test_raster_hip.py::test_exp_contract_sensitivity, bench.py `parity.exp_contract`): it stands in for
 * "some other correctly-behaving exp", e.g. the CUDA extension's __expf. */
static int g_exp_mode = 0;
FSO_API void fso_set_exp_mode(int mode) { g_exp_mode = mode; }
static inline float fso_blend_exp(float x) { return g_exp_mode ? expf(x) : fso_exp(x); }

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static inline void xf43(const float* m, const float* p, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xf44(const float* m, const float* p, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* SH basis values b[k] for direction (x,y,z), k < (deg+1)^2; colour = sum_k b[k]*sh[k] */
static inline void sh_basis(int deg, float x, float y, float z, float* b)
{
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y;
        b[2] = SH_C1 * z;
        b[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy;
            b[5] = SH_C2[1] * yz;
            b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            b[7] = SH_C2[3] * xz;
            b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.0f * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

/* cov2D (a,b,c) incl. +0.3 dilation; also returns the clamped t and M = J*R rows (2x3) */
static inline void cov2d(const fso_params* P, const float* mean, const float* c3, float* abc,
                         float* t_out, float* Mrow0, float* Mrow1, float* gmul)
{
    const float fx = P->W / (2.0f * P->tanfovx), fy = P->H / (2.0f * P->tanfovy);
    float t[3];
    xf43(P->view, mean, t);
    const float limx = 1.3f * P->tanfovx, limy = 1.3f * P->tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    gmul[0] = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    gmul[1] = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float j00 = fx / t[2], j02 = -(fx * t[0]) / (t[2] * t[2]);
    const float j11 = fy / t[2], j12 = -(fy * t[1]) / (t[2] * t[2]);
    const float* V = P->view; /* R[k][i] = V[k + 4*i] */
    for (int i = 0; i < 3; ++i) {
        Mrow0[i] = j00 * V[0 + 4 * i] + j02 * V[2 + 4 * i];
        Mrow1[i] = j11 * V[1 + 4 * i] + j12 * V[2 + 4 * i];
    }
    /* S = Sigma (symmetric) */
    const float s00 = c3[0], s01 = c3[1], s02 = c3[2], s11 = c3[3], s12 = c3[4], s22 = c3[5];
    float u0[3], u1[3]; /* u = M * Sigma */
    u0[0] = Mrow0[0] * s00 + Mrow0[1] * s01 + Mrow0[2] * s02;
    u0[1] = Mrow0[0] * s01 + Mrow0[1] * s11 + Mrow0[2] * s12;
    u0[2] = Mrow0[0] * s02 + Mrow0[1] * s12 + Mrow0[2] * s22;
    u1[0] = Mrow1[0] * s00 + Mrow1[1] * s01 + Mrow1[2] * s02;
    u1[1] = Mrow1[0] * s01 + Mrow1[1] * s11 + Mrow1[2] * s12;
    u1[2] = Mrow1[0] * s02 + Mrow1[1] * s12 + Mrow1[2] * s22;
    abc[0] = (u0[0] * Mrow0[0] + u0[1] * Mrow0[1] + u0[2] * Mrow0[2]) + 0.3f;
    abc[1] = u0[0] * Mrow1[0] + u0[1] * Mrow1[1] + u0[2] * Mrow1[2];
    abc[2] = (u1[0] * Mrow1[0] + u1[1] * Mrow1[1] + u1[2] * Mrow1[2]) + 0.3f;
    t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2];
}

/*
 * Per-Gaussian preprocess.  Outputs (all caller-allocated):
 *   depths[N], radii[N], means2D[2N], conic_opacity[4N], rgb[3N], clamped[3N] (u8),
 *   rect[4N] (minx,miny,maxx,maxy in tiles), tiles_touched[N].
 * Returns total number of (gaussian,tile) instances.
 */
FSO_API long fso_preprocess(const fso_params* P, const float* means3D, const float* cov3D,
                            const float* shs, const float* colors_precomp, const float* opacities,
                            float* depths, int* radii, float* means2D, float* conic_opacity,
                            float* rgb, unsigned char* clamped, int* rect, unsigned* tiles_touched)
{
    const int N = P->N, W = P->W, H = P->H;
    const int gx = (W + FSO_TILE - 1) / FSO_TILE, gy = (H + FSO_TILE - 1) / FSO_TILE;
    long total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        depths[i] = 0.0f;
        means2D[2 * i] = means2D[2 * i + 1] = 0.0f;
        for (int k = 0; k < 4; ++k) { conic_opacity[4 * i + k] = 0.0f; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0.0f; clamped[3 * i + k] = 0; }
        const float* p = means3D + 3 * i;
        float pv[3];
        xf43(P->view, p, pv);
        if (pv[2] <= 0.2f) continue;
        float ph[4];
        xf44(P->proj, p, ph);
        const float pw = 1.0f / (ph[3] + 0.0000001f);
        const float ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        float abc[3], t[3], m0[3], m1[3], gm[2];
        cov2d(P, p, cov3D + 6 * i, abc, t, m0, m1, gm);
        const float det = abc[0] * abc[2] - abc[1] * abc[1];
        if (det == 0.0f) continue;
        const float det_inv = 1.0f / det;
        const float cx = abc[2] * det_inv, cy = -abc[1] * det_inv, cz = abc[0] * det_inv;
        const float mid = 0.5f * (abc[0] + abc[2]);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + sq, l2 = mid - sq;
        const float my_radius = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
        const float px = ((ndcx + 1.0f) * W - 1.0f) * 0.5f;
        const float py = ((ndcy + 1.0f) * H - 1.0f) * 0.5f;
        const int r = (int)my_radius;
        const int rminx = imin(gx, imax(0, (int)((px - r) / FSO_TILE)));
        const int rminy = imin(gy, imax(0, (int)((py - r) / FSO_TILE)));
        const int rmaxx = imin(gx, imax(0, (int)((px + r + FSO_TILE - 1) / FSO_TILE)));
        const int rmaxy = imin(gy, imax(0, (int)((py + r + FSO_TILE - 1) / FSO_TILE)));
        const int area = (rmaxx - rminx) * (rmaxy - rminy);
        if (area == 0) continue;
        if (colors_precomp) {
            for (int c = 0; c < 3; ++c) rgb[3 * i + c] = colors_precomp[3 * i + c];
        } else {
            float d[3] = {p[0] - P->campos[0], p[1] - P->campos[1], p[2] - P->campos[2]};
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] = d[0] / len; d[1] = d[1] / len; d[2] = d[2] / len;
            float b[16];
            sh_basis(P->sh_degree, d[0], d[1], d[2], b);
            const int nb = (P->sh_degree + 1) * (P->sh_degree + 1);
            const float* sh = shs + (size_t)i * P->M * 3;
            for (int c = 0; c < 3; ++c) {
                float acc = b[0] * sh[c];
                for (int k = 1; k < nb; ++k) acc = acc + b[k] * sh[3 * k + c];
                acc = acc + 0.5f;
                clamped[3 * i + c] = acc < 0.0f;
                rgb[3 * i + c] = fmaxf(acc, 0.0f);
            }
        }
        depths[i] = pv[2];
        radii[i] = r;
        means2D[2 * i] = px; means2D[2 * i + 1] = py;
        conic_opacity[4 * i] = cx; conic_opacity[4 * i + 1] = cy; conic_opacity[4 * i + 2] = cz;
        conic_opacity[4 * i + 3] = opacities[i];
        rect[4 * i] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy;
        tiles_touched[i] = (unsigned)area;
        total += area;
    }
    return total;
}

typedef struct { uint32_t dbits; uint32_t id; } fso_inst;
static int inst_cmp(const void* a, const void* b)
{
    const fso_inst* x = (const fso_inst*)a; const fso_inst* y = (const fso_inst*)b;
    if (x->dbits != y->dbits) return x->dbits < y->dbits ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

/*
 * Binning + ordering: stable ascending sort of (tile, depth bits) with
 * emission order (gaussian index) as tie-break.  ranges[2T], point_list[I].
 */
FSO_API void fso_bin(const fso_params* P, const float* depths, const int* radii, const int* rect,
                     unsigned* ranges, unsigned* point_list, long I)
{
    const int N = P->N;
    const int gx = (P->W + FSO_TILE - 1) / FSO_TILE, gy = (P->H + FSO_TILE - 1) / FSO_TILE;
    const int T = gx * gy;
    unsigned* cnt = (unsigned*)calloc((size_t)T + 1, sizeof(unsigned));
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) cnt[y * gx + x + 1]++;
    }
    for (int t = 0; t < T; ++t) cnt[t + 1] += cnt[t];
    for (int t = 0; t < T; ++t) { ranges[2 * t] = cnt[t]; ranges[2 * t + 1] = cnt[t + 1]; }
    fso_inst* inst = (fso_inst*)malloc(sizeof(fso_inst) * (size_t)(I > 0 ? I : 1));
    unsigned* cur = (unsigned*)malloc(sizeof(unsigned) * (size_t)T);
    memcpy(cur, cnt, sizeof(unsigned) * (size_t)T);
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        uint32_t db; memcpy(&db, depths + i, 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
                unsigned s = cur[y * gx + x]++;
                inst[s].dbits = db; inst[s].id = (uint32_t)i;
            }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < T; ++t) {
        unsigned a = cnt[t], b = cnt[t + 1];
        if (b - a > 1) qsort(inst + a, b - a, sizeof(fso_inst), inst_cmp);
        for (unsigned s = a; s < b; ++s) point_list[s] = inst[s].id;
    }
    free(cur); free(inst); free(cnt);
}

/* Per-pixel front-to-back blend. */
FSO_API void fso_render(const fso_params* P, const unsigned* ranges, const unsigned* point_list,
                        const float* means2D, const float* conic_opacity, const float* rgb,
                        const float* depths, float* out_color, float* out_depth, float* out_alpha,
                        float* final_T, int* n_contrib)
{
    const int W = P->W, H = P->H;
    const int gx = (W + FSO_TILE - 1) / FSO_TILE, gy = (H + FSO_TILE - 1) / FSO_TILE;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const unsigned a = ranges[2 * tile], b = ranges[2 * tile + 1];
        for (int ly = 0; ly < FSO_TILE; ++ly)
            for (int lx = 0; lx < FSO_TILE; ++lx) {
                const int px = tx * FSO_TILE + lx, py = ty * FSO_TILE + ly;
                if (px >= W || py >= H) continue;
                const float pfx = (float)px, pfy = (float)py;
                float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, D = 0.0f;
                int k = 0, last = 0;
                for (unsigned s = a; s < b; ++s) {
                    ++k;
                    const unsigned g = point_list[s];
                    const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
                    const float* co = conic_opacity + 4 * g;
                    const float hA = -0.5f * co[0], hC = -0.5f * co[2], nB = -co[1];
                    const float power = fmaf(hA * dx, dx, dy * fmaf(hC, dy, nB * dx));
                    if (power > 0.0f) continue;
                    const float alpha = fminf(0.99f, co[3] * fso_blend_exp(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1.0f - alpha);
                    if (test_T < 0.0001f) break;
                    const float w = alpha * T;
                    C0 = fmaf(rgb[3 * g], w, C0);
                    C1 = fmaf(rgb[3 * g + 1], w, C1);
                    C2 = fmaf(rgb[3 * g + 2], w, C2);
                    D = fmaf(depths[g], w, D);
                    T = test_T;
                    last = k;
                }
                const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
                final_T[pix] = T;
                n_contrib[pix] = last;
                out_color[pix] = fmaf(T, P->bg[0], C0);
                out_color[HW + pix] = fmaf(T, P->bg[1], C1);
                out_color[2 * HW + pix] = fmaf(T, P->bg[2], C2);
                out_depth[pix] = D;
                out_alpha[pix] = 1.0f - T;
            }
    }
}

/*
 * Backward of the blend.  dL_dcolor [3,H,W], dL_ddepth [H,W] or NULL.
 * Accumulates into dL_dmean2D[2N], dL_dconic[3N] (x, y(half), z), dL_dopacity[N], dL_drgb[3N], dL_dz[N]
 * (buffers zeroed by the caller), in double, to serve as the accuracy reference for the float-atomic
 * HIP path.
 *
 * Parallel AND deterministic: tiles run on OpenMP threads, each summing the per-pixel terms of its own list
 * entries (fixed pixel order) into a private [entries x 10] block of doubles; the per-Gaussian totals are then
 * formed by one sequential pass over the instance list (tile order).  No atomics, no thread-count dependence.
 */
#define FSO_NPART 10
FSO_API void fso_render_backward(const fso_params* P, const unsigned* ranges,
                                 const unsigned* point_list, const float* means2D,
                                 const float* conic_opacity, const float* rgb, const float* depths,
                                 const float* final_T, const int* n_contrib,
                                 const float* dL_dcolor, const float* dL_ddepth, double* dL_dmean2D,
                                 double* dL_dconic, double* dL_dopacity, double* dL_drgb,
                                 double* dL_dz)
{
    const int W = P->W, H = P->H;
    const int gx = (W + FSO_TILE - 1) / FSO_TILE, gy = (H + FSO_TILE - 1) / FSO_TILE;
    const size_t HW = (size_t)H * W;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const size_t I = ranges[2 * (gx * gy - 1) + 1];
    double* part = (double*)calloc((I > 0 ? I : 1) * FSO_NPART, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const unsigned a = ranges[2 * tile], b = ranges[2 * tile + 1];
        for (int ly = 0; ly < FSO_TILE; ++ly)
            for (int lx = 0; lx < FSO_TILE; ++lx) {
                const int px = tx * FSO_TILE + lx, py = ty * FSO_TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pix = (size_t)py * W + px;
                const float pfx = (float)px, pfy = (float)py;
                const float Tf = final_T[pix];
                float T = Tf;
                const int last = n_contrib[pix];
                const float g0 = dL_dcolor[pix], g1 = dL_dcolor[HW + pix], g2 = dL_dcolor[2 * HW + pix];
                const float gd = dL_ddepth ? dL_ddepth[pix] : 0.0f;
                const float bgdot = P->bg[0] * g0 + P->bg[1] * g1 + P->bg[2] * g2;
                float acc[4] = {0, 0, 0, 0}, lastc[4] = {0, 0, 0, 0}, last_alpha = 0.0f;
                for (int k = last; k >= 1; --k) {
                    const size_t slot = (size_t)a + (unsigned)k - 1;
                    const unsigned g = point_list[slot];
                    const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
                    const float* co = conic_opacity + 4 * g;
                    const float hA = -0.5f * co[0], hC = -0.5f * co[2], nB = -co[1];
                    const float power = fmaf(hA * dx, dx, dy * fmaf(hC, dy, nB * dx));
                    if (power > 0.0f) continue;
                    const float G = fso_blend_exp(power);
                    const float alpha = fminf(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.0f - alpha);
                    const float w = alpha * T;
                    const float c[4] = {rgb[3 * g], rgb[3 * g + 1], rgb[3 * g + 2], depths[g]};
                    const float gg[4] = {g0, g1, g2, gd};
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 4; ++ch) {
                        acc[ch] = last_alpha * lastc[ch] + (1.0f - last_alpha) * acc[ch];
                        lastc[ch] = c[ch];
                        dL_dalpha += (c[ch] - acc[ch]) * gg[ch];
                    }
                    double* q = part + slot * FSO_NPART;
                    q[6] += (double)(w * g0);
                    q[7] += (double)(w * g1);
                    q[8] += (double)(w * g2);
                    q[9] += (double)(w * gd);
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-Tf / (1.0f - alpha)) * bgdot;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    q[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    q[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    q[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    q[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    q[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    q[5] += (double)(G * dL_dalpha);
                }
            }
    }
    for (size_t s = 0; s < I; ++s) {   /* fixed order: tile-major, then list position */
        const unsigned g = point_list[s];
        const double* q = part + s * FSO_NPART;
        dL_dmean2D[2 * g] += q[0]; dL_dmean2D[2 * g + 1] += q[1];
        dL_dconic[3 * g] += q[2]; dL_dconic[3 * g + 1] += q[3]; dL_dconic[3 * g + 2] += q[4];
        dL_dopacity[g] += q[5];
        dL_drgb[3 * g] += q[6]; dL_drgb[3 * g + 1] += q[7]; dL_drgb[3 * g + 2] += q[8];
        dL_dz[g] += q[9];
    }
    free(part);
}

/*
 * Per-Gaussian second stage of the backward: conic -> cov2D -> Sigma and t;
 * mean2D -> mean through the perspective divide; rgb -> SH and view direction; z -> mean.
 * Inputs are the float-cast accumulators of fso_render_backward.
 */
FSO_API void fso_preprocess_backward(const fso_params* P, const float* means3D,
                                     const float* cov3D, const float* shs, int have_sh,
                                     const int* radii, const unsigned char* clamped,
                                     const float* dL_dmean2D, const float* dL_dconic,
                                     const float* dL_dopacity_in, const float* dL_drgb,
                                     const float* dL_dz, float* dL_dmeans3D, float* dL_dcov3D,
                                     float* dL_dshs, float* dL_dcolors_precomp,
                                     float* dL_dopacities)
{
    const int N = P->N;
    const float fx = P->W / (2.0f * P->tanfovx), fy = P->H / (2.0f * P->tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float* gm3 = dL_dmeans3D + 3 * i;
        gm3[0] = gm3[1] = gm3[2] = 0.0f;
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0.0f;
        dL_dopacities[i] = 0.0f;
        if (have_sh) for (int k = 0; k < P->M * 3; ++k) dL_dshs[(size_t)i * P->M * 3 + k] = 0.0f;
        else for (int k = 0; k < 3; ++k) dL_dcolors_precomp[3 * i + k] = 0.0f;
        if (radii[i] <= 0) continue;
        const float* p = means3D + 3 * i;
        dL_dopacities[i] = dL_dopacity_in[i];

        /* ---- conic -> cov2D -> Sigma, t ---- */
        float abc[3], t[3], m0[3], m1[3], gmul[2];
        cov2d(P, p, cov3D + 6 * i, abc, t, m0, m1, gmul);
        const float a = abc[0], b = abc[1], c = abc[2];
        const float denom = a * c - b * b;
        const float d2inv = 1.0f / (denom * denom + 0.0000001f);
        const float gcx = dL_dconic[3 * i], gcy = dL_dconic[3 * i + 1], gcz = dL_dconic[3 * i + 2];
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float gt[3] = {0, 0, 0};
        if (d2inv != 0.0f) {
            dL_da = d2inv * (-c * c * gcx + 2.0f * b * c * gcy + (denom - a * c) * gcz);
            dL_dc = d2inv * (-a * a * gcz + 2.0f * a * b * gcy + (denom - a * c) * gcx);
            dL_db = d2inv * 2.0f * (b * c * gcx - (denom + 2.0f * b * b) * gcy + a * b * gcz);
            float* gc = dL_dcov3D + 6 * i;
            gc[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
            gc[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
            gc[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
            gc[1] = 2.0f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.0f * m1[0] * m1[1] * dL_dc;
            gc[2] = 2.0f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.0f * m1[0] * m1[2] * dL_dc;
            gc[4] = 2.0f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.0f * m1[1] * m1[2] * dL_dc;
            /* dL/dM, M = J R (2x3) */
            const float* c3 = cov3D + 6 * i;
            const float s00 = c3[0], s01 = c3[1], s02 = c3[2], s11 = c3[3], s12 = c3[4], s22 = c3[5];
            float u0[3], u1[3];
            u0[0] = m0[0] * s00 + m0[1] * s01 + m0[2] * s02;
            u0[1] = m0[0] * s01 + m0[1] * s11 + m0[2] * s12;
            u0[2] = m0[0] * s02 + m0[1] * s12 + m0[2] * s22;
            u1[0] = m1[0] * s00 + m1[1] * s01 + m1[2] * s02;
            u1[1] = m1[0] * s01 + m1[1] * s11 + m1[2] * s12;
            u1[2] = m1[0] * s02 + m1[1] * s12 + m1[2] * s22;
            float gM0[3], gM1[3];
            for (int k = 0; k < 3; ++k) {
                gM0[k] = 2.0f * dL_da * u0[k] + dL_db * u1[k];
                gM1[k] = 2.0f * dL_dc * u1[k] + dL_db * u0[k];
            }
            const float* V = P->view;
            /* dL/dJ_pk = sum_i gM_p[i] * R[k][i], R[k][i] = V[k + 4 i] */
            float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
            for (int k = 0; k < 3; ++k) {
                gJ00 += gM0[k] * V[0 + 4 * k];
                gJ02 += gM0[k] * V[2 + 4 * k];
                gJ11 += gM1[k] * V[1 + 4 * k];
                gJ12 += gM1[k] * V[2 + 4 * k];
            }
            const float tz = 1.0f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            gt[0] = gmul[0] * (-fx * tz2) * gJ02;
            gt[1] = gmul[1] * (-fy * tz2) * gJ12;
            gt[2] = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2.0f * fx * t[0]) * tz3 * gJ02 + (2.0f * fy * t[1]) * tz3 * gJ12;
        }
        /* depth output gradient flows into t.z as well (p_view.z) */
        gt[2] += dL_dz[i];
        {
            const float* V = P->view; /* dL/dp_i = sum_k R[k][i] gt[k] */
            for (int k = 0; k < 3; ++k)
                gm3[k] += V[0 + 4 * k] * gt[0] + V[1 + 4 * k] * gt[1] + V[2 + 4 * k] * gt[2];
        }
        /* ---- mean2D -> mean ---- */
        {
            const float* M = P->proj;
            float mh[4];
            xf44(M, p, mh);
            const float mw = 1.0f / (mh[3] + 0.0000001f);
            const float mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
            const float g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
            gm3[0] += (M[0] * mw - M[3] * mul1) * g2x + (M[1] * mw - M[3] * mul2) * g2y;
            gm3[1] += (M[4] * mw - M[7] * mul1) * g2x + (M[5] * mw - M[7] * mul2) * g2y;
            gm3[2] += (M[8] * mw - M[11] * mul1) * g2x + (M[9] * mw - M[11] * mul2) * g2y;
        }
        /* ---- colour -> SH, direction ---- */
        if (!have_sh) {
            for (int k = 0; k < 3; ++k) dL_dcolors_precomp[3 * i + k] = dL_drgb[3 * i + k];
            continue;
        }
        {
            const int deg = P->sh_degree;
            float dorig[3] = {p[0] - P->campos[0], p[1] - P->campos[1], p[2] - P->campos[2]};
            const float len = sqrtf(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
            const float x = dorig[0] / len, y = dorig[1] / len, z = dorig[2] / len;
            float gr[3];
            for (int ch = 0; ch < 3; ++ch) gr[ch] = clamped[3 * i + ch] ? 0.0f : dL_drgb[3 * i + ch];
            float bas[16];
            sh_basis(deg, x, y, z, bas);
            const int nb = (deg + 1) * (deg + 1);
            float* gsh = dL_dshs + (size_t)i * P->M * 3;
            const float* sh = shs + (size_t)i * P->M * 3;
            for (int k = 0; k < nb; ++k)
                for (int ch = 0; ch < 3; ++ch) gsh[3 * k + ch] = bas[k] * gr[ch];
            /* d basis / d(x,y,z) */
            float dbx[16] = {0}, dby[16] = {0}, dbz[16] = {0};
            if (deg > 0) {
                dby[1] = -SH_C1; dbz[2] = SH_C1; dbx[3] = -SH_C1;
                if (deg > 1) {
                    dbx[4] = SH_C2[0] * y; dby[4] = SH_C2[0] * x;
                    dby[5] = SH_C2[1] * z; dbz[5] = SH_C2[1] * y;
                    dbx[6] = SH_C2[2] * (-2.0f * x); dby[6] = SH_C2[2] * (-2.0f * y); dbz[6] = SH_C2[2] * (4.0f * z);
                    dbx[7] = SH_C2[3] * z; dbz[7] = SH_C2[3] * x;
                    dbx[8] = SH_C2[4] * (2.0f * x); dby[8] = SH_C2[4] * (-2.0f * y);
                    if (deg > 2) {
                        const float xx = x * x, yy = y * y, zz = z * z;
                        dbx[9] = SH_C3[0] * 6.0f * x * y; dby[9] = SH_C3[0] * (3.0f * xx - 3.0f * yy);
                        dbx[10] = SH_C3[1] * y * z; dby[10] = SH_C3[1] * x * z; dbz[10] = SH_C3[1] * x * y;
                        dbx[11] = SH_C3[2] * (-2.0f * x * y); dby[11] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy); dbz[11] = SH_C3[2] * 8.0f * y * z;
                        dbx[12] = SH_C3[3] * (-6.0f * x * z); dby[12] = SH_C3[3] * (-6.0f * y * z); dbz[12] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                        dbx[13] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); dby[13] = SH_C3[4] * (-2.0f * x * y); dbz[13] = SH_C3[4] * 8.0f * x * z;
                        dbx[14] = SH_C3[5] * 2.0f * x * z; dby[14] = SH_C3[5] * (-2.0f * y * z); dbz[14] = SH_C3[5] * (xx - yy);
                        dbx[15] = SH_C3[6] * (3.0f * xx - 3.0f * yy); dby[15] = SH_C3[6] * (-6.0f * x * y);
                    }
                }
            }
            float gd[3] = {0, 0, 0};
            for (int k = 1; k < nb; ++k) {
                const float s = sh[3 * k] * gr[0] + sh[3 * k + 1] * gr[1] + sh[3 * k + 2] * gr[2];
                gd[0] += dbx[k] * s; gd[1] += dby[k] * s; gd[2] += dbz[k] * s;
            }
            /* through normalisation */
            const float s2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
            const float inv32 = 1.0f / sqrtf(s2 * s2 * s2);
            gm3[0] += ((s2 - dorig[0] * dorig[0]) * gd[0] - dorig[1] * dorig[0] * gd[1] - dorig[2] * dorig[0] * gd[2]) * inv32;
            gm3[1] += (-dorig[0] * dorig[1] * gd[0] + (s2 - dorig[1] * dorig[1]) * gd[1] - dorig[2] * dorig[1] * gd[2]) * inv32;
            gm3[2] += (-dorig[0] * dorig[2] * gd[0] - dorig[1] * dorig[2] * gd[1] + (s2 - dorig[2] * dorig[2]) * gd[2]) * inv32;
        }
    }
}

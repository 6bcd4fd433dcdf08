// fs_common.h -- shared device/host helpers of libfreesplat_hip.so (gfx950 only).
//
// Arithmetic contract for the rasterizer forward (see DESIGN.md "bit-reproducible forward"):
// the translation units are compiled with -ffp-contract=off, every fused multiply-add is an
// explicit fmaf(), exp() is fs_exp() (IEEE fma / add + an integer add on the exponent only), sqrt and
// division are the correctly rounded forms hipcc emits by default.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/freesplat_amd.h"

#define FS_API extern "C" __attribute__((visibility("default")))

namespace fs {

constexpr int kTile = FS_TILE;       // 16x16 px
constexpr int kTilePix = kTile * kTile;  // 256 threads = 4 wavefronts
constexpr int kWave = 64;

void set_last_error(const char* what, hipError_t e);

#define FS_CHECK_LAUNCH(what)                                   \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) {                                \
            fs::set_last_error(what, e__);                      \
            return FS_ERR_LAUNCH;                               \
        }                                                       \
    } while (0)

// ---- optional per-kernel event timing (fs_profile_*; off by default, zero cost when off) ------
enum Stage {
    kStPreprocess = 0, kStTileScan, kStRender, kStRenderBwd, kStPreprocessBwd,
    kStCostVolume, kStPtf, kStEncoderTail, kNumStages
};
struct ScopedStage {
    ScopedStage(Stage s, hipStream_t st, int units = 1);  // units: views covered by the launch (reported as launches)
    ~ScopedStage();
    int slot_;
    hipStream_t st_;
};

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- deterministic exp (x <= 0 on the hot path): the same operation sequence as the CPU checker's exp ----
constexpr float kExpC1 = 0.9999997019767761f, kExpC2 = 0.4999915063381195f, kExpC3 = 0.1666763573884964f, kExpC4 = 0.04189793020486832f, kExpC5 = 0.008290314115583897f;
__device__ __forceinline__ float fs_exp(float x)
{
    if (x < -80.0f) return 0.0f;
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    const float r = fmaf(n, -0.693147182464599609375f, x);   // one fma with the fp32 ln2 (exact product; see the oracle's note)
    float p = fmaf(r, kExpC5, kExpC4);
    p = fmaf(r, p, kExpC3);
    p = fmaf(r, p, kExpC2);
    p = fmaf(r, p, kExpC1);
    p = fmaf(r, p, 1.0f);
    return ldexpf(p, (int)n);
}

// ---- phase timestamps inside kernels (debug builds: make EXTRA=-DFS_PHASE_TRACE; profiles/phase_trace.py) ----
// FS_PT(kernel, k): thread 0 of the first 4096 workgroups stamps wall_clock64() (100 MHz) into slot k (< 8).
#ifdef FS_PHASE_TRACE
constexpr int kPtKernels = 4, kPtBlocks = 4096, kPtSlots = 10;
static __device__ unsigned long long g_phase_trace[kPtKernels * kPtBlocks * kPtSlots];  // one per translation unit
#define FS_PT(kern, k)                                                                                  \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < fs::kPtBlocks)                                             \
            fs::g_phase_trace[((kern) * fs::kPtBlocks + blockIdx.x) * fs::kPtSlots + (k)] = wall_clock64(); \
    } while (0)
#else
#define FS_PT(kern, k) do {} while (0)
#endif

// positional encoding of two scalars, 6 octaves each, (sin, cos) interleaved (encoder_freesplat.py:62-77): 24 floats.
// Hardware v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS and are accurate to ~1e-6 on the fraction; the
// libm forms cost ~40 VALU each, 48 of them per fused pair.  The arguments are accumulated densities and weights times
// 2^k, k <= 5 -- they grow with every fused view and reach the hundreds in a long fold --, so the revolutions are formed
// in two pieces: rev = fl(x / 2 pi) and its exact residual (one fma with the hi part of 1 / 2 pi, one with the lo part);
// 2^k rev is exact, its integer part is removed exactly, and the residual is added to the small remaining fraction.
// The phase then carries ~1e-8 revolutions of error whatever |x| is (a plain fp32 x / 2 pi: up to 3.6e-5 revolutions =
// 2e-4 rad at x = 1234, k = 5 -- the bound the round-2 comment got wrong).
__device__ __forceinline__ void rev2pi(float x, float& rev, float& res)
{
    constexpr float kInv2PiHi = 0.15915494f, kInv2PiLo = 6.4206382e-9f;    // 1 / (2 pi) = hi + lo
    rev = x * kInv2PiHi;
    res = fmaf(x, kInv2PiLo, fmaf(x, kInv2PiHi, -rev));
}
__device__ __forceinline__ void pos_enc2(float a, float b, float* __restrict__ out)  // 24 floats
{
    float ra, ea, rb, eb;
    rev2pi(a, ra, ea);
    rev2pi(b, rb, eb);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float f = (float)(1 << k);
        const float ta = ra * f, tb = rb * f;                       // (exact)
        const float pa = (ta - rintf(ta)) + ea * f, pb = (tb - rintf(tb)) + eb * f;
        out[2 * k] = __builtin_amdgcn_sinf(pa); out[2 * k + 1] = __builtin_amdgcn_cosf(pa);
        out[12 + 2 * k] = __builtin_amdgcn_sinf(pb); out[12 + 2 * k + 1] = __builtin_amdgcn_cosf(pb);
    }
}

// ptf_gru.hip: the GRU of a fold step (n pairs, or at most n_max with the count in counts[1] on the device)
int launch_ptf_gru(int n_max, const int32_t* counts, const float* cat, const float* tables, float* fused, hipStream_t st);
// ... with the input rows gathered and encoded inside the kernel (no [n,176] intermediate): the fold's own path
int launch_ptf_gru_gather(int n_max, const int32_t* counts, const long long* fuse_idx, const long long* fuse_pix,
                          const float* G, const float* R, const float* O, const float* g_i, const float* rho_i,
                          const float* om_i, const float* tables, float* fused, bool out_after_keep, hipStream_t st,
                          float* save_side = nullptr, float* save_act = nullptr, float* save_cat = nullptr);
// (save_side / save_act: the training fold keeps the hidden activations and gates for fs_ptf_gru_backward_saved)
// (out_after_keep: `fused` is the out state's G array and pair t is written to its row counts[0] + t)

// LDS operations of one wavefront execute in order: between phases of a wavefront-private LDS exchange only the
// compiler must be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- packed fp32 (v_pk_{add,mul,fma}_f32): two Gaussians per lane-instruction in the blend loops ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }

// fs_exp() of two arguments in [-80, 0], bit-identical to the scalar form: the low bits of the biased sum t are the
// integer n for the exponent (ldexpf == integer add on the exponent field while the result stays normal).  Arguments
// outside the range give garbage, never a trap.
__device__ __forceinline__ f32x2 fs_exp2_nonpos(f32x2 x)
{
    const f32x2 magic = splat2(12582912.0f);
    const f32x2 t = fma2(x, splat2(1.44269504088896341f), magic);
    const f32x2 n = t - magic;
    const f32x2 r = fma2(n, splat2(-0.693147182464599609375f), x);
    f32x2 q = fma2(r, splat2(kExpC5), splat2(kExpC4));
    q = fma2(r, q, splat2(kExpC3));
    q = fma2(r, q, splat2(kExpC2));
    q = fma2(r, q, splat2(kExpC1));
    q = fma2(r, q, splat2(1.0f));
    const i32x2 e = __builtin_bit_cast(i32x2, q) + (__builtin_bit_cast(i32x2, t) << 23);
    return __builtin_bit_cast(f32x2, e);
}

// The same exp for the NEGATED argument q = -x >= 0 (the blend loops carry -power so that "0 >= power >= skip threshold"
// is ONE unsigned compare of the bit patterns): every product / sum below is the exact negation of its counterpart
// above, so the result has the same bits as fs_exp2_nonpos(-q).
__device__ __forceinline__ f32x2 fs_exp2_of_neg(f32x2 q)
{
    const f32x2 magic = splat2(12582912.0f);
    const f32x2 t = fma2(q, splat2(-1.44269504088896341f), magic);   // == fma(-q, log2 e, magic)
    const f32x2 n = t - magic;
    const f32x2 r = fma2(n, splat2(-0.693147182464599609375f), -q);
    f32x2 p = fma2(r, splat2(kExpC5), splat2(kExpC4));
    p = fma2(r, p, splat2(kExpC3));
    p = fma2(r, p, splat2(kExpC2));
    p = fma2(r, p, splat2(kExpC1));
    p = fma2(r, p, splat2(1.0f));
    const i32x2 e = __builtin_bit_cast(i32x2, p) + (__builtin_bit_cast(i32x2, t) << 23);
    return __builtin_bit_cast(f32x2, e);
}
// exp(-q) of the blend loops: FAST = hardware v_exp_f32 (2^x, <= 1 ulp; FS_RASTER_FAST_EXP), else the contract exp.
// On gfx950 a wave64 v_fma_f32 issues in ~2.6 cycles, v_pk_fma_f32 in ~5.2 (no packed-fp32 throughput gain on this
// chip), v_exp_f32 in ~8.4 (profiles/tools/valu_rates.hip): contract exp = 26 cycles per value, hardware = 11.
template <bool FAST>
__device__ __forceinline__ f32x2 blend_exp_of_neg(f32x2 q)
{
    if constexpr (FAST) {
        const f32x2 t = q * splat2(-1.44269504088896341f);
        return (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    } else {
        return fs_exp2_of_neg(q);
    }
}
// Guard band of the hardware-exp mode around the alpha threshold 1/255 (bits 0x3B808081): alpha from v_exp_f32 and
// alpha from the contract exp differ by at most ~7 ulp (rounding of q * log2(e) at |q| <= 5.6: 2.8 ulp; the constant: 1;
// v_exp_f32: 1; the contract exp: 1; the product with the opacity: 0.5 each), so a value outside
// [1/255 - 16 ulp, 1/255 + 16 ulp) falls on the same side of the threshold in both modes.  A step with any value
// inside the band is re-evaluated with the contract exp: the accept / reject decisions of the hardware-exp mode are
// those of the exact mode, what remains is the ~4e-7 relative difference of the accepted alphas.
constexpr uint32_t kAlphaMinBits = 0x3B808081u;  // 1.0f / 255.0f
constexpr uint32_t kAlphaGuardUlps = 16u;
__device__ __forceinline__ float alpha_guard_lo() { return __uint_as_float(kAlphaMinBits - kAlphaGuardUlps); }
__device__ __forceinline__ float alpha_guard_hi() { return __uint_as_float(kAlphaMinBits + kAlphaGuardUlps); }

// Skip threshold of a record (r1.z: power below it cannot reach alpha >= 1/255) as the bit pattern the unsigned compare
// uses: -threshold for a negative threshold; 0 otherwise (opacity <= 1/255: only q == +0 passes, and fails the alpha test).
__device__ __forceinline__ float skip_bits(float thr) { return thr < 0.0f ? -thr : 0.0f; }

// ---- camera transforms: torch hands the matrices over transposed => column-major here -------
__device__ __forceinline__ float3 xform43(const float* __restrict__ m, float3 p)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform44(const float* __restrict__ m, float3 p)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// ---- real SH basis up to degree 3 ------------------------------------------------------------
constexpr float kSH0 = 0.28209479177387814f;
constexpr float kSH1 = 0.4886025119029199f;
__device__ constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f,
                                      0.31539156525252005f, -1.0925484305920792f,
                                      0.5462742152960396f};
__device__ constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                      -0.4570457994644658f, 0.3731763325901154f,
                                      -0.4570457994644658f, 1.445305721320277f,
                                      -0.5900435899266435f};

template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b)
{
    b[0] = kSH0;
    if constexpr (DEG > 0) {
        b[1] = -kSH1 * y;
        b[2] = kSH1 * z;
        b[3] = -kSH1 * x;
    }
    if constexpr (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = kSH2[0] * xy;
        b[5] = kSH2[1] * yz;
        b[6] = kSH2[2] * (2.0f * zz - xx - yy);
        b[7] = kSH2[3] * xz;
        b[8] = kSH2[4] * (xx - yy);
        if constexpr (DEG > 2) {
            b[9] = kSH3[0] * y * (3.0f * xx - yy);
            b[10] = kSH3[1] * xy * z;
            b[11] = kSH3[2] * y * (4.0f * zz - xx - yy);
            b[12] = kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = kSH3[4] * x * (4.0f * zz - xx - yy);
            b[14] = kSH3[5] * z * (xx - yy);
            b[15] = kSH3[6] * x * (xx - 3.0f * yy);
        }
    }
}

// EWA projection of the 3D covariance: M = J*R (2x3), cov2D = M Sigma M^T (+0.3 px^2 on the
// diagonal).  Shared by the forward preprocess and the per-Gaussian backward.
struct Cov2D {
    float a, b, c;       // dilated 2D covariance
    float m0[3], m1[3];  // rows of M
    float tx, ty, tz;    // clamped view-space position
    float gmx, gmy;      // 0 where the tan-fov clamp is active
};
__device__ __forceinline__ Cov2D project_cov(const float* __restrict__ V, float3 mean,
                                             const float* c3, float fx, float fy, float tanfovx,
                                             float tanfovy)
{
    Cov2D o;
    float3 t = xform43(V, mean);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    o.gmx = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    o.gmy = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float j00 = fx / t.z, j02 = -(fx * t.x) / (t.z * t.z);
    const float j11 = fy / t.z, j12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.m0[i] = j00 * V[0 + 4 * i] + j02 * V[2 + 4 * i];
        o.m1[i] = j11 * V[1 + 4 * i] + j12 * V[2 + 4 * i];
    }
    const float s00 = c3[0], s01 = c3[1], s02 = c3[2], s11 = c3[3], s12 = c3[4], s22 = c3[5];
    const float u00 = o.m0[0] * s00 + o.m0[1] * s01 + o.m0[2] * s02;
    const float u01 = o.m0[0] * s01 + o.m0[1] * s11 + o.m0[2] * s12;
    const float u02 = o.m0[0] * s02 + o.m0[1] * s12 + o.m0[2] * s22;
    const float u10 = o.m1[0] * s00 + o.m1[1] * s01 + o.m1[2] * s02;
    const float u11 = o.m1[0] * s01 + o.m1[1] * s11 + o.m1[2] * s12;
    const float u12 = o.m1[0] * s02 + o.m1[1] * s12 + o.m1[2] * s22;
    o.a = (u00 * o.m0[0] + u01 * o.m0[1] + u02 * o.m0[2]) + 0.3f;
    o.b = u00 * o.m1[0] + u01 * o.m1[1] + u02 * o.m1[2];
    o.c = (u10 * o.m1[0] + u11 * o.m1[1] + u12 * o.m1[2]) + 0.3f;
    o.tx = t.x; o.ty = t.y; o.tz = t.z;
    return o;
}

// ---- rasterizer buffer layouts (opaque to the caller; see fs_raster_buffer_sizes) ------------
// geom: [N] x 3 float4 screen-space records, [N] ushort4 tile rects, [N] u8 clamp bits.
//   r0 = {px, py, -A/2, -C/2}   r1 = {-B, opacity, power_skip_threshold, view_z}
//   r2 = {r, g, b, 0}           (A,B,C) = conic (inverse dilated 2D covariance)
// (round 2 also kept packed quadrant masks and a dense depth array for the separate emit pass: 12 B per Gaussian
//  written and re-read per view; the binning now happens in the projection launch, on registers)
struct GeomView {
    float4* rec;
    ushort4* rect;
    uint8_t* clamp;
};
__host__ __device__ inline size_t geom_bytes(int N)
{
    return align_up((size_t)N * 48, 256) + align_up((size_t)N * 8, 256) + align_up((size_t)N, 256);
}
__host__ __device__ inline GeomView geom_view(void* base, int N)
{
    char* p = (char*)base;
    GeomView g;
    g.rec = (float4*)p;
    p += align_up((size_t)N * 48, 256);
    g.rect = (ushort4*)p;
    p += align_up((size_t)N * 8, 256);
    g.clamp = (uint8_t*)p;
    return g;
}
// Tile -> workgroup order of the per-tile kernels (sort_blend, render_bwd).  Workgroup b runs on XCD b % 8
// (hardware round-robin).  Tiles are grouped in 4x4 super-tiles -- neighbouring tiles share most of their
// Gaussians, so their records stay in one L2 -- and super-tile s goes to XCD s % 8: every XCD gets super-tiles
// from all over the image.  (One contiguous band of tiles per XCD left the XCDs up to 25 % apart in work and
// the slowest one set the kernel time: profiles/render_trace.py.)
__host__ __device__ inline int tile_grid_blocks(int gx, int gy)
{
    const int ns = ((gx + 3) >> 2) * ((gy + 3) >> 2);
    return 8 * ((ns + 7) >> 3) * 16;
}
__device__ __forceinline__ int tile_for_block(int b, int gx, int gy)
{
    const int i = b >> 3;
    const int s = (b & 7) + 8 * (i >> 4), w = i & 15;
    const int sgx = (gx + 3) >> 2;
    const int tx = (s % sgx) * 4 + (w & 3), ty = (s / sgx) * 4 + (w >> 2);
    return (tx < gx && ty < gy) ? ty * gx + tx : -1;
}

// binning: [T+1] u32 tile offsets (exclusive scan of per-tile counts), then [cap] u32 ids.
__host__ __device__ inline int num_tiles(int H, int W)
{
    return ((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
}
__host__ __device__ inline size_t binning_offsets_bytes(int H, int W)
{
    return align_up((size_t)(num_tiles(H, W) + 1) * 4, 256);
}
// image: [P] f32 final_T, [P] i32 n_contrib
// scratch: [T] u32 tile counts (= the binning pass' slot cursors), [T x tile_capacity] u64 keys -- a FIXED number of key
// slots per tile, so that the projection pass can bin in the same launch (no count -> scan -> second pass over the
// Gaussians): 4x the mean list length at the instance capacity, a power of two >= 2048 (HBM is 288 GB; config 3:
// 8192 slots x 4941 tiles x 8 B = 324 MB of address space per stream, of which the 46 MB of real keys are touched).
__host__ __device__ inline uint32_t tile_capacity(long long cap, int T)
{
    const unsigned long long want = (4ull * (unsigned long long)(cap > 0 ? cap : 1) + (unsigned long long)T - 1ull) / (unsigned long long)(T > 0 ? T : 1);
    unsigned long long c = 2048;
    while (c < want && c < (1ull << 26)) c <<= 1;
    while (c > 1 && c * (unsigned long long)(T > 0 ? T : 1) > 0xFFFFFFFFull) c >>= 1;   // key slots are indexed in 32 bits
    return (uint32_t)c;
}

}  // namespace fs

// Per-voxel arithmetic of the unprojection path, shared by every kernel.
//
// The operation ORDER below is the contract: it reproduces, rounding step by
// rounding step, what the reference's chain of fp32 torch ops computes on the
// CPU (tempo/models/project_layer.py:111-126 and :255-268 of the reference,
// plus mmpose's SimpleCameraTorch.world_to_pixel / affine_transform_torch and
// torch's grid_sample unnormalisation).  Element-wise torch ops round after
// every operation.  The three K=3 matmuls (X @ R, [x y 1] @ K^T, trans @
// [x y 1]^T) go through the host's BLAS, whose rounding differs between CPUs
// (probed with torch 2.10 + oneMKL: an FMA chain ascending in k that starts
// with a plain multiply on Intel, three products and two adds ascending in k
// on AMD EPYC), so every function takes the flavour as `unfused`;
// tempo_amd.numerics probes the host so that the kernels reproduce torch on
// the machine they run on.  The file is compiled with -ffp-contract=off and
// every fused step is an explicit fmaf(), so neither hipcc nor gcc may
// re-associate or fuse anything.
//
// TEMPO_HD lets tests/emul build the same functions with gcc for a CPU-side
// bit-level comparison against the oracle; the product only ever runs them
// on the GPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TEMPO_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define TEMPO_HD static inline
#endif

// Packed per-(frame, view) parameter record, 32 floats = 128 B; the same
// layout is documented for callers in include/tempo_unproject.h.
typedef struct TempoCam {
  float R[9];      // R[k][j] row-major, used as X(row) @ R
  float T[3];      // T_w2c = -(T^T @ R)
  float K[6];      // K[0][0..2], K[1][0..2]  (pix = [x y 1] @ K^T)
  float kd[3];     // radial k0 k1 k2
  float pd[2];     // tangential p0 p1
  float has_dist;  // != 0 when k and p are present
  float trans[6];  // 2x3 image affine (get_affine_transform)
  float width;     // center[0] * 2
  float height;    // center[1] * 2
} TempoCam;
enum { CAM_FLOATS = 32 };

// flavour bits of the three matmul sites (argument `mm_unfused` of the ABI)
enum { TEMPO_UNFUSED_R = 1, TEMPO_UNFUSED_K = 2, TEMPO_UNFUSED_A = 4, TEMPO_UNFUSED_ALL = 7 };

// Rare slow paths (numerators outside the range the fast division is proven for)
// must stay behind a branch: left as a plain `if`, the compiler if-converts them
// and every lane pays for the full IEEE division (div_scale/rcp/div_fmas/
// div_fixup) on every call.  A wave-level test keeps the branch uniform.
#if defined(__HIP_DEVICE_COMPILE__)
#define TEMPO_ANY_LANE(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define TEMPO_ANY_LANE(c) (c)
#endif

// ---- pairs: the x and y coordinate go through identical operation chains, so
// they are carried as 2-vectors; on gfx950 these become packed-fp32 VALU ops
// (v_pk_mul/add/fma_f32: two lanes' worth per issue slot)
#if defined(__HIPCC__)
typedef float tempo_f2 __attribute__((ext_vector_type(2)));
#define TEMPO_FMA2(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#else
typedef float tempo_f2 __attribute__((vector_size(8)));
static inline tempo_f2 TEMPO_FMA2(tempo_f2 a, tempo_f2 b, tempo_f2 c) {
  tempo_f2 r = {fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
  return r;
}
#endif
TEMPO_HD tempo_f2 tempo_splat(float v) {
  tempo_f2 r = {v, v};
  return r;
}
TEMPO_HD tempo_f2 tempo_pair(float a, float b) {
  tempo_f2 r = {a, b};
  return r;
}

// sum_k a_k * b_k, k ascending, in the flavour the host BLAS uses
TEMPO_HD float tempo_dot3(float a0, float b0, float a1, float b1, float a2,
                          float b2, int unfused) {
  if (unfused) return (a0 * b0 + a1 * b1) + a2 * b2;
  return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}
TEMPO_HD tempo_f2 tempo_dot3_2(tempo_f2 a0, tempo_f2 b0, tempo_f2 a1, tempo_f2 b1,
                               tempo_f2 a2, tempo_f2 b2, int unfused) {
  if (unfused) return (a0 * b0 + a1 * b1) + a2 * b2;
  return TEMPO_FMA2(a2, b2, TEMPO_FMA2(a1, b1, a0 * b0));
}

// ---- exact IEEE fp32 division by way of one double-precision multiply.
// a / b for fp32 a, b and rb = 1/b in double, accurate to about 1 ulp (the host's
// 1.0 / (double)b, or tempo_rcp64_newton): RN32((double)a * rb) == RN32(a / b).
// Why: the product carries a relative error below 2^-50.5 (reciprocal 2^-52, product
// 2^-53).  A quotient of two 24-bit significands A/B that is not itself a rounding
// boundary m of the fp32 grid (a 25-bit significand M) is at least |A - M*B| / (M*B)
// >= 2^-49 away from it, relatively, so the error cannot carry the product across a
// boundary, and it is never ON one for a normal result (M odd cannot divide A < 2^24;
// the overflow threshold is such a boundary as well).  Zero (of either sign), infinities
// and NaN go through unchanged; a denormal numerator is exact in double.  Only results
// in the fp32 DENORMAL range can sit exactly on a tie of the coarser grid there: callers
// whose result is used as it is (the masked mean) test for a denormal result
// (tempo_is_denormal) and redo those with the plain division; the sample-grid chain
// needs no test, because its quotients next go through q * 2 - 1, which rounds to -1
// for every |q| < 2^-126 alike.  Three full-rate instructions per value (convert,
// multiply, convert) and no range guard, against five packed-pair steps and a six-compare
// guard for the two-correction fp32 form this replaces.
TEMPO_HD float tempo_div_r64(float a, double rb) { return (float)((double)a * rb); }
TEMPO_HD tempo_f2 tempo_div_r64_2(tempo_f2 a, double rb0, double rb1) {
  tempo_f2 q = {tempo_div_r64(a[0], rb0), tempo_div_r64(a[1], rb1)};
  return q;
}
// nonzero with a zero exponent field
TEMPO_HD int tempo_is_denormal(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_classf(v, 0x010 | 0x080);  // -denormal | +denormal
#else
  return v != 0.0f && fabsf(v) < 0x1p-126f;
#endif
}
// 1 / b in double from an fp32 estimate: two Newton steps (2^-23 -> 2^-46 -> below
// half an ulp of the double)
TEMPO_HD double tempo_rcp64_newton(float b, float rcp_est) {
  const double d = (double)b;
  double r = (double)rcp_est;
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// [xc, yc] / zc: reciprocal estimate, one Newton step, then the same two
// correction steps -- the operation sequence of the compiler's own fp32
// division expansion minus its range scaling, shared between both numerators.
// Outside the range where no scaling is needed (and for a zero numerator) the
// plain division runs.  `rcp_est` may be any approximation within a few ulp.
TEMPO_HD tempo_f2 tempo_div_shared2(tempo_f2 a, float b, float rcp_est) {
  const float e = fmaf(-b, rcp_est, 1.0f);
  const float r1 = fmaf(e, rcp_est, rcp_est);
  const tempo_f2 rb = tempo_splat(r1), nb = tempo_splat(-b);
  tempo_f2 q = a * rb;
  tempo_f2 r = TEMPO_FMA2(nb, q, a);
  q = TEMPO_FMA2(r, rb, q);
  r = TEMPO_FMA2(nb, q, a);
  q = TEMPO_FMA2(r, rb, q);
  const float m0 = fabsf(a[0]), m1 = fabsf(a[1]), mb = fabsf(b);
  const float hi = fmaxf(fmaxf(m0, m1), mb), lo = fminf(fminf(m0, m1), mb);
  const int bad = !(hi <= 0x1p60f) || !(lo >= 0x1p-60f);
  if (TEMPO_ANY_LANE(bad)) {
    const tempo_f2 slow = a / tempo_splat(b);
    if (bad) q = slow;
  }
  return q;
}
TEMPO_HD float tempo_rcp_est(float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(b);
#else
  return 1.0f / b;
#endif
}

// per-call constants of the pixel -> sample-grid chain
typedef struct TempoGeom {
  float hm[2];     // the module's heatmap_size (w, h)
  float img[2];    // its image_size (w, h)
  float hm1[2];    // heatmap_size - 1
  double r_img[2];  // 1 / img
  double r_hm1[2];  // 1 / (heatmap_size - 1)
} TempoGeom;

static inline TempoGeom tempo_make_geom(float hm_w, float hm_h, float img_w, float img_h) {
  TempoGeom g;
  g.hm[0] = hm_w; g.hm[1] = hm_h;
  g.img[0] = img_w; g.img[1] = img_h;
  g.hm1[0] = hm_w - 1.0f; g.hm1[1] = hm_h - 1.0f;
  for (int i = 0; i < 2; i++) {
    g.r_img[i] = 1.0 / (double)g.img[i];
    g.r_hm1[i] = 1.0 / (double)g.hm1[i];
  }
  return g;
}

typedef struct TempoPix {
  float px, py;  // source-image pixel (before the clamp)
  int inside;    // bounds test of project_layer.py:113-115
} TempoPix;

TEMPO_HD float tempo_nan_clamp(float v, float lo, float hi) {
  // torch.clamp: NaN stays NaN
  float r = v < lo ? lo : v;
  r = r > hi ? hi : r;
  return v != v ? v : r;
}

// clamp for values whose NaN case the caller patches itself (fminf/fmaxf drop a
// NaN operand; the GPU's median-of-three returns the minimum then: both give lo)
TEMPO_HD float tempo_clamp(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
  return fminf(fmaxf(v, lo), hi);
#endif
}

// world point -> source-image pixel (SimpleCameraTorch.world_to_pixel)
TEMPO_HD TempoPix tempo_world_to_pixel(const TempoCam* cam, float x, float y,
                                       float z, int unfused) {
  const float* R = cam->R;
  // X @ R_w2c then + T_w2c (separate add)
  const tempo_f2 xy_c = tempo_dot3_2(tempo_splat(x), tempo_pair(R[0], R[1]), tempo_splat(y),
                                     tempo_pair(R[3], R[4]), tempo_splat(z),
                                     tempo_pair(R[6], R[7]), unfused) +
                        tempo_pair(cam->T[0], cam->T[1]);
  const float zc = tempo_dot3(x, R[2], y, R[5], z, R[8], unfused) + cam->T[2];
  // _X = X / X[..., 2:]   (no z > 0 test, as the reference)
  tempo_f2 n = tempo_div_shared2(xy_c, zc, tempo_rcp_est(zc));
  // the third homogeneous coordinate is zc / zc: 1 for finite non-zero zc (then
  // 1 * K[.][2] is exact and the last step of the product below is a plain add),
  // NaN for 0, +-inf and NaN -- which makes both pixel coordinates NaN
  const int zbad = !(zc != 0.0f && zc - zc == 0.0f);
  if (cam->has_dist != 0.0f) {
    const float k0 = cam->kd[0], k1 = cam->kd[1], k2 = cam->kd[2];
    const tempo_f2 sq = n * n;
    const float r2 = sq[0] + sq[1];
    const float r4 = r2 * r2;  // r2**2
    const float r6 = r4 * r2;  // r2**3 = (r2*r2)*r2
    const float radial = 1.0f + ((k0 * r2 + k1 * r4) + k2 * r6);
    const tempo_f2 tp = tempo_pair(cam->pd[1], cam->pd[0]) * n;  // p1*x, p0*y
    const float tang = 2.0f * (tp[0] + tp[1]);
    const float rt = radial + tang;
    n = n * tempo_splat(rt) + tempo_splat(r2) * tempo_pair(cam->pd[1], cam->pd[0]);
  }
  const float* K = cam->K;
  const tempo_f2 kc0 = tempo_pair(K[0], K[3]), kc1 = tempo_pair(K[1], K[4]);
  const tempo_f2 kc2 = tempo_pair(K[2], K[5]);
  const tempo_f2 n0 = tempo_splat(n[0]), n1 = tempo_splat(n[1]);
  const tempo_f2 pix =
      unfused ? (n0 * kc0 + n1 * kc1) + kc2 : TEMPO_FMA2(n1, kc1, n0 * kc0) + kc2;
  TempoPix o;
  o.px = zbad ? __builtin_nanf("") : pix[0];
  o.py = zbad ? __builtin_nanf("") : pix[1];
  o.inside = (o.px >= 0.0f) & (o.py >= 0.0f) & (o.px < cam->width) & (o.py < cam->height);
  return o;
}

// source-image pixel -> grid_sample coordinate in [-1.1, 1.1]
// (project_layer.py:116-126)
TEMPO_HD void tempo_pixel_to_grid(const TempoCam* cam, float px, float py,
                                  const TempoGeom* g, int unfused, float* gx, float* gy) {
  const float wmax = cam->width > cam->height ? cam->width : cam->height;
  // torch.clamp keeps NaN; the affine below mixes px and py (0 * NaN = NaN), so
  // one NaN coordinate makes both outputs NaN: clamp NaN-blind, patch at the end
  const int poisoned = (px != px) | (py != py);
  px = tempo_clamp(px, -1.0f, wmax);
  py = tempo_clamp(py, -1.0f, wmax);
  const float* t = cam->trans;
  // torch.mm(t, [x y 1]^T)
  tempo_f2 a = tempo_dot3_2(tempo_pair(t[0], t[3]), tempo_splat(px), tempo_pair(t[1], t[4]),
                            tempo_splat(py), tempo_pair(t[2], t[5]), tempo_splat(1.0f),
                            unfused);
  // xy * [w, h] / image_size;  / [w-1, h-1] * 2 - 1
  a = tempo_div_r64_2(a * tempo_pair(g->hm[0], g->hm[1]), g->r_img[0], g->r_img[1]);
  a = tempo_div_r64_2(a, g->r_hm1[0], g->r_hm1[1]);
  // (a * 2) - 1: the doubling is exact, so one FMA rounds identically
  a = TEMPO_FMA2(a, tempo_splat(2.0f), tempo_splat(-1.0f));
  const float nan = __builtin_nanf("");
  *gx = poisoned ? nan : tempo_clamp(a[0], -1.1f, 1.1f);
  *gy = poisoned ? nan : tempo_clamp(a[1], -1.1f, 1.1f);
}

// grid_sample(align_corners=False) unnormalisation ((g + 1) * size - 1) / 2.
// torch's vectorised CPU kernel evaluates it as (g + 1) * (size / 2) - 0.5 and
// its build contracts the multiply-subtract into one FMA (probed bit-exact).
TEMPO_HD float tempo_unnormalize(float g, float size) {
  return fmaf(g + 1.0f, size * 0.5f, -0.5f);
}

// Bilinear footprint of one sample: top-left texel and the four weights.
// NaN coordinates give x0 = y0 = 0 and NaN weights, which poison the sample
// like torch's arithmetic does (0 * NaN).
typedef struct TempoTap {
  int x0, y0;            // top-left texel, may be outside the map
  float nw, ne, sw, se;  // weights of (x0,y0) (x0+1,y0) (x0,y0+1) (x0+1,y0+1)
} TempoTap;

TEMPO_HD TempoTap tempo_bilinear_tap(float gx, float gy, int W, int H) {
  float ix = tempo_unnormalize(gx, (float)W);
  float iy = tempo_unnormalize(gy, (float)H);
  float fx = floorf(ix), fy = floorf(iy);
  float w = ix - fx, e = 1.0f - w;
  float n = iy - fy, s = 1.0f - n;
  TempoTap t;
  // |g| <= 1.1 keeps ix, iy far inside int range; NaN falls to 0
  t.x0 = (ix == ix) ? (int)fx : 0;
  t.y0 = (iy == iy) ? (int)fy : 0;
  t.nw = s * e;
  t.ne = s * w;
  t.sw = n * e;
  t.se = n * w;
  return t;
}

// The same footprint as its top-left texel and the two fractions; the weights
// are then (1-n)(1-w), (1-n)w, n(1-w), nw exactly as above.
TEMPO_HD void tempo_bilinear_frac(float gx, float gy, int W, int H, int* x0, int* y0,
                                  float* w, float* n) {
  const float ix = tempo_unnormalize(gx, (float)W);
  const float iy = tempo_unnormalize(gy, (float)H);
  const float fx = floorf(ix), fy = floorf(iy);
  *w = ix - fx;
  *n = iy - fy;
  *x0 = (ix == ix) ? (int)fx : 0;
  *y0 = (iy == iy) ? (int)fy : 0;
}

// sample = fma(se_val, se, fma(sw_val, sw, fma(ne_val, ne, nw_val * nw)))
// (the contraction torch's CPU grid_sample is built with); a texel outside the
// map reads as 0 (padding_mode='zeros')
TEMPO_HD float tempo_bilinear_mix(float a, float b, float c, float d, float nw,
                                  float ne, float sw, float se) {
  return fmaf(d, se, fmaf(c, sw, fmaf(b, ne, a * nw)));
}

// Order in which torch's CPU sum(dim=-1) adds the V per-view terms (probed on
// torch 2.10: rows shorter than one 8-float vector go through a 4-way
// interleaved scalar sum, longer rows add the tail first, then the vector's
// lanes).  slot -> view; slots >= V are empty.  From V = 16 on the vector path
// also pre-adds lane pairs; there the kernel sums views in index order, which
// differs from torch by rounding only.
TEMPO_HD int tempo_view_of_slot(int slot, int V) {
  if (slot >= V) return -1;
  if (V < 8) {
    if (V < 5 || slot == 0) return slot;
    return slot <= V - 4 ? slot + 3 : slot - (V - 4);
  }
  if (V < 16) return slot < V - 8 ? slot + 8 : slot - (V - 8);
  return slot;
}

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

// sum_k a_k * b_k, k ascending, in the flavour the host BLAS uses
TEMPO_HD float tempo_dot3(float a0, float b0, float a1, float b1, float a2,
                          float b2, int unfused) {
  if (unfused) return (a0 * b0 + a1 * b1) + a2 * b2;
  return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
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

// clamp for values known not to be NaN (fminf/fmaxf drop a NaN operand)
TEMPO_HD float tempo_clamp(float v, float lo, float hi) {
  return fminf(fmaxf(v, lo), hi);
}

// world point -> source-image pixel (SimpleCameraTorch.world_to_pixel)
TEMPO_HD TempoPix tempo_world_to_pixel(const TempoCam* cam, float x, float y,
                                       float z, int unfused) {
  const float* R = cam->R;
  // X @ R_w2c then + T_w2c (separate add)
  float xc = tempo_dot3(x, R[0], y, R[3], z, R[6], unfused) + cam->T[0];
  float yc = tempo_dot3(x, R[1], y, R[4], z, R[7], unfused) + cam->T[1];
  float zc = tempo_dot3(x, R[2], y, R[5], z, R[8], unfused) + cam->T[2];
  // _X = X / X[..., 2:]   (no z > 0 test, as the reference)
  float xn = xc / zc;
  float yn = yc / zc;
  // zc / zc: 1 for finite non-zero zc, NaN for 0 and +-inf
  float zn = (zc != 0.0f && zc - zc == 0.0f) ? 1.0f : __builtin_nanf("");
  if (cam->has_dist != 0.0f) {
    const float k0 = cam->kd[0], k1 = cam->kd[1], k2 = cam->kd[2];
    const float p0 = cam->pd[0], p1 = cam->pd[1];
    float r2 = xn * xn + yn * yn;
    float r4 = r2 * r2;  // r2**2
    float r6 = r4 * r2;  // r2**3 = (r2*r2)*r2
    float radial = 1.0f + ((k0 * r2 + k1 * r4) + k2 * r6);
    float tang = 2.0f * (p1 * xn + p0 * yn);
    float rt = radial + tang;
    float xd = xn * rt + r2 * p1;
    float yd = yn * rt + r2 * p0;
    xn = xd;
    yn = yd;
  }
  const float* K = cam->K;
  TempoPix o;
  o.px = tempo_dot3(xn, K[0], yn, K[1], zn, K[2], unfused);
  o.py = tempo_dot3(xn, K[3], yn, K[4], zn, K[5], unfused);
  o.inside = (o.px >= 0.0f) & (o.py >= 0.0f) & (o.px < cam->width) & (o.py < cam->height);
  return o;
}

// source-image pixel -> grid_sample coordinate in [-1.1, 1.1]
// (project_layer.py:116-126).  hm_w/hm_h are the module's heatmap_size,
// img_w/img_h its image_size.
TEMPO_HD void tempo_pixel_to_grid(const TempoCam* cam, float px, float py,
                                  float hm_w, float hm_h, float img_w, float img_h,
                                  int unfused, float* gx, float* gy) {
  const float wmax = cam->width > cam->height ? cam->width : cam->height;
  // torch.clamp keeps NaN; the affine below mixes px and py (0 * NaN = NaN), so
  // one NaN coordinate makes both outputs NaN: clamp NaN-blind, patch at the end
  const int poisoned = (px != px) | (py != py);
  px = tempo_clamp(px, -1.0f, wmax);
  py = tempo_clamp(py, -1.0f, wmax);
  const float* t = cam->trans;
  // torch.mm(t, [x y 1]^T)
  float ax = tempo_dot3(t[0], px, t[1], py, t[2], 1.0f, unfused);
  float ay = tempo_dot3(t[3], px, t[4], py, t[5], 1.0f, unfused);
  ax = (ax * hm_w) / img_w;
  ay = (ay * hm_h) / img_h;
  float sx = (ax / (hm_w - 1.0f)) * 2.0f - 1.0f;
  float sy = (ay / (hm_h - 1.0f)) * 2.0f - 1.0f;
  const float nan = __builtin_nanf("");
  *gx = poisoned ? nan : tempo_clamp(sx, -1.1f, 1.1f);
  *gy = poisoned ? nan : tempo_clamp(sy, -1.1f, 1.1f);
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

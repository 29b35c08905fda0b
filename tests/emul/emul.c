// TEST INFRASTRUCTURE: builds tempo_amd/csrc/unproject_math.h with gcc so the
// per-voxel arithmetic of the HIP kernels can be compared bit-for-bit with the
// torch oracle on a machine without a GPU.  Never loaded by the product.
#include <stdint.h>

#include "../../tempo_amd/csrc/unproject_math.h"

// pts [n,3] -> pix [n,2], inside [n], grid [n,2]
void emul_project(const float* cam_, const float* pts, long n, float hm_w,
                  float hm_h, float img_w, float img_h, int unf, float* pix, int* inside,
                  float* grid) {
  const TempoCam* cam = (const TempoCam*)cam_;
  const TempoGeom geom = tempo_make_geom(hm_w, hm_h, img_w, img_h);
  for (long i = 0; i < n; i++) {
    TempoPix p = tempo_world_to_pixel(cam, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], unf);
    pix[2 * i] = p.px;
    pix[2 * i + 1] = p.py;
    inside[i] = p.inside;
    tempo_pixel_to_grid(cam, p.px, p.py, &geom, unf, &grid[2 * i], &grid[2 * i + 1]);
  }
}

void emul_view_order(int V, int* order) {
  for (int s = 0; s < V; s++) order[s] = tempo_view_of_slot(s, V);
}

static float texel(const float* hm, int W, int H, int CP, int x, int y, int c) {
  if (x < 0 || x >= W || y < 0 || y >= H) return 0.0f;
  return hm[((long)y * W + x) * CP + c];
}

// One voxel, all channels: what phases A-C of unproject_kernel compute.
// hm: this frame's [V,H,W,CP]; cam: this frame's [V,32]; root != 0 applies the
// bounds test.  Returns the valid-view bitmask.
static unsigned voxel(const float* hm, const float* cam, int V, int C, int CP,
                      int H, int W, float x, float y, float z, int root,
                      float hm_w, float hm_h, float img_w, float img_h, int unf, float* out) {
  float acc[64];
  const TempoGeom geom = tempo_make_geom(hm_w, hm_h, img_w, img_h);
  for (int c = 0; c < C; c++) acc[c] = 0.0f;
  int cnt = 0;
  unsigned bits = 0;
  for (int s = 0; s < V; s++) {
    const int v = tempo_view_of_slot(s, V);
    const TempoCam* cm = (const TempoCam*)(cam + v * CAM_FLOATS);
    TempoPix p = tempo_world_to_pixel(cm, x, y, z, unf);
    float gx, gy;
    tempo_pixel_to_grid(cm, p.px, p.py, &geom, unf, &gx, &gy);
    TempoTap t = tempo_bilinear_tap(gx, gy, W, H);
    const int skip = root && !p.inside && gx == gx && gy == gy;
    cnt += p.inside;
    bits |= (unsigned)p.inside << v;
    const float* m = hm + (long)v * H * W * CP;
    for (int c = 0; c < C; c++) {
      float a = 0, b = 0, cc = 0, d = 0;
      if (!skip) {
        a = texel(m, W, H, CP, t.x0, t.y0, c);
        b = texel(m, W, H, CP, t.x0 + 1, t.y0, c);
        cc = texel(m, W, H, CP, t.x0, t.y0 + 1, c);
        d = texel(m, W, H, CP, t.x0 + 1, t.y0 + 1, c);
      }
      acc[c] = acc[c] + tempo_bilinear_mix(a, b, cc, d, t.nw, t.ne, t.sw, t.se);
    }
  }
  const float den = (root ? (float)cnt : (float)V) + 1e-6f;
  const double rden = tempo_rcp64_newton(den, tempo_rcp_est(den));
  for (int c = 0; c < C; c++) {
    float q = tempo_div_r64(acc[c], rden);
    if (tempo_is_denormal(q)) q = acc[c] / den;  // as normalise_tile
    out[c] = tempo_nan_clamp(q, 0.0f, 1.0f);
  }
  return bits;
}

// hm [B,V,H,W,CP], cam [B,V,32], axes shared [X],[Y],[Z] -> cubes [B,C,N], mask [B,N]
void emul_root(const float* hm, const float* cam, const float* ax, const float* ay,
               const float* az, const uint8_t* live, int B, int V, int C, int CP, int H,
               int W, int X, int Y, int Z, float hm_w, float hm_h, float img_w,
               float img_h, int unf, float* cubes, uint32_t* mask) {
  const long N = (long)X * Y * Z;
  float out[64];
  for (int b = 0; b < B; b++)
    for (long n = 0; n < N; n++) {
      const int iz = n % Z, iy = (n / Z) % Y, ix = n / ((long)Z * Y);
      unsigned bits = 0;
      if (live && !live[b]) {
        for (int c = 0; c < C; c++) out[c] = 0.0f;
      } else {
        bits = voxel(hm + (long)b * V * H * W * CP, cam + (long)b * V * CAM_FLOATS, V, C,
                     CP, H, W, ax[ix], ay[iy], az[iz], 1, hm_w, hm_h, img_w, img_h, unf, out);
      }
      for (int c = 0; c < C; c++) cubes[((long)b * C + c) * N + n] = out[c];
      mask[(long)b * N + n] = bits;
    }
}

// window variant: fine axes, tl/start/end [NW,3], one frame per window
void emul_window(const float* hm, const float* cam, const float* fx, const float* fy,
                 const float* fz, const int* tl, const int* st, const int* en,
                 const uint8_t* live, int NW, int V, int C, int CP, int H, int W, int VX,
                 int VY, int VZ, float hm_w, float hm_h, float img_w, float img_h, int unf,
                 float* cubes) {
  const long N = (long)VX * VY * VZ;
  float out[64];
  for (int b = 0; b < NW; b++)
    for (long n = 0; n < N; n++) {
      const int k = n % VZ, j = (n / VZ) % VY, i = n / ((long)VZ * VY);
      const int f0 = tl[3 * b] + i, f1 = tl[3 * b + 1] + j, f2 = tl[3 * b + 2] + k;
      const int act = (!live || live[b]) && f0 >= st[3 * b] && f0 < en[3 * b] &&
                      f1 >= st[3 * b + 1] && f1 < en[3 * b + 1] &&
                      f2 >= st[3 * b + 2] && f2 < en[3 * b + 2];
      if (act)
        voxel(hm + (long)b * V * H * W * CP, cam + (long)b * V * CAM_FLOATS, V, C, CP, H, W,
              fx[f0], fy[f1], fz[f2], 0, hm_w, hm_h, img_w, img_h, unf, out);
      else
        for (int c = 0; c < C; c++) out[c] = 0.0f;
      for (int c = 0; c < C; c++) cubes[((long)b * C + c) * N + n] = out[c];
    }
}

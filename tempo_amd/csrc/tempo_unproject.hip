// MI355X (gfx950) kernels + C ABI for TEMPO's heatmap -> voxel unprojection.
//
// Reference behaviour replaced (paths relative to the reference's root):
//   tempo/models/project_layer.py:65-152   ProjectLayer.get_voxel        (MODE_ROOT)
//   tempo/models/project_layer.py:225-378  ProjectLayerWithMask.get_voxel (MODE_WINDOW)
// See include/tempo_unproject.h for the ABI and DESIGN.md for the layout and
// the roofline reasoning.  No MFMA: the path is a gather + streaming store.
//
// Kernel structure.  One wave = 64 consecutive voxels of one batch item / window
// and is autonomous (no workgroup barrier; a 256-thread workgroup is four such
// waves sharing an LDS allocation).  Per view, in torch's summation order:
//   project   lane = voxel.  The camera record comes in through scalar loads
//             (wave-uniform); the lane projects, distorts, bounds-tests and turns
//             the sample point into 4 texel byte-offsets + 4 bilinear weights,
//             parked in LDS (double-buffered over views).  Count and bitmap of
//             the views that pass the bounds test stay in this layout.
//   fetch     L lanes per voxel (lane cq owns channels 4cq..4cq+3), 64/L voxels
//             per sub-step: one 16-byte buffer_load per texel, so L lanes read
//             one texel's CP*4 contiguous bytes; texels outside the map and
//             lanes that do not need the view carry an out-of-range offset,
//             which the buffer hardware answers with 0 and no memory traffic.
//   mix       bilinear mix (packed fp32) and view sum in registers.
//   The loop is software-pipelined inside the wave: fetches of view s fly under
//   the projection of view s+1.
// Then: normalise (exact division by count + 1e-6), clamp, transpose through
// LDS, store 256 contiguous bytes per channel plane and instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "../../include/tempo_unproject.h"
#include "unproject_math.h"

namespace {

#ifndef TEMPO_SB_ROOT
#define TEMPO_SB_ROOT 8
#endif
#ifndef TEMPO_SB_WINDOW
#define TEMPO_SB_WINDOW 1
#endif
// Cache policy of the volume stores: 2 = `nt` (streaming).  The volume is written once and
// read by nobody in this launch; with the default policy its lines displace the heatmaps
// from L2 and the Infinity Cache (config 2, maps read in place: 0.267 -> 0.240 ms; repacked
// in 16-frame chunks 0.262 -> 0.211 ms; sc0 / sc1 change nothing)
#ifndef TEMPO_STORE_AUX
#define TEMPO_STORE_AUX 2
#endif
constexpr int kThreads = 256;
#ifndef TEMPO_TILE_WAVES
#define TEMPO_TILE_WAVES 4
#endif
constexpr int kTileWaves = TEMPO_TILE_WAVES;  // waves per workgroup of the tile kernel
constexpr int kWaveVox = 64;                // voxels per wave
// LDS bytes per wave: 2 x 64 records of 32 B while gathering, then the [CP][65]
// output tile
// (measurement build -DTEMPO_TIMELINE: + 32 time stamps per wave, written to the `grids`
// buffer in place of the grids: tools/timeline.py)
#ifdef TEMPO_TIMELINE
#define TEMPO_TL(k)                                                        \
  do {                                                                     \
    const unsigned long long tl_now = __builtin_readcyclecounter();        \
    if (lane == 0) reinterpret_cast<unsigned long long*>(wbase + tl_off)[k] = tl_now; \
  } while (0)
constexpr int kTimelineBytes = 256;
#else
#define TEMPO_TL(k) do {} while (0)
constexpr int kTimelineBytes = 0;
#endif
constexpr int wave_lds_bytes(int L) {
  return (4 * L * 65 * 4 > 4096 ? 4 * L * 65 * 4 : 4096) + kTimelineBytes;
}
// clamp-mask words per 64-voxel tile: one ballot per (sub-step, channel of the quad)
constexpr int mask_words(int L) { return ((kWaveVox + 64 / L - 1) / (64 / L)) * 4; }
constexpr unsigned kInvalid = 0x80000000u;  // >= any view's byte size (host-checked)
constexpr int kBrickEdge = 4;  // voxels per edge of a brick (window brick kernels)
constexpr int MODE_ROOT = 0;
constexpr int MODE_WINDOW = 1;
// the root grid's arithmetic on K z-columns per item picked by a cell index (the detector's
// second stage, tempo_detector.py:204-206): voxel n = k * Z + z of the virtual [1, K, Z] grid
constexpr int MODE_COLUMNS = 2;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef tempo_f2 f32x2;
#define TEMPO_AS4 __attribute__((address_space(4)))

struct UnprojArgs {
  const float* hm;
  const float* cam;
  const float* ax[3];
  const uint8_t* live;
  const int32_t* win_frame;
  const int32_t* tl;
  const int32_t* start;
  const int32_t* end;
  float* cubes;
  float* grids;
  void* mask;
  long long ax_stride[3];  // root: floats between batch items (0 = shared axes)
  long long frame_stride;  // floats per frame of hm = V*H*W*CP
  unsigned view_bytes;     // H*W*CP*4
  int NB;                  // batch items (root) / windows
  int V, C, CP, H, W;
  int Cs;  // channels per item of the OUTPUT tensors (== C, or the full count when this
           // launch computes one 16-channel group of a 32-channel call: forward kernels only)
  int dim[3];   // output voxels per axis
  int fdim[3];  // window: extent of the fine axes
  int F;        // frames in hm / cam
  int N;        // dim[0]*dim[1]*dim[2]
  int tiles;    // tiles (one per workgroup) per batch item
  int tile_vox; // voxels per tile = wave_vox * waves per workgroup
  int wave_vox; // voxels per wave: 64, or whole z-columns only (root z-max: floor(64 / Z) * Z)
  int total;    // NB * tiles
  int chunk;    // ceil(total / 8): contiguous run of work items per XCD label
  int mask_bytes;
  TempoGeom geom;
  // axis-max planes (fused epilogue), or nullptr: [NB][C][d0][d1], [NB][C][d0][d2],
  // [NB][C][d1][d2] as unsigned bit patterns of non-negative floats, zero-initialised
  unsigned* pl01;
  unsigned* pl02;
  unsigned* pl12;
  // training: the forward records which elements the final clamp lets a gradient
  // through (unclamped value in [0,1]); ballot words in the kernels' own lane
  // layout, [NB][ceil(N/64)][mask_words(L)]
  unsigned long long* cmask;
  float* ghm;  // backward: gradient of the channel-last heatmaps, accumulated
  const int32_t* groups;  // grouped launch: [NG][2] = (first item, count), or nullptr
  int NG;
  int wave_lds;  // grouped launch: LDS bytes per wave
  int slot0, slot1;  // root / per-sample backward: slots [slot0, slot1) are scattered by this launch
  unsigned order[10];  // slot -> view in torch's summation order (one byte per slot, 255 = empty)
  unsigned magic[2];   // floor(2^32 / dim[2]), floor(2^32 / dim[1]): voxel index -> (i0, i1, i2)
  // Heatmap layout of the forward kernels: per view a base pointer (frame 0's map), frames
  // frame_bytes apart, texels texel_bytes apart (channels contiguous), rows row_bytes apart.
  // The repacked [F][V][H][W][CP] buffer and V separate channel-last [F][H][W][C] tensors
  // (torch.channels_last) are both of this form.
  const char* vbase[TEMPO_MAX_VIEWS];
  long long frame_bytes;
  unsigned map_bytes;  // the buffer range of one map (bytes the fetches may touch)
  int texel_bytes, row_bytes;
  // MODE_COLUMNS: cell index [NB][K] (cell = x * Y + y of the fdim[0] x fdim[1] root grid),
  // output [NB][K][C][dim[2]]; plane_bytes != 0: the maps are NCHW (channel c of a texel lies
  // c * plane_bytes further on), read with one 4-byte load per channel
  const long long* col_index;
  float* col_out;
  int K;
  unsigned plane_bytes;
  // Trainable axis-max planes (window): per plane entry the CLAIM of the voxel torch.max
  // routes the entry's gradient to = 2 * (index along the reduced axis of the FIRST voxel
  // that attains the maximum) + 1 if the clamp blocks that voxel's gradient; laid out like
  // the planes (pl01 / pl02 / pl12).  window_claim_kernel writes them (atomic minimum, from
  // 0x7f7f7f7f), the backward reads them together with the planes' incoming gradients.
  int32_t* claim[3];
  const float* gplane[3];
};

__device__ __forceinline__ int slot_view(const UnprojArgs& a, int slot) {
  return (int)((a.order[slot >> 2] >> ((slot & 3) * 8)) & 255u);
}

// host side: slot -> view table and the voxel-index reciprocals of a launch
void fill_order_magic(unsigned* order, unsigned* magic, int V, const int* dim) {
  for (int i = 0; i < 2; i++) {
    const unsigned long long m = 0x100000000ULL / (unsigned)dim[2 - i];
    magic[i] = m > 0xffffffffULL ? 0xffffffffu : (unsigned)m;
  }
  for (int w = 0; w < 10; w++) order[w] = 0xffffffffu;
  for (int sl = 0; sl < V; sl++) {
    const unsigned v = (unsigned)tempo_view_of_slot(sl, V);
    order[sl >> 2] = (order[sl >> 2] & ~(255u << ((sl & 3) * 8))) | (v << ((sl & 3) * 8));
  }
}

// buffer descriptor of one (frame, view) map, from provably wave-uniform words (or hipcc
// wraps every fetch in a waterfall loop); `real` = false gives a zero-sized buffer, whose
// loads return 0 without memory traffic
__device__ __forceinline__ __amdgpu_buffer_rsrc_t map_rsrc(const UnprojArgs& a, int frame,
                                                           int view, bool real) {
  const unsigned long long vb = (unsigned long long)(uintptr_t)(
      a.vbase[real ? view : 0] + (real ? (long long)frame * a.frame_bytes : 0ll));
  const unsigned vb_lo = __builtin_amdgcn_readfirstlane((unsigned)vb);
  const unsigned vb_hi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>((uintptr_t)(((unsigned long long)vb_hi << 32) | vb_lo)), 0,
      __builtin_amdgcn_readfirstlane(real ? (int)a.map_bytes : 0), 0x00020000);
}

// wave-uniform read through the scalar data cache (constant address space)
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p, long long i) {
  const T TEMPO_AS4* q = (const T TEMPO_AS4*)(uintptr_t)(p + i);
  return *q;
}

__device__ __forceinline__ int texel_index(int y0, int W, int x0) {
  return __mul24(y0, W) + x0;
}

// byte offset of texel (x0, y0) in a map with the given pitches (may be negative:
// a texel outside the map, which the buffer range check then answers with 0)
__device__ __forceinline__ int texel_offset(int y0, int x0, int row_bytes, int texel_bytes) {
  return __mul24(y0, row_bytes) + __mul24(x0, texel_bytes);
}

// one byte, wave-uniform: the scalar unit has no byte loads (a byte read would
// go through the vector memory pipe and its wait would drain the texel fetches
// in flight), so read the aligned word that holds it
__device__ __forceinline__ unsigned uniform_load_u8(const uint8_t* p, long long i) {
  const uintptr_t addr = (uintptr_t)(p + i);
  const unsigned TEMPO_AS4* q = (const unsigned TEMPO_AS4*)(addr & ~(uintptr_t)3);
  return (*q >> (unsigned)((addr & 3) * 8)) & 255u;
}

__device__ __forceinline__ TempoCam load_cam(const float* cam, int rec) {
  // 64-bit multiplies run on the VALU and lose the compiler's uniformity proof
  // (the record would then come in through vector loads): 32-bit index, pinned
  const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(rec) * (CAM_FLOATS * 4u);
  const float TEMPO_AS4* q =
      (const float TEMPO_AS4*)((uintptr_t)cam + (unsigned long long)off);
  TempoCam c;
  float* d = reinterpret_cast<float*>(&c);
#pragma unroll
  for (int i = 0; i < CAM_FLOATS; i++) d[i] = q[i];
  return c;
}

// maximum over the wave of non-negative-float bit patterns (0 = identity);
// the result is valid in lane 63
__device__ __forceinline__ unsigned wave_umax_to_lane63(unsigned v) {
  auto step = [&](auto ctrl, auto rmask) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(
        0, (int)v, decltype(ctrl)::value, decltype(rmask)::value, 0xf, true);
    v = o > v ? o : v;
  };
  using std::integral_constant;
  step(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});  // row_shr:1
  step(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});  // row_shr:2
  step(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});  // row_shr:4
  step(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});  // row_shr:8
  step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15
  step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31
  return v;
}

__device__ __forceinline__ void voxel_index(const UnprojArgs& a, int n, int* i0, int* i1,
                                            int* i2) {
  unsigned t = __umulhi((unsigned)n, a.magic[0]);
  unsigned r2 = (unsigned)n - t * (unsigned)a.dim[2];
  if (r2 >= (unsigned)a.dim[2]) { r2 -= (unsigned)a.dim[2]; t++; }
  unsigned q0 = __umulhi(t, a.magic[1]);
  unsigned r1 = t - q0 * (unsigned)a.dim[1];
  if (r1 >= (unsigned)a.dim[1]) { r1 -= (unsigned)a.dim[1]; q0++; }
  *i2 = (int)r2;
  *i1 = (int)r1;
  *i0 = (int)q0;
}

// Root z-max alone: voxels per wave when waves are made of whole z-columns (every column's
// maximum is then complete inside one wave and is stored, not max-accumulated), or 64 when
// columns do not fit a wave or more than 1/8 of the lanes would idle.
__host__ __device__ inline int zmax_wave_vox(int Z) {
  if (Z >= kWaveVox) return kWaveVox;
  const int w = (kWaveVox / Z) * Z;
  return (kWaveVox - w) * 8 <= kWaveVox ? w : kWaveVox;
}

// z-max planes of a dead item in whole-column mode (nothing else clears them)
__device__ __forceinline__ void zmax_clear_columns(const UnprojArgs& a, int b, int n0,
                                                   int lane) {
  const int Z = a.dim[2];
  const int cols = a.wave_vox / Z;
  const int q0 = n0 / Z, ncol = a.dim[0] * a.dim[1];
  for (int s0 = 0; s0 < cols * a.C; s0 += kWaveVox) {
    const int slot = s0 + lane;
    const int c = slot / a.C, ch = slot - c * a.C;
    if (c < cols && q0 + c < ncol)
      a.pl01[((long long)b * a.Cs + ch) * ncol + q0 + c] = 0u;
  }
}

// Fused axis-max epilogue: the finished [C][64 voxels] tile of one wave (s_out,
// lane = voxel) goes into the three max-projection planes the pose regressor
// builds from the window volume (tempo_pose.py:172-176; center_net.py:230 for
// the root grid's z-max).  Values are clamped to [0,1] (or NaN), so their bit
// patterns order like unsigned integers and NaN (larger than 1.0f) wins, as in
// torch.max.  Along the fastest axis the wave reduces in registers when it
// holds exactly one column; everything else is a no-return atomic max.
template <int CH>  // channel slots of the tile (4 L)
__device__ __forceinline__ void planes_epilogue(const UnprojArgs& a, int b, int n0, int lane,
                                                const float* s_out, int tstr) {
  const int n = n0 + lane;
  const bool on = lane < a.wave_vox && n < a.N;
  int i0 = 0, i1 = 0, i2 = 0;
  if (on) voxel_index(a, n, &i0, &i1, &i2);
  const bool column = a.dim[2] == kWaveVox;  // n0 is a multiple of 64: one whole column
  // waves made of whole columns: every column's maximum is complete in this wave, so its
  // first lane stores it (no atomic, no zero-initialised plane)
  const bool whole = !column && a.wave_vox % a.dim[2] == 0;
  if (whole && a.pl01 != nullptr && a.pl02 == nullptr && a.pl12 == nullptr) {
    // lane = (column of the wave, channel slot): the column's Z values are read straight out
    // of the tile (rows of different channels and columns fall into different banks) and the
    // maximum is stored -- no cross-lane traffic at all.  n0 is a multiple of Z here, so the
    // column's place in the [X][Y] plane is n0 / Z + column: no index arithmetic per lane.
    const int Z = a.dim[2];
    const int cols = a.wave_vox / Z;
    const int q0 = __builtin_amdgcn_readfirstlane(n0 / Z);
    const int ncol = a.dim[0] * a.dim[1];
    for (int c0 = 0; c0 < cols; c0 += kWaveVox / CH) {
      const int c = c0 + lane / CH, ch = lane % CH;
      if (c < cols && ch < a.C && q0 + c < ncol) {
        const float* col = s_out + ch * tstr + c * Z;
        unsigned m = 0u;
#pragma unroll 4
        for (int k = 0; k < Z; k++) {
          unsigned v = __float_as_uint(col[k]);
          v = v == 0x80000000u ? 0u : v;  // -0.0 orders as +0.0
          m = v > m ? v : m;
        }
        a.pl01[((long long)b * a.Cs + ch) * ncol + q0 + c] = m;
      }
    }
    return;
  }
  const long long bc = (long long)b * a.Cs;
  // shorter columns (the root grid's 20 planes): the lanes of one column are
  // consecutive, so a segmented shift-and-max leaves the column's maximum in its
  // first lane and only that lane issues the atomic (a column may continue in
  // the next wave)
  const int col = on ? n - i2 : -1 - lane;
  bool same[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    // (the shuffle runs on every lane: a cross-lane read of an inactive lane gives 0)
    const int other = __shfl_down(col, 1 << k);
    same[k] = lane + (1 << k) < kWaveVox && other == col;
  }
  const int before = __shfl_up(col, 1);
  const bool head = lane == 0 || before != col;
  for (int ch = 0; ch < a.C; ch++) {
    unsigned v = on ? __float_as_uint(s_out[ch * tstr + lane]) : 0u;
    v = v == 0x80000000u ? 0u : v;  // -0.0 (a clamped negative underflow) orders as +0.0
    if (v != 0u) {
      if (a.pl02 != nullptr)
        __hip_atomic_fetch_max(a.pl02 + ((bc + ch) * a.dim[0] + i0) * a.dim[2] + i2, v,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.pl12 != nullptr)
        __hip_atomic_fetch_max(a.pl12 + ((bc + ch) * a.dim[1] + i1) * a.dim[2] + i2, v,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.pl01 != nullptr) {
      unsigned* dst = a.pl01 + ((bc + ch) * a.dim[0] + i0) * a.dim[1] + i1;
      if (column) {
        const unsigned m = wave_umax_to_lane63(v);
        if (lane == 63) *dst = m;
      } else {
        unsigned m = v;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if ((1 << k) >= a.dim[2]) break;  // no column is longer than dim[2] lanes
          const unsigned o = __shfl_down(m, 1 << k);
          if (same[k]) m = o > m ? o : m;
        }
        if (whole) {
          if (head && on) *dst = m;
        } else if (head && m != 0u) {
          __hip_atomic_fetch_max(dst, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

// Masked mean of one wave's tile: every channel value divided by its voxel's denominator
// (count + 1e-6) -- exact fp32 division through one double multiply (tempo_div_r64);
// results in the denormal range, where that form can miss an exact tie, are redone with
// the plain division.  Returns true (wave-uniform) when a NaN or such a result is present
// anywhere in the wave: the caller then clamps NaN-aware.
// (One case is not tested for: a product that rounds to zero while the true quotient rounds
// to the smallest denormal, i.e. a quotient within 2^-50, relatively, of 2^-150.  With a
// denormal numerator k * 2^-149 that needs a denominator within 2^-50 of the even integer 2 k,
// and the denominators here are count + 1e-6 rounded to fp32, at least 9.5e-7 above an
// integer: it cannot occur.  tools/verify_constdiv.c says the same for its divisors.)
struct DenRec {
  double r;   // 1 / den
  float den;  // (float)count + 1e-6f
  float pad;
};
__device__ __forceinline__ DenRec make_den(float den) {
  DenRec d;
  d.r = tempo_rcp64_newton(den, tempo_rcp_est(den));
  d.den = den;
  d.pad = 0.0f;
  return d;
}
template <int NSUB>
__device__ __forceinline__ bool normalise_tile(f32x2 (&acc01)[NSUB], f32x2 (&acc23)[NSUB],
                                               const DenRec (&dr)[NSUB]) {
  f32x2 q01[NSUB], q23[NSUB];
  bool odd = false;
  constexpr int kOdd = 0x001 | 0x002 | 0x010 | 0x080;  // sNaN, qNaN, -denormal, +denormal
#pragma unroll
  for (int s = 0; s < NSUB; s++) {
    q01[s] = tempo_div_r64_2(acc01[s], dr[s].r, dr[s].r);
    q23[s] = tempo_div_r64_2(acc23[s], dr[s].r, dr[s].r);
    odd |= __builtin_amdgcn_classf(q01[s].x, kOdd) | __builtin_amdgcn_classf(q01[s].y, kOdd) |
           __builtin_amdgcn_classf(q23[s].x, kOdd) | __builtin_amdgcn_classf(q23[s].y, kOdd);
  }
  const bool slow = __builtin_amdgcn_ballot_w64(odd) != 0;
  if (slow) {
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      if (tempo_is_denormal(q01[s].x)) q01[s].x = acc01[s].x / dr[s].den;
      if (tempo_is_denormal(q01[s].y)) q01[s].y = acc01[s].y / dr[s].den;
      if (tempo_is_denormal(q23[s].x)) q23[s].x = acc23[s].x / dr[s].den;
      if (tempo_is_denormal(q23[s].y)) q23[s].y = acc23[s].y / dr[s].den;
    }
  }
#pragma unroll
  for (int s = 0; s < NSUB; s++) {
    acc01[s] = q01[s];
    acc23[s] = q23[s];
  }
  return slow;
}

template <int MODE>
__device__ __forceinline__ bool voxel_world(const UnprojArgs& a, int b, int n,
                                            float* x, float* y, float* z) {
  // n = (i0 * dim[1] + i1) * dim[2] + i2.  Division by a host-known divisor:
  // the multiply-high estimate is q or q - 1, one correction step makes it exact
  unsigned t = __umulhi((unsigned)n, a.magic[0]);
  unsigned r2 = (unsigned)n - t * (unsigned)a.dim[2];
  if (r2 >= (unsigned)a.dim[2]) { r2 -= (unsigned)a.dim[2]; t++; }
  unsigned q0 = __umulhi(t, a.magic[1]);
  unsigned r1 = t - q0 * (unsigned)a.dim[1];
  if (r1 >= (unsigned)a.dim[1]) { r1 -= (unsigned)a.dim[1]; q0++; }
  const int i2 = (int)r2, i1 = (int)r1, i0 = (int)q0;
  if (MODE == MODE_ROOT) {
    *x = a.ax[0][b * a.ax_stride[0] + i0];
    *y = a.ax[1][b * a.ax_stride[1] + i1];
    *z = a.ax[2][b * a.ax_stride[2] + i2];
    return true;
  } else if (MODE == MODE_COLUMNS) {
    // column i1 of item b stands at the grid cell its index names; a cell outside the grid
    // gives a column of zeros (as tempo_gather_zcolumns does)
    const long long cell = a.col_index[(long long)b * a.K + i1];
    if (cell < 0 || cell >= (long long)a.fdim[0] * a.fdim[1]) return false;
    const int cx = (int)(cell / a.fdim[1]);
    const int cy = (int)(cell - (long long)cx * a.fdim[1]);
    *x = a.ax[0][b * a.ax_stride[0] + cx];
    *y = a.ax[1][b * a.ax_stride[1] + cy];
    *z = a.ax[2][b * a.ax_stride[2] + i2];
    return true;
  } else {
    const int f0 = uniform_load(a.tl, 3 * b + 0) + i0;
    const int f1 = uniform_load(a.tl, 3 * b + 1) + i1;
    const int f2 = uniform_load(a.tl, 3 * b + 2) + i2;
    const bool act =
        f0 >= uniform_load(a.start, 3 * b + 0) && f0 < uniform_load(a.end, 3 * b + 0) &&
        f1 >= uniform_load(a.start, 3 * b + 1) && f1 < uniform_load(a.end, 3 * b + 1) &&
        f2 >= uniform_load(a.start, 3 * b + 2) && f2 < uniform_load(a.end, 3 * b + 2) &&
        (unsigned)f0 < (unsigned)a.fdim[0] &&  // never read past an axis
        (unsigned)f1 < (unsigned)a.fdim[1] && (unsigned)f2 < (unsigned)a.fdim[2];
    if (!act) return false;
    *x = a.ax[0][f0];
    *y = a.ax[1][f1];
    *z = a.ax[2][f2];
    return true;
  }
}

// One wave's 64 voxels [n0, n0 + 64) of batch item / window b, all views: leaves
// the finished tile (normalised, clamped) in the wave's LDS, [4L channels][65]
// floats at wbase, lane = voxel; writes the optional grids / mask outputs.
// KEEP: also hand the finished values back in the gather layout -- keep[s * 4 + j] = bit
// pattern of channel 4 cq + j of voxel s * VPS + vq (lane = vq * L + cq), clamped, -0.0 as
// +0.0, so that the patterns order like the values with NaN on top (0 for slots that hold no
// voxel) -- and write the LDS tile only if the volume is asked for (a.cubes).
// KEEP == 2: the LDS tile holds the mean BEFORE the clamp (window_claim_kernel applies the
// clamp itself, because it also needs to know where the clamp blocks the gradient).
template <int MODE, int L, int SB, int UNF, int KEEP = 0>
__device__ __forceinline__ void unproject_wave(const UnprojArgs& a, int b, int frame, int n0,
                                               int lane, unsigned char* wbase,
                                               unsigned* keep = nullptr) {
  constexpr int VPS = 64 / L;                       // voxels per sub-step of phase B
  constexpr int NSUB = (kWaveVox + VPS - 1) / VPS;  // sub-steps per wave
  // wave-private LDS: 2 x 64 records of (4 texel offsets, 4 weights): the view
  // being gathered and the view being projected
  u32x4* s_off = reinterpret_cast<u32x4*>(wbase);
  float4* s_wt = reinterpret_cast<float4*>(wbase + (size_t)kWaveVox * 32);
  const int n = n0 + lane;  // phase A: this lane's voxel
#ifdef TEMPO_TIMELINE
  constexpr int tl_off = wave_lds_bytes(L) - kTimelineBytes;
  TEMPO_TL(0);
#endif

  float x = 0.f, y = 0.f, z = 0.f;
  const bool have = lane < a.wave_vox && n < a.N && voxel_world<MODE>(a, b, n, &x, &y, &z);
#ifndef TEMPO_TIMELINE
  if (MODE == MODE_ROOT && a.grids != nullptr && have) {
    float* g = a.grids + ((long long)b * a.N + n) * 3;
    g[0] = x;
    g[1] = y;
    g[2] = z;
  }
#endif

  // The coordinates came in through vector loads.  Launder them through an empty
  // asm so the waitcnt pass sees them as landed: otherwise it re-waits for
  // "possibly pending" x/y/z at the top of every phase A with vmcnt(0), which
  // also drains the texel fetches that phase A is meant to run under.
  asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
  TEMPO_TL(1);

  // phase B lane roles: L lanes per voxel, VPS voxels per sub-step
  const bool lane_on = lane < VPS * L;
  const int vq = lane / L;
  const int cq = lane - vq * L;
  const unsigned lo = (unsigned)cq * 16u;
  f32x2 acc01[NSUB], acc23[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; s++) {
    acc01[s] = tempo_splat(0.f);
    acc23[s] = tempo_splat(0.f);
  }
  // views that pass the bounds test, for this lane's own voxel (phase A layout)
  int cnt = 0;
  unsigned mbits = 0;

  // phase A of one view: project this lane's voxel, park (4 offsets, 4 weights)
  // in LDS buffer `buf`.  Returns false when no lane of the wave needs the view
  // (nothing written then).
  auto phase_a = [&](int slot, int buf) -> bool {
    const int view = slot_view(a, slot);
    const TempoCam cam = load_cam(a.cam, frame * a.V + view);
    TempoPix p;
    p.px = p.py = 0.f;
    p.inside = 0;
    bool need = false;
    if (have) {
      p = tempo_world_to_pixel(&cam, x, y, z, UNF);
      // root: a view that fails the bounds test is multiplied by 0 in the
      // reference, so its texels are never fetched -- unless a coordinate is
      // NaN: then the reference's 0 * NaN poisons the voxel and so must we
      need = MODE == MODE_WINDOW || p.inside || p.px != p.px || p.py != p.py;
    }
    cnt += p.inside;
    mbits |= (unsigned)p.inside << view;
    const bool active = __builtin_amdgcn_ballot_w64(need) != 0;
    if (active) {
      float gx, gy;
      tempo_pixel_to_grid(&cam, p.px, p.py, &a.geom, UNF, &gx, &gy);
      const TempoTap t = tempo_bilinear_tap(gx, gy, a.W, a.H);
      // rows outside the map need no test: their byte offset is negative (wraps
      // past 2^31) or >= view_bytes, and the buffer load returns 0 for it
      // a lane that does not need the view fetches nothing either; its weights
      // are finite, so its (all-zero) texels mix to 0
      bool x0ok = need && (unsigned)t.x0 < (unsigned)a.W;
      bool x1ok = need && (unsigned)(t.x0 + 1) < (unsigned)a.W;
      // (columns on planar maps: a row outside the map would land in a neighbouring
      // channel's plane instead of outside the buffer, so the rows are tested as well)
      bool r0ok = true, r1ok = true;
      if (MODE == MODE_COLUMNS) {
        r0ok = (unsigned)t.y0 < (unsigned)a.H;
        r1ok = (unsigned)(t.y0 + 1) < (unsigned)a.H;
      }
      // 24-bit multiply-adds (full rate; y0, x0, the row pitch and the texel pitch are
      // far below 2^23: host-checked)
      const unsigned base = (unsigned)texel_offset(t.y0, t.x0, a.row_bytes, a.texel_bytes);
      const unsigned row = (unsigned)a.row_bytes, tb = (unsigned)a.texel_bytes;
      u32x4 off;
      off.x = x0ok && r0ok ? base : kInvalid;
      off.y = x1ok && r0ok ? base + tb : kInvalid;
      off.z = x0ok && r1ok ? base + row : kInvalid;
      off.w = x1ok && r1ok ? base + row + tb : kInvalid;
      const float4 wt = make_float4(t.nw, t.ne, t.sw, t.se);
#if defined(TEMPO_ABLATE) && TEMPO_ABLATE == 1  // no gather traffic
      off.x = off.y = off.z = off.w = kInvalid;
#endif
#if defined(TEMPO_ABLATE) && TEMPO_ABLATE == 4  // same fetch count, all L1 hits
      if (off.x != kInvalid) off.x = (unsigned)(lane & 15) * 64u;
      if (off.y != kInvalid) off.y = (unsigned)(lane & 15) * 64u + 1024u;
      if (off.z != kInvalid) off.z = (unsigned)(lane & 15) * 64u + 2048u;
      if (off.w != kInvalid) off.w = (unsigned)(lane & 15) * 64u + 3072u;
#endif
      s_off[buf * kWaveVox + lane] = off;
      s_wt[buf * kWaveVox + lane] = wt;
    }
    return active;
  };

  // views in torch's summation order, software-pipelined inside the wave:
  //   issue the texel fetches of view s  ->  phase A of view s+1 (pure VALU,
  //   runs while the fetches fly)  ->  mix view s.
  bool act_cur = phase_a(0, 0), act_nxt = false;
  TEMPO_TL(2);
#pragma unroll 1
  for (int slot = 0; slot < a.V; slot++) {
    const int buf = slot & 1;
    const int view = slot_view(a, slot);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // SB sub-steps' fetches (4 * SB registers-quads) are in flight together; the
    // first group flies under the projection of the next view
    constexpr int G = SB < NSUB ? SB : NSUB;
    u32x4 tex[G][4];
    __amdgpu_buffer_rsrc_t rs;
    auto fetch = [&](int s0) {
#pragma unroll
      for (int g = 0; g < G; g++) {
        const int s = s0 + g;
        // (columns: short waves, sub-steps past the wave's voxels are skipped -- wave-uniform)
        if (s < NSUB && (MODE != MODE_COLUMNS || s * VPS < a.wave_vox)) {
          const int vsl = s * VPS + vq;
          const int rd = (lane_on && vsl < kWaveVox) ? vsl : 0;
          const u32x4 o = s_off[buf * kWaveVox + rd];
          if (MODE == MODE_COLUMNS && a.plane_bytes != 0) {
            // NCHW maps: channel 4 cq + j of a texel lies (4 cq + j) planes further on; a
            // channel past the slice is past the buffer's range and reads as 0
            const unsigned ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
              for (int j = 0; j < 4; j++)
                tex[g][t][j] = __builtin_amdgcn_raw_buffer_load_b32(
                    rs, ov[t] + (unsigned)(4 * cq + j) * a.plane_bytes, 0, 0);
          } else {
            tex[g][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, o.x + lo, 0, 0);
            tex[g][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, o.y + lo, 0, 0);
            tex[g][2] = __builtin_amdgcn_raw_buffer_load_b128(rs, o.z + lo, 0, 0);
            tex[g][3] = __builtin_amdgcn_raw_buffer_load_b128(rs, o.w + lo, 0, 0);
          }
        }
      }
    };
    auto mix = [&](int s0) {
#pragma unroll
      for (int g = 0; g < G; g++) {
        const int s = s0 + g;
        if (s < NSUB && (MODE != MODE_COLUMNS || s * VPS < a.wave_vox)) {
          const int vsl = s * VPS + vq;
          const int rd = (lane_on && vsl < kWaveVox) ? vsl : 0;
          const float4 w = s_wt[buf * kWaveVox + rd];
          // channels 4cq..4cq+3 as two packed pairs; same chain as
          // tempo_bilinear_mix, then the view sum
          const f32x2 a01 = {__uint_as_float(tex[g][0][0]), __uint_as_float(tex[g][0][1])};
          const f32x2 a23 = {__uint_as_float(tex[g][0][2]), __uint_as_float(tex[g][0][3])};
          const f32x2 b01 = {__uint_as_float(tex[g][1][0]), __uint_as_float(tex[g][1][1])};
          const f32x2 b23 = {__uint_as_float(tex[g][1][2]), __uint_as_float(tex[g][1][3])};
          const f32x2 c01 = {__uint_as_float(tex[g][2][0]), __uint_as_float(tex[g][2][1])};
          const f32x2 c23 = {__uint_as_float(tex[g][2][2]), __uint_as_float(tex[g][2][3])};
          const f32x2 d01 = {__uint_as_float(tex[g][3][0]), __uint_as_float(tex[g][3][1])};
          const f32x2 d23 = {__uint_as_float(tex[g][3][2]), __uint_as_float(tex[g][3][3])};
          const f32x2 nw = tempo_splat(w.x), ne = tempo_splat(w.y);
          const f32x2 sw = tempo_splat(w.z), se = tempo_splat(w.w);
          f32x2 v01 = a01 * nw, v23 = a23 * nw;
          v01 = __builtin_elementwise_fma(b01, ne, v01);
          v23 = __builtin_elementwise_fma(b23, ne, v23);
          v01 = __builtin_elementwise_fma(c01, sw, v01);
          v23 = __builtin_elementwise_fma(c23, sw, v23);
          v01 = __builtin_elementwise_fma(d01, se, v01);
          v23 = __builtin_elementwise_fma(d23, se, v23);
          acc01[s] = acc01[s] + v01;
          acc23[s] = acc23[s] + v23;
        }
      }
    };
    if (act_cur) {
      // descriptor from provably wave-uniform words, or hipcc wraps every fetch
      // in a waterfall loop
      rs = map_rsrc(a, frame, view, true);
      fetch(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (slot < 8) TEMPO_TL(3 + 3 * slot);
    if (slot + 1 < a.V) act_nxt = phase_a(slot + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if (slot < 8) TEMPO_TL(4 + 3 * slot);
    if (act_cur) {
      mix(0);
#pragma unroll
      for (int s0 = G; s0 < NSUB; s0 += G) {
        __builtin_amdgcn_sched_barrier(0);
        fetch(s0);
        __builtin_amdgcn_sched_barrier(0);
        mix(s0);
      }
    }
    act_cur = act_nxt;
    if (slot < 8) TEMPO_TL(5 + 3 * slot);
  }
  TEMPO_TL(27);

  // ------------------------------------------------------------------ phase C
  // masked mean: sum / (count + 1e-6), clamp; transpose through the wave's LDS
  // (the records are dead) so that every store instruction writes 64 consecutive
  // voxels of one channel: 256 contiguous bytes
  constexpr int TSTR = kWaveVox + 1;
  float* s_out = reinterpret_cast<float*>(wbase);  // [4L][TSTR] floats <= 64*64 bytes for L <= 15
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  DenRec dr[NSUB];
  if (MODE == MODE_WINDOW) {
    // every voxel of a window divides by V + 1e-6 (project_layer.py:367-371): one record,
    // computed once on scalar-uniform values, instead of one per voxel handed over in LDS
    const DenRec d = make_den((float)a.V + 1e-6f);
#pragma unroll
    for (int s = 0; s < NSUB; s++) dr[s] = d;
  } else {
    // the denominator belongs to the voxel: one IEEE reciprocal per voxel in the
    // phase A layout, handed to the voxel's L channel lanes through LDS
    reinterpret_cast<DenRec*>(wbase)[lane] = make_den((float)cnt + 1e-6f);
    if (MODE == MODE_ROOT && a.mask != nullptr && lane < a.wave_vox && n < a.N) {
      char* m = (char*)a.mask + ((long long)b * a.N + n) * a.mask_bytes;
      if (a.mask_bytes == 1) *(uint8_t*)m = (uint8_t)mbits;
      else if (a.mask_bytes == 2) *(uint16_t*)m = (uint16_t)mbits;
      else *(uint32_t*)m = mbits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      const int vsl = s * VPS + vq;
      dr[s] = reinterpret_cast<const DenRec*>(wbase)[(lane_on && vsl < kWaveVox) ? vsl : 0];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // exact division; a NaN anywhere in the wave (torch.clamp keeps it) sends the whole
  // wave through the NaN-aware clamp
  const bool slow = normalise_tile<NSUB>(acc01, acc23, dr);
  if (KEEP == 2) {
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      const int vsl = s * VPS + vq;
      if (lane_on && vsl < kWaveVox) {
        float* o = s_out + (4 * cq) * TSTR + vsl;
        o[0 * TSTR] = acc01[s].x;
        o[1 * TSTR] = acc01[s].y;
        o[2 * TSTR] = acc23[s].x;
        o[3 * TSTR] = acc23[s].y;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return;
  }
  if (a.cmask != nullptr) {
    // torch.clamp's backward passes the gradient where min <= x <= max (NaN: no)
    unsigned long long* mw =
        a.cmask + ((long long)b * ((a.N + kWaveVox - 1) / kWaveVox) + (n0 >> 6)) * mask_words(L);
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      const bool live_el = lane_on && s * VPS + vq < kWaveVox;
      const float e[4] = {acc01[s].x, acc01[s].y, acc23[s].x, acc23[s].y};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned long long bal =
            __builtin_amdgcn_ballot_w64(live_el && e[j] >= 0.0f && e[j] <= 1.0f);
        if (lane == 0) mw[s * 4 + j] = bal;
      }
    }
  }
  if (KEEP) {
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      const bool live_el = lane_on && s * VPS + vq < kWaveVox;
      const float e[4] = {acc01[s].x, acc01[s].y, acc23[s].x, acc23[s].y};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        unsigned bits;
        if (!slow) {
          // clamp to [0, 1] on the bit pattern (one v_med3_i32): every negative value,
          // -0.0 included, has a negative pattern and becomes +0.0; no NaN in this wave
          int t = (int)__float_as_uint(e[j]);
          t = t < 0 ? 0 : t;
          t = t > 0x3f800000 ? 0x3f800000 : t;
          bits = (unsigned)t;
        } else {
          const float v = tempo_nan_clamp(e[j], 0.0f, 1.0f);
          bits = v == 0.0f ? 0u : __float_as_uint(v);
        }
        keep[s * 4 + j] = live_el ? bits : 0u;
      }
    }
    if (a.cubes == nullptr) return;  // (wave-uniform) nobody reads the tile
  }
#pragma unroll
  for (int s = 0; s < NSUB; s++) {
    const int vsl = s * VPS + vq;
    if (lane_on && vsl < kWaveVox) {
      float* o = s_out + (4 * cq) * TSTR + vsl;
      if (!slow) {
        o[0 * TSTR] = __builtin_amdgcn_fmed3f(acc01[s].x, 0.0f, 1.0f);
        o[1 * TSTR] = __builtin_amdgcn_fmed3f(acc01[s].y, 0.0f, 1.0f);
        o[2 * TSTR] = __builtin_amdgcn_fmed3f(acc23[s].x, 0.0f, 1.0f);
        o[3 * TSTR] = __builtin_amdgcn_fmed3f(acc23[s].y, 0.0f, 1.0f);
      } else {
        o[0 * TSTR] = tempo_nan_clamp(acc01[s].x, 0.0f, 1.0f);
        o[1 * TSTR] = tempo_nan_clamp(acc01[s].y, 0.0f, 1.0f);
        o[2 * TSTR] = tempo_nan_clamp(acc23[s].x, 0.0f, 1.0f);
        o[3 * TSTR] = tempo_nan_clamp(acc23[s].y, 0.0f, 1.0f);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  TEMPO_TL(28);
}

// One wave's finished tile -> the volume: value ch of this lane's voxel n is src[ch * tstr]
// (LDS), its place item[ch * N + n].  One buffer descriptor per batch item, the channel
// plane in the instruction's scalar offset: per channel one LDS read, one store and one
// scalar add (as plain pointer arithmetic hipcc spends 4-5 VALU instructions per store on
// 64-bit addresses).  Volumes of 4 GB and more per item keep the pointer form.
__device__ __forceinline__ void store_tile(float* item, int C, int N, int n, const float* src,
                                           int tstr) {
  const unsigned long long bytes = (unsigned long long)C * (unsigned)N * 4ull;
  if (bytes < 0x100000000ull) {
    const unsigned long long p = (unsigned long long)(uintptr_t)item;
    const unsigned p_lo = __builtin_amdgcn_readfirstlane((unsigned)p);
    const unsigned p_hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>((uintptr_t)(((unsigned long long)p_hi << 32) | p_lo)), 0,
        __builtin_amdgcn_readfirstlane((int)(unsigned)bytes), 0x00020000);
    const unsigned plane = (unsigned)N * 4u;
    unsigned so = 0;
    for (int ch = 0; ch < C; ch++, so += plane)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(src[ch * tstr]), rs,
                                            (unsigned)n * 4u, so, TEMPO_STORE_AUX);
  } else {
    for (int ch = 0; ch < C; ch++) item[(long long)ch * N + n] = src[ch * tstr];
  }
}

// One workgroup of the tile kernel; `bid` is its position in the launch order.
template <int MODE, int L, int SB, int UNF>
__device__ __forceinline__ void unproject_block(const UnprojArgs& a, int bid,
                                                unsigned char* smem) {
  // XCD-aware work order: workgroups b, b+8, b+16.. share an XCD (and its L2);
  // give each such group one contiguous run of (batch item, tile) pairs so an
  // XCD's L2 holds the heatmaps of one frame at a time.
  const int run = bid >> 3;
  const int item = (bid & 7) * a.chunk + run;
  if (run >= a.chunk || item >= a.total) return;
  // (integer division runs on the VALU: tell the compiler the result is uniform)
  const int b = __builtin_amdgcn_readfirstlane(item / a.tiles);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (item - b * a.tiles) * a.tile_vox + wave * a.wave_vox;  // wave's first voxel
  if (n0 >= a.N) return;
  unsigned char* wbase = smem + (size_t)wave * wave_lds_bytes(L);

  int frame = b;
  if (MODE == MODE_WINDOW && a.win_frame != nullptr) frame = uniform_load(a.win_frame, b);
  const bool dead = (a.live != nullptr && uniform_load_u8(a.live, b) == 0) ||
                    (unsigned)frame >= (unsigned)a.F;
  const int n = n0 + lane;  // phase A: this lane's voxel
  if (dead) {
    if (lane < a.wave_vox && n < a.N) {
      if (a.cubes != nullptr)
        for (int ch = 0; ch < a.C; ch++) a.cubes[((long long)b * a.Cs + ch) * a.N + n] = 0.0f;
      if (a.grids != nullptr) {
        float* g = a.grids + ((long long)b * a.N + n) * 3;
        g[0] = g[1] = g[2] = 0.0f;
      }
      if (a.mask != nullptr) {
        uint8_t* m = (uint8_t*)a.mask + ((long long)b * a.N + n) * a.mask_bytes;
        for (int k = 0; k < a.mask_bytes; k++) m[k] = 0;
      }
    }
    if (MODE == MODE_ROOT && a.pl01 != nullptr && a.wave_vox != kWaveVox)
      zmax_clear_columns(a, b, n0, lane);
    return;
  }

  unproject_wave<MODE, L, SB, UNF>(a, b, frame, n0, lane, wbase);
  constexpr int TSTR = kWaveVox + 1;
  const float* s_out = reinterpret_cast<const float*>(wbase);
  if (lane < a.wave_vox && n < a.N && a.cubes != nullptr) {
#if defined(TEMPO_ABLATE) && TEMPO_ABLATE == 2  // (almost) no stores
    if (s_out[lane] == 123.456f)
#endif
    store_tile(a.cubes + (long long)b * a.Cs * a.N, a.C, a.N, n, s_out + lane, TSTR);
  }
  if (a.pl01 != nullptr || a.pl02 != nullptr || a.pl12 != nullptr)
    planes_epilogue<4 * L>(a, b, n0, lane, s_out, TSTR);
#ifdef TEMPO_TIMELINE
  if (a.grids != nullptr) {
    constexpr int tl_off = wave_lds_bytes(L) - kTimelineBytes;
    TEMPO_TL(29);
    __builtin_amdgcn_s_waitcnt(0);  // stores drained
    TEMPO_TL(30);
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    if (lane == 0)
      reinterpret_cast<unsigned long long*>(wbase + tl_off)[31] =
          ((unsigned long long)xcc_id << 32) | hw_id;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 32 stamps of 8 bytes -> the first 256 bytes of this wave's 768 in `grids`
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(
        a.grids + ((long long)b * a.N + n0) * 3);
    if (lane < 32) dst[lane] = reinterpret_cast<unsigned long long*>(wbase + tl_off)[lane];
  }
#endif
}

template <int MODE, int L, int SB, int UNF>
__global__ __launch_bounds__(kTileWaves * 64) void unproject_kernel(const UnprojArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  unproject_block<MODE, L, SB, UNF>(a, blockIdx.x, smem);
}

// ------------------------------------------------- root grid, K z-columns per item
// The root detector's second stage reads the z-columns of its K best (x, y) cells
// (tempo_detector.py:204-206).  Instead of gathering them out of a written volume this kernel
// computes exactly those K * Z voxels: the same per-voxel arithmetic (unproject_wave), so the
// values are bit-identical to the volume's, and the volume itself need never exist.  One wave
// = 64 consecutive (column, z) pairs of one item; output [NB][K][C][Z].
template <int L, int UNF>
__global__ __launch_bounds__(kTileWaves * 64) void unproject_columns_kernel(const UnprojArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int item = blockIdx.x;
  if (item >= a.total) return;
  const int b = __builtin_amdgcn_readfirstlane(item / a.tiles);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // short waves (a.wave_vox = 16 voxels): the launch is a few thousand voxels, bound by the
  // latency of its dependent fetches, so it wants many waves rather than full ones
  const int n0 = (item - b * a.tiles) * a.tile_vox + wave * a.wave_vox;
  if (n0 >= a.N) return;
  unsigned char* wbase = smem + (size_t)wave * wave_lds_bytes(L);
  const int n = n0 + lane;
  const bool mine = lane < a.wave_vox && n < a.N;
  int k = 0, i2 = 0, unused = 0;
  if (mine) voxel_index(a, n, &unused, &k, &i2);
  const int Z = a.dim[2];
  float* dst = a.col_out + (((long long)b * a.K + k) * a.C) * Z + i2;  // + ch * Z
  const bool dead = (a.live != nullptr && uniform_load_u8(a.live, b) == 0);
  if (dead) {
    if (mine)
      for (int ch = 0; ch < a.C; ch++) dst[ch * Z] = 0.0f;
    return;
  }
  unproject_wave<MODE_COLUMNS, L, 4, UNF>(a, b, b, n0, lane, wbase);
  constexpr int TSTR = kWaveVox + 1;
  const float* s_out = reinterpret_cast<const float*>(wbase);
  if (mine)
    for (int ch = 0; ch < a.C; ch++) dst[ch * Z] = s_out[ch * TSTR + lane];
}

// ------------------------------- window: which voxel does each plane entry's gradient go to?
// Training through the fused planes (tempo_pose.py:172-176 takes torch.max of the window volume
// over z, y and x): autograd sends a plane entry's gradient to ONE voxel of its line -- the
// first that attains the maximum (CPU torch.max; once a NaN is met it is the maximum) -- and
// torch.clamp's backward lets it through only if that voxel's mean was inside [0, 1].  With
// the FINAL planes given, this kernel recomputes every voxel's mean (unproject_wave, the same
// arithmetic as the kernel that made the planes), and a voxel whose clamped value equals its
// line's maximum files a claim, 2 * index + blocked, by atomic minimum: the smallest index
// wins, and its own blocked bit with it.
// Windows of 64 voxels along z make a wave one z-column: its claims for the xy plane all go
// to one word and are reduced inside the wave first.
template <int L, int UNF>
__global__ __launch_bounds__(kTileWaves * 64) void window_claim_kernel(const UnprojArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int SB = L >= 7 ? 2 : TEMPO_SB_WINDOW;
  const int run = blockIdx.x >> 3;
  const int item = (blockIdx.x & 7) * a.chunk + run;
  if (run >= a.chunk || item >= a.total) return;
  const int b = __builtin_amdgcn_readfirstlane(item / a.tiles);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (item - b * a.tiles) * a.tile_vox + wave * a.wave_vox;
  if (n0 >= a.N) return;
  unsigned char* wbase = smem + (size_t)wave * wave_lds_bytes(L);
  int frame = b;
  if (a.win_frame != nullptr) frame = uniform_load(a.win_frame, b);
  // a skipped window's volume is constant zero: whatever voxel its entries point at, the
  // backward adds nothing for it, so its claims are left alone
  if ((a.live != nullptr && uniform_load_u8(a.live, b) == 0) || (unsigned)frame >= (unsigned)a.F)
    return;
  unproject_wave<MODE_WINDOW, L, SB, UNF, 2>(a, b, frame, n0, lane, wbase);
  constexpr int TSTR = kWaveVox + 1;
  const float* s_out = reinterpret_cast<const float*>(wbase);
  const int n = n0 + lane;
  const bool mine = lane < a.wave_vox && n < a.N;
  int i0 = 0, i1 = 0, i2 = 0;
  if (mine) voxel_index(a, n, &i0, &i1, &i2);
  const int d0 = a.dim[0], d1 = a.dim[1], d2 = a.dim[2];
  const bool column = d2 == kWaveVox;  // the wave is one z-column: one xy entry per channel
  // The maxima of a few channels at a time, their loads in flight together: one round trip
  // to the L2 per group instead of one per channel and plane (a chain of dependent misses
  // otherwise), without holding all 3 x 16 values (136 VGPRs, 3 waves per SIMD, when all
  // channels were loaded at once).  32-bit element offsets from one base per plane.
  // (Measured at 320 windows: groups of 4 channels 5.36 ms, of 8 6.41; the xy maxima through
  // the scalar cache instead -- they are wave-uniform in a z-column wave -- 5.98.)
  constexpr int G = 4;
  const unsigned* q01 = a.pl01 + (long long)b * a.C * d0 * d1;
  const unsigned* q02 = a.pl02 + (long long)b * a.C * d0 * d2;
  const unsigned* q12 = a.pl12 + (long long)b * a.C * d1 * d2;
  int32_t* k01 = a.claim[0] + (long long)b * a.C * d0 * d1;
  int32_t* k02 = a.claim[1] + (long long)b * a.C * d0 * d2;
  int32_t* k12 = a.claim[2] + (long long)b * a.C * d1 * d2;
  const int s01 = d0 * d1, s02 = d0 * d2, s12 = d1 * d2;  // (C * N * 4 < 2^36, s* < 2^31 / 4)
  const int o01 = mine ? i0 * d1 + i1 : 0, o02 = mine ? i0 * d2 + i2 : 0;
  const int o12 = mine ? i1 * d2 + i2 : 0;
  // A maximum of exactly 0 is attained by every voxel of its line (clamped values are >= 0),
  // so its first voxel holds the claim and nobody else needs to file one: with real
  // heatmaps (mostly zero) that is nearly every line.  Other ties are rare; their claims
  // simply race to the minimum.  No claim is read back here: fire-and-forget atomics.
  auto file = [&](int32_t* claim, int word) { atomicMin(claim, word); };
  // (a z-column wave: the xy claims of its channels are minima over the wave's own lanes,
  // collected with LDS atomics in the tile's padding column -- word 64 of every channel row,
  // which nobody reads -- and stored by one instruction at the end; as 15 DPP wave reductions
  // and 15 single-lane stores they were a third of the epilogue's 680 VALU instructions)
  unsigned* pad = reinterpret_cast<unsigned*>(wbase) + kWaveVox;  // + ch * TSTR
  if (column) {
    if (lane < 4 * L) pad[lane * TSTR] = 0xffffffffu;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll 1
  for (int c0 = 0; c0 < a.C; c0 += G) {
    unsigned m01[G], m02[G], m12[G];
    float raw[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int ch = c0 + g < a.C ? c0 + g : c0;  // (the tail group repeats its first channel)
      m01[g] = q01[(long long)ch * s01 + o01];
      m02[g] = q02[(long long)ch * s02 + o02];
      m12[g] = q12[(long long)ch * s12 + o12];
      raw[g] = s_out[ch * TSTR + lane];
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int ch = c0 + g;
      if (ch >= a.C) continue;  // (wave-uniform)
      const bool isnan = raw[g] != raw[g];
      const float v = isnan ? raw[g] : fminf(fmaxf(raw[g], 0.0f), 1.0f);
      const int blocked = (raw[g] >= 0.0f && raw[g] <= 1.0f) ? 0 : 1;
      auto hit = [&](unsigned bits, int index) {
        const float m = __uint_as_float(bits);
        return mine && (v == m || (isnan && m != m)) && (m != 0.0f || index == 0);
      };
      const bool h01 = hit(m01[g], i2);
      if (h01) {
        if (column)  // the wave IS the line
          __hip_atomic_fetch_min(pad + ch * TSTR, (unsigned)((i2 << 1) | blocked),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
          file(k01 + (long long)ch * s01 + o01, (i2 << 1) | blocked);
      }
      if (hit(m02[g], i1)) file(k02 + (long long)ch * s02 + o02, (i1 << 1) | blocked);
      if (hit(m12[g], i0)) file(k12 + (long long)ch * s12 + o12, (i0 << 1) | blocked);
    }
  }
  if (column) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < a.C) {
      const unsigned w = pad[lane * TSTR];
      const int e = __builtin_amdgcn_readfirstlane(o01);  // (every lane of the wave is a voxel)
      if (w != 0xffffffffu) k01[(long long)lane * s01 + e] = (int)w;
    }
  }
}

// --------------------------------------------- window + axis-max planes, fused
// 64-voxel columns (VZ == 64).  A workgroup is kPlaneWaves waves = neighbouring y
// of one window; every wave walks kPlaneXR columns along x, so that
//   max over z  is a reduction over the wave's tile (one plain store per channel),
//   max over x  is a running maximum in registers, one global atomic max per
//               channel at the end of the walk (1/kPlaneXR per column),
//   max over y  goes through LDS atomics into per-workgroup accumulators that
//               are flushed every kPlaneXB columns with one global atomic max per
//               (x, channel) row (1/kPlaneWaves per column).
// The work order keeps one window per XCD at a time, like unproject_kernel.
// (One global atomic per column, channel and plane -- the generic
// planes_epilogue -- doubles the kernel time: the L2 executes about one atomic
// lane per clock and channel.)
#ifndef TEMPO_PL_WAVES
#define TEMPO_PL_WAVES 4
#endif
#ifndef TEMPO_PL_XB
#define TEMPO_PL_XB 4
#endif
#ifndef TEMPO_PL_XR
#define TEMPO_PL_XR 8
#endif
#ifndef TEMPO_PL_SB
#define TEMPO_PL_SB 2
#endif
constexpr int kPlaneXR = TEMPO_PL_XR;  // columns a wave walks along x
constexpr int kPlaneWaves = TEMPO_PL_WAVES;
constexpr int kPlaneXB = TEMPO_PL_XB;
constexpr int kAccStr = kWaveVox + 2;  // row stride of the y-max accumulators (words)
constexpr int plane_lds_bytes(int L) {
  // the waves' record / tile regions, the y-max accumulators, 32 z-max words per wave
  return kPlaneWaves * wave_lds_bytes(L) + kPlaneXB * 4 * L * kAccStr * 4 + kPlaneWaves * 128;
}

// Epilogue in registers (CP = 16; 32-channel calls come in as two 16-channel launches,
// launch_forward): the finished values stay in the gather layout (unproject_wave<KEEP>) -- lane =
// (voxel vq, channel quad cq), slot (s, j) = channel 4 cq + j of voxel z = s * VPS + vq -- and
//   max over x  is 4 NSUB running maxima per lane,
//   max over y  one ds_max_u32 per slot into the workgroup's accumulators (same layout),
//   max over z  the maximum over a lane's sub-steps, over the lanes of its 16-lane row that
//               hold the same quad (two DPP rotations), then one ds_max_u32 per row into
//               16 words of the wave's LDS,
// and the LDS tile is written only when the volume is stored.  Against the tile form (every
// value through LDS twice, -0.0 patched on every read: 165 VALU + 64 LDS instructions per
// column on top of the tile kernel's 1141 + 78) this is ~60 VALU + 21 LDS.
template <int L, int UNF>
__global__ __launch_bounds__(kPlaneWaves * 64) void unproject_planes_kernel(const UnprojArgs a) {
  constexpr int SB = TEMPO_PL_SB;  // two sub-steps of fetches in flight (fewer waves fit here)
  constexpr int CH = 4 * L;
  constexpr int TSTR = kWaveVox + 1;
  constexpr bool REG = L == 4;
  constexpr int VPS = 64 / L;
  constexpr int NSUB = (kWaveVox + VPS - 1) / VPS;
  constexpr int SL = REG ? 4 * NSUB : CH;  // accumulator rows per column (== CH for L = 4, 8)
  extern __shared__ __align__(16) unsigned char smem[];
  const int bid = blockIdx.x;
  const int run = bid >> 3;
  const int item = (bid & 7) * a.chunk + run;  // XCD-aware order, as unproject_kernel
  if (run >= a.chunk || item >= a.total) return;
  const int VX = a.dim[0], VY = a.dim[1];
  const int yblocks = VY / kPlaneWaves;
  const int xblocks = (VX + kPlaneXR - 1) / kPlaneXR;
  const int per_win = yblocks * xblocks;
  const int b = __builtin_amdgcn_readfirstlane(item / per_win);
  const int rem = item - b * per_win;
  const int xb = __builtin_amdgcn_readfirstlane(rem / yblocks);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int y = (rem - xb * yblocks) * kPlaneWaves + wave;
  // The walk of x-block xb visits x = xb, xb + xblocks, xb + 2 xblocks, ...: the workgroups
  // of a window (resident on one XCD together) are at neighbouring x at any time, a slab of
  // the window a few columns thick, rather than spread over its whole depth (4.147 against
  // 4.175 ms at config 3 for x-block xb walking xb * kPlaneXR + step)
  const int x_begin = xb, x_step = xblocks, x_end = VX;
  unsigned char* wbase = smem + (size_t)wave * wave_lds_bytes(L);
  unsigned* acc = reinterpret_cast<unsigned*>(smem + (size_t)kPlaneWaves * wave_lds_bytes(L));
  unsigned* zacc = acc + kPlaneXB * CH * kAccStr + wave * 32;  // [CH <= 32] per wave

  int frame = b;
  if (a.win_frame != nullptr) frame = uniform_load(a.win_frame, b);
  const bool dead = (a.live != nullptr && uniform_load_u8(a.live, b) == 0) ||
                    (unsigned)frame >= (unsigned)a.F;
  if (dead) {  // the planes were cleared by the host; the volume is zero-filled here
    if (a.cubes != nullptr)
      for (int x = x_begin; x < x_end; x += x_step)
        for (int ch = 0; ch < a.C; ch++)
          a.cubes[((long long)b * a.Cs + ch) * a.N + (x * VY + y) * kWaveVox + lane] = 0.0f;
    return;
  }
  for (int r = wave; r < kPlaneXB * CH; r += kPlaneWaves) acc[r * kAccStr + lane] = 0u;
  if (lane < 32) zacc[lane] = 0u;
  __syncthreads();

  const long long bc = (long long)b * a.Cs;
  unsigned over_x[REG ? SL : CH];
#pragma unroll
  for (int i = 0; i < (REG ? SL : CH); i++) over_x[i] = 0u;
  const float* s_out = reinterpret_cast<const float*>(wbase);
  // gather layout of this lane (REG)
  const int vq = lane / L, cq = lane - vq * L;
  const bool lane_on = lane < VPS * L;
#pragma unroll 1
  for (int x = x_begin, walk = 0; x < x_end; x += x_step, walk++) {
    const int n0 = (x * VY + y) * kWaveVox;
    const int xs = walk % kPlaneXB;
    if (REG) {
      unsigned pv[SL];
      unproject_wave<MODE_WINDOW, L, SB, UNF, 1>(a, b, frame, n0, lane, wbase, pv);
      if (a.cubes != nullptr)
        store_tile(a.cubes + (long long)b * a.Cs * a.N, a.C, a.N, n0 + lane, s_out + lane, TSTR);
      // the accumulators are [channel][z] rows (kAccStr apart: with 66 the 32 lanes of an
      // LDS cycle, 8 voxels x 4 quads, fall into 32 different banks), so that the flush reads
      // them as rows and every global atomic instruction covers 256 contiguous bytes
#pragma unroll
      for (int i = 0; i < SL; i++) {
        over_x[i] = pv[i] > over_x[i] ? pv[i] : over_x[i];
        const int ch = 4 * cq + (i & 3), z = (i >> 2) * VPS + vq;
        __hip_atomic_fetch_max(acc + (xs * CH + ch) * kAccStr + z, pv[i], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (a.pl01 != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          unsigned m = pv[j];
#pragma unroll
          for (int s = 1; s < NSUB; s++) m = pv[s * 4 + j] > m ? pv[s * 4 + j] : m;
          // the lanes of this 16-lane row that hold the same quad are L apart
          if (L == 4) {
            unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, true);
            m = o > m ? o : m;  // row_ror:4
          }
          {
            unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, true);
            m = o > m ? o : m;  // row_ror:8
          }
          if ((lane & 15) < L)
            __hip_atomic_fetch_max(zacc + 4 * cq + j, m, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < CH) {
          const unsigned m = zacc[lane];
          zacc[lane] = 0u;
          if (lane < a.C) a.pl01[((bc + lane) * VX + x) * VY + y] = m;
        }
      }
    } else {
      unproject_wave<MODE_WINDOW, L, SB, UNF>(a, b, frame, n0, lane, wbase);
      if (a.cubes != nullptr)
        store_tile(a.cubes + (long long)b * a.Cs * a.N, a.C, a.N, n0 + lane, s_out + lane, TSTR);
#pragma unroll
      for (int ch = 0; ch < CH; ch++) {
        if ((ch & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // four rows in flight, not CH
        unsigned v = __float_as_uint(s_out[ch * TSTR + lane]);
        v = v == 0x80000000u ? 0u : v;  // -0.0 orders as +0.0
        over_x[ch] = v > over_x[ch] ? v : over_x[ch];
        if (v != 0u)
          __hip_atomic_fetch_max(acc + (xs * CH + ch) * kAccStr + lane, v, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (a.pl01 != nullptr) {
        // max over z: four lanes per channel row of the tile, 16 voxels each, then
        // the quad is combined (16 channels per pass)
#pragma unroll
        for (int c0 = 0; c0 < CH; c0 += 16) {
          const int ch = c0 + (lane >> 2);
          const float* row = s_out + (ch < CH ? ch : 0) * TSTR + (lane & 3) * 16;
          unsigned m = 0u;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            if ((k & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            unsigned v = __float_as_uint(row[k]);
            v = v == 0x80000000u ? 0u : v;  // -0.0 orders as +0.0
            m = v > m ? v : m;
          }
          __builtin_amdgcn_sched_barrier(0);
          unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true);
          m = o > m ? o : m;  // quad_perm:[1,0,3,2]
          o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, true);
          m = o > m ? o : m;  // quad_perm:[2,3,0,1]
          if ((lane & 3) == 0 && ch < a.C) a.pl01[((bc + ch) * VX + x) * VY + y] = m;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (xs == kPlaneXB - 1 || x + x_step >= x_end) {
      __syncthreads();
      for (int r = wave; r < (xs + 1) * CH; r += kPlaneWaves) {
        const int rx = r / CH, ch = r - rx * CH;
        const unsigned v = acc[r * kAccStr + lane];
        acc[r * kAccStr + lane] = 0u;
        if (v != 0u && ch < a.C && a.pl02 != nullptr)
          __hip_atomic_fetch_max(
              a.pl02 + ((bc + ch) * VX + (x - (xs - rx) * x_step)) * kWaveVox + lane, v,
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
  if (a.pl12 != nullptr) {
    if (REG) {  // gather layout -> [channel][z] rows through the wave's (now idle) LDS region
      unsigned* t = reinterpret_cast<unsigned*>(wbase);
#pragma unroll
      for (int i = 0; i < SL; i++) {
        const int z = (i >> 2) * VPS + vq;
        if (lane_on && z < kWaveVox) t[(4 * cq + (i & 3)) * TSTR + z] = over_x[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    auto put = [&](int ch, unsigned v) {
      unsigned* dst = a.pl12 + ((bc + ch) * VY + y) * kWaveVox + lane;
      if (xblocks == 1) *dst = v;  // this wave owns (window, y) entirely
      else if (v != 0u)
        __hip_atomic_fetch_max(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (REG) {
      for (int ch = 0; ch < a.C; ch++)
        put(ch, reinterpret_cast<const unsigned*>(wbase)[ch * TSTR + lane]);
    } else {
#pragma unroll
      for (int ch = 0; ch < CH; ch++)
        if (ch < a.C) put(ch, over_x[ch]);
    }
  }
}

// ------------------------------------------------------------ grouped launch
// Batch items (root) / windows that share one calibration and one voxel lattice
// -- the frames of one sequence, the T timesteps of one proposal -- project to
// the same texel footprints.  One wave = 64 voxels x the items of one group:
//   project   once per view, as in unproject_kernel; the footprint is parked in
//             LDS as a 16-byte record (top-left texel byte offset, edge flags,
//             the two bilinear fractions) for ALL views;
//   per item  fetch + mix over the views the wave needs, reading the records:
//             pure gather, two half-views of fetches in flight, the next
//             half-view (or the next item's first) issued before each mix;
//             then normalise, clamp, transpose through LDS and store, with the
//             reciprocal of the view count computed once per group.
// The arithmetic per (voxel, view, item) is the one of unproject_kernel, so the
// results are bit-identical to an ungrouped launch.
struct __align__(16) GroupRec {
  unsigned base;   // byte offset of texel (x0, y0) in the view's map (may wrap: outside)
  unsigned flags;  // bit 0: column x0 inside the map and needed, bit 1: column x0 + 1
  float w, n;      // ix - floor(ix), iy - floor(iy)
};

constexpr int group_wave_lds(int V, int L) {
  return V * kWaveVox * (int)sizeof(GroupRec) + kWaveVox * 16 + 4 * L * (kWaveVox + 1) * 4;
}

template <int MODE, int L, int UNF>
__global__ __launch_bounds__(kThreads) void unproject_group_kernel(const UnprojArgs a) {
  constexpr int VPS = 64 / L;
  constexpr int NSUB = (kWaveVox + VPS - 1) / VPS;
  constexpr int G = (NSUB + 1) / 2;  // sub-steps per half view
  extern __shared__ __align__(16) unsigned char smem[];

  const int bid = blockIdx.x;
  const int run = bid >> 3;
  const int item = (bid & 7) * a.chunk + run;  // XCD-aware order, as unproject_kernel
  if (run >= a.chunk || item >= a.total) return;
  const int grp = __builtin_amdgcn_readfirstlane(item / a.tiles);
  const int first = uniform_load(a.groups, 2 * grp);
  const int nfr = uniform_load(a.groups, 2 * grp + 1);
  if (first < 0 || nfr <= 0 || first > a.NB - nfr) return;  // malformed table: touch nothing
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (item - grp * a.tiles) * a.tile_vox + wave * kWaveVox;
  if (n0 >= a.N) return;
  const int n = n0 + lane;
  constexpr int CPB = L * 16;  // bytes per texel: CP = 4 L channels (host-checked)

  unsigned char* wbase = smem + (size_t)wave * a.wave_lds;
  GroupRec* s_rec = reinterpret_cast<GroupRec*>(wbase);  // [V][64]
  DenRec* s_den = reinterpret_cast<DenRec*>(wbase + (size_t)a.V * kWaveVox * sizeof(GroupRec));
  constexpr int TSTR = kWaveVox + 1;
  float* s_out = reinterpret_cast<float*>(s_den + kWaveVox);  // [4L][TSTR]

  auto frame_of = [&](int b) -> int {
    return (MODE == MODE_WINDOW && a.win_frame != nullptr) ? uniform_load(a.win_frame, b) : b;
  };
  auto dead_item = [&](int b) -> bool {
    return (a.live != nullptr && uniform_load_u8(a.live, b) == 0) ||
           (unsigned)frame_of(b) >= (unsigned)a.F;
  };

  // the group's calibration is the one of its first item
  const int frame0 = frame_of(first);
  const bool cam_ok = (unsigned)frame0 < (unsigned)a.F;

  float x = 0.f, y = 0.f, z = 0.f;
  const bool have = cam_ok && n < a.N && voxel_world<MODE>(a, first, n, &x, &y, &z);
  asm volatile("" : "+v"(x), "+v"(y), "+v"(z));

  // ---- project: every view once, records to LDS
  int cnt = 0;
  unsigned mbits = 0;
  unsigned amask = 0;  // slots some lane of the wave needs
  if (cam_ok) {
#pragma unroll 1
    for (int slot = 0; slot < a.V; slot++) {
      const int view = slot_view(a, slot);
      const TempoCam cam = load_cam(a.cam, frame0 * a.V + view);
      TempoPix p;
      p.px = p.py = 0.f;
      p.inside = 0;
      bool need = false;
      if (have) {
        p = tempo_world_to_pixel(&cam, x, y, z, UNF);
        need = MODE != MODE_ROOT || p.inside || p.px != p.px || p.py != p.py;
      }
      cnt += p.inside;
      mbits |= (unsigned)p.inside << view;
      if (__builtin_amdgcn_ballot_w64(need) != 0) {
        float gx, gy;
        tempo_pixel_to_grid(&cam, p.px, p.py, &a.geom, UNF, &gx, &gy);
        float fw, fn;
        int x0, y0;
        tempo_bilinear_frac(gx, gy, a.W, a.H, &x0, &y0, &fw, &fn);
        GroupRec r;
        r.base = (unsigned)(texel_index(y0, a.W, x0) * CPB);
        r.flags = ((need && (unsigned)x0 < (unsigned)a.W) ? 1u : 0u) |
                  ((need && (unsigned)(x0 + 1) < (unsigned)a.W) ? 2u : 0u);
        r.w = fw;
        r.n = fn;
        s_rec[slot * kWaveVox + lane] = r;
        amask |= 1u << slot;
      }
    }
  }
  amask = __builtin_amdgcn_readfirstlane(amask);
  {
    s_den[lane] = make_den((MODE == MODE_ROOT ? (float)cnt : (float)a.V) + 1e-6f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- gather layout: L lanes per voxel, VPS voxels per sub-step
  const bool lane_on = lane < VPS * L;
  const int vq = lane / L;
  const int cq = lane - vq * L;
  const unsigned lo = (unsigned)cq * 16u;
  const unsigned rowb = (unsigned)(a.W * CPB);
  DenRec dr[NSUB];
  int rdv[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; s++) {
    const int vsl = s * VPS + vq;
    rdv[s] = (lane_on && vsl < kWaveVox) ? vsl : 0;
    dr[s] = s_den[rdv[s]];
  }

  u32x4 tex[2][G][4];
  auto make_rsrc = [&](int b, int view, bool real) -> __amdgpu_buffer_rsrc_t {
    // a dead item's maps are never read: zero records
    const bool dd = !real || dead_item(b) || !cam_ok;
    const int fr = dd ? 0 : frame_of(b);
    const unsigned long long vb = (unsigned long long)(uintptr_t)(
        reinterpret_cast<const char*>(a.hm + (long long)fr * a.frame_stride) +
        (long long)view * a.view_bytes);
    const unsigned vb_lo = __builtin_amdgcn_readfirstlane((unsigned)vb);
    const unsigned vb_hi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>((uintptr_t)(((unsigned long long)vb_hi << 32) | vb_lo)), 0,
        __builtin_amdgcn_readfirstlane(dd ? 0 : (int)a.view_bytes), 0x00020000);
  };
  // half h of the view in `slot` -> tex[h].  Always issued (an unconditional
  // instruction stream keeps the wait counts exact); `real` = false reads a
  // zero-sized buffer: no memory traffic
  auto fetch = [&](auto hc, int b, int slot, bool real) {
    constexpr int h = decltype(hc)::value;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(b, slot_view(a, slot), real);
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int s = h * G + g;
      if (s < NSUB) {
        const GroupRec r = s_rec[slot * kWaveVox + rdv[s]];
        const unsigned o0 = (r.flags & 1u) ? r.base : kInvalid;
        const unsigned o1 = (r.flags & 2u) ? r.base + CPB : kInvalid;
        const unsigned o2 = (r.flags & 1u) ? r.base + rowb : kInvalid;
        const unsigned o3 = (r.flags & 2u) ? r.base + rowb + CPB : kInvalid;
        tex[h][g][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, o0 + lo, 0, 0);
        tex[h][g][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, o1 + lo, 0, 0);
        tex[h][g][2] = __builtin_amdgcn_raw_buffer_load_b128(rs, o2 + lo, 0, 0);
        tex[h][g][3] = __builtin_amdgcn_raw_buffer_load_b128(rs, o3 + lo, 0, 0);
      }
    }
  };
  f32x2 acc01[NSUB], acc23[NSUB];
  auto mix = [&](auto hc, int slot) {
    constexpr int h = decltype(hc)::value;
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int s = h * G + g;
      if (s < NSUB) {
        const GroupRec r = s_rec[slot * kWaveVox + rdv[s]];
        // the weights of tempo_bilinear_tap, from the stored fractions
        const float e = 1.0f - r.w, sth = 1.0f - r.n;
        const f32x2 nw = tempo_splat(sth * e), ne = tempo_splat(sth * r.w);
        const f32x2 sw = tempo_splat(r.n * e), se = tempo_splat(r.n * r.w);
        const f32x2 a01 = {__uint_as_float(tex[h][g][0][0]), __uint_as_float(tex[h][g][0][1])};
        const f32x2 a23 = {__uint_as_float(tex[h][g][0][2]), __uint_as_float(tex[h][g][0][3])};
        const f32x2 b01 = {__uint_as_float(tex[h][g][1][0]), __uint_as_float(tex[h][g][1][1])};
        const f32x2 b23 = {__uint_as_float(tex[h][g][1][2]), __uint_as_float(tex[h][g][1][3])};
        const f32x2 c01 = {__uint_as_float(tex[h][g][2][0]), __uint_as_float(tex[h][g][2][1])};
        const f32x2 c23 = {__uint_as_float(tex[h][g][2][2]), __uint_as_float(tex[h][g][2][3])};
        const f32x2 d01 = {__uint_as_float(tex[h][g][3][0]), __uint_as_float(tex[h][g][3][1])};
        const f32x2 d23 = {__uint_as_float(tex[h][g][3][2]), __uint_as_float(tex[h][g][3][3])};
        f32x2 v01 = a01 * nw, v23 = a23 * nw;
        v01 = __builtin_elementwise_fma(b01, ne, v01);
        v23 = __builtin_elementwise_fma(b23, ne, v23);
        v01 = __builtin_elementwise_fma(c01, sw, v01);
        v23 = __builtin_elementwise_fma(c23, sw, v23);
        v01 = __builtin_elementwise_fma(d01, se, v01);
        v23 = __builtin_elementwise_fma(d23, se, v23);
        acc01[s] = acc01[s] + v01;
        acc23[s] = acc23[s] + v23;
      }
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  const int slot0 = amask ? __builtin_ctz(amask) : 0;
  fetch(H0{}, first, slot0, amask != 0);
#pragma unroll 1
  for (int fi = 0; fi < nfr; fi++) {
    const int b = first + fi;
    const bool dead = !cam_ok || dead_item(b);
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      acc01[s] = tempo_splat(0.f);
      acc23[s] = tempo_splat(0.f);
    }
    unsigned mm = amask;
#pragma unroll 1
    while (mm) {
      const int slot = __builtin_ctz(mm);
      mm &= mm - 1;
      __builtin_amdgcn_sched_barrier(0);
      fetch(H1{}, b, slot, true);
      __builtin_amdgcn_sched_barrier(0);
      mix(H0{}, slot);
      __builtin_amdgcn_sched_barrier(0);
      {
        // the next half view: of this item, or the first of the next item
        const bool more = mm != 0, last = !more && fi + 1 >= nfr;
        fetch(H0{}, (more || last) ? b : b + 1, more ? __builtin_ctz(mm) : slot0, !last);
      }
      __builtin_amdgcn_sched_barrier(0);
      mix(H1{}, slot);
    }

    // ---- normalise, clamp, transpose, store (as unproject_kernel's phase C)
    const bool slow = normalise_tile<NSUB>(acc01, acc23, dr);
#pragma unroll
    for (int s = 0; s < NSUB; s++) {
      const int vsl = s * VPS + vq;
      if (lane_on && vsl < kWaveVox) {
        float* o = s_out + (4 * cq) * TSTR + vsl;
        if (!slow) {
          o[0 * TSTR] = __builtin_amdgcn_fmed3f(acc01[s].x, 0.0f, 1.0f);
          o[1 * TSTR] = __builtin_amdgcn_fmed3f(acc01[s].y, 0.0f, 1.0f);
          o[2 * TSTR] = __builtin_amdgcn_fmed3f(acc23[s].x, 0.0f, 1.0f);
          o[3 * TSTR] = __builtin_amdgcn_fmed3f(acc23[s].y, 0.0f, 1.0f);
        } else {
          o[0 * TSTR] = tempo_nan_clamp(acc01[s].x, 0.0f, 1.0f);
          o[1 * TSTR] = tempo_nan_clamp(acc01[s].y, 0.0f, 1.0f);
          o[2 * TSTR] = tempo_nan_clamp(acc23[s].x, 0.0f, 1.0f);
          o[3 * TSTR] = tempo_nan_clamp(acc23[s].y, 0.0f, 1.0f);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (n < a.N) {
      float* dst = a.cubes + (long long)b * a.C * a.N + n;
      if (dead) {
        for (int ch = 0; ch < a.C; ch++) dst[(long long)ch * a.N] = 0.0f;
      } else {
        store_tile(a.cubes + (long long)b * a.C * a.N, a.C, a.N, n, s_out + lane, TSTR);
      }
      if (MODE == MODE_ROOT && a.grids != nullptr) {
        float* gp = a.grids + ((long long)b * a.N + n) * 3;
        gp[0] = dead ? 0.0f : x;
        gp[1] = dead ? 0.0f : y;
        gp[2] = dead ? 0.0f : z;
      }
      if (MODE == MODE_ROOT && a.mask != nullptr) {
        char* m = (char*)a.mask + ((long long)b * a.N + n) * a.mask_bytes;
        const unsigned mb = dead ? 0u : mbits;
        if (a.mask_bytes == 1) *(uint8_t*)m = (uint8_t)mb;
        else if (a.mask_bytes == 2) *(uint16_t*)m = (uint16_t)mb;
        else *(uint32_t*)m = mb;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------ backward
// Gradient of the volume with respect to the heatmaps (the only input with a
// gradient: cameras and grids are constants).  Mirrors autograd through the
// reference's op chain: clamp (mask recorded by the forward) -> division by
// count + 1e-6 -> bounding mask -> grid_sample's backward, which adds
// weight * gradient into the four texels of every sample.  Same tiling and lane
// layout as unproject_kernel; every view is projected first (records in LDS, as
// in the grouped launch), then each lane adds its four channels into the four
// texels with hardware fp32 atomics.  The accumulation order is not fixed, so
// results agree with autograd to rounding, not bit for bit.
constexpr int backward_wave_lds(int V, int L) {
  return V * kWaveVox * (int)sizeof(GroupRec) + kWaveVox * 4 + 4 * L * (kWaveVox + 1) * 4;
}

template <int MODE, int L, int UNF>
__global__ __launch_bounds__(kThreads) void unproject_backward_kernel(const UnprojArgs a) {
  constexpr int VPS = 64 / L;
  constexpr int CPB = L * 16;
  constexpr int TSTR = kWaveVox + 1;
  extern __shared__ __align__(16) unsigned char smem[];
  const int bid = blockIdx.x;
  const int run = bid >> 3;
  const int item = (bid & 7) * a.chunk + run;
  if (run >= a.chunk || item >= a.total) return;
  const int b = __builtin_amdgcn_readfirstlane(item / a.tiles);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = (item - b * a.tiles) * a.tile_vox + wave * kWaveVox;
  if (n0 >= a.N) return;
  const int n = n0 + lane;
  int frame = b;
  if (MODE == MODE_WINDOW && a.win_frame != nullptr) frame = uniform_load(a.win_frame, b);
  if ((a.live != nullptr && uniform_load_u8(a.live, b) == 0) || (unsigned)frame >= (unsigned)a.F)
    return;  // a skipped item's volume is constant zero: no gradient

  unsigned char* wbase = smem + (size_t)wave * a.wave_lds;
  GroupRec* s_rec = reinterpret_cast<GroupRec*>(wbase);  // [slot1 - slot0][64]
  float* s_den = reinterpret_cast<float*>(
      wbase + (size_t)(a.slot1 - a.slot0) * kWaveVox * sizeof(GroupRec));
  float* s_g = s_den + kWaveVox;  // [4L][TSTR] incoming gradient, lane = voxel

  const float* gout = a.cubes + (long long)b * a.C * a.N;
  for (int ch = 0; ch < 4 * L; ch++)
    s_g[ch * TSTR + lane] = (ch < a.C && n < a.N) ? gout[(long long)ch * a.N + n] : 0.0f;

  float x = 0.f, y = 0.f, z = 0.f;
  const bool have = n < a.N && voxel_world<MODE>(a, b, n, &x, &y, &z);
  int cnt = 0;
  unsigned amask = 0;
  // every view is projected (the masked mean's denominator counts them all); the
  // footprints of slots [slot0, slot1) are kept and scattered by this launch
#pragma unroll 1
  for (int slot = 0; slot < a.V; slot++) {
    const int view = slot_view(a, slot);
    const TempoCam cam = load_cam(a.cam, frame * a.V + view);
    TempoPix p;
    p.px = p.py = 0.f;
    p.inside = 0;
    bool need = false;
    if (have) {
      p = tempo_world_to_pixel(&cam, x, y, z, UNF);
      need = MODE != MODE_ROOT || p.inside || p.px != p.px || p.py != p.py;
    }
    cnt += p.inside;
    const bool mine = slot >= a.slot0 && slot < a.slot1;  // wave-uniform
    if (mine && __builtin_amdgcn_ballot_w64(need) != 0) {
      float gx, gy, fw, fn;
      int x0, y0;
      tempo_pixel_to_grid(&cam, p.px, p.py, &a.geom, UNF, &gx, &gy);
      tempo_bilinear_frac(gx, gy, a.W, a.H, &x0, &y0, &fw, &fn);
      GroupRec r;
      r.base = (unsigned)(texel_index(y0, a.W, x0) * CPB);
      r.flags = ((need && (unsigned)x0 < (unsigned)a.W) ? 1u : 0u) |
                ((need && (unsigned)(x0 + 1) < (unsigned)a.W) ? 2u : 0u);
      r.w = fw;
      r.n = fn;
      s_rec[(slot - a.slot0) * kWaveVox + lane] = r;
      amask |= 1u << (slot - a.slot0);
    }
  }
  amask = __builtin_amdgcn_readfirstlane(amask);
  s_den[lane] = (MODE == MODE_ROOT ? (float)cnt : (float)a.V) + 1e-6f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // Scatter layout: CP adjacent lanes = the CP channels of one texel, so every
  // atomic instruction adds whole 64-byte (CP = 16) texels: the L2 retires atomics
  // line by line, and a quarter-filled line costs as much as a full one (the
  // forward's 4-channels-per-lane layout measured 4x slower here).
  constexpr int CPL = 4 * L;        // lanes per voxel
  constexpr int VP2 = 64 / CPL;     // voxels per step
  constexpr int NS2 = (kWaveVox + VP2 - 1) / VP2;
  const bool lane_on = lane < VP2 * CPL;
  const int vq2 = lane / CPL;
  const int c = lane - vq2 * CPL;  // this lane's channel
  const unsigned long long* mw =
      a.cmask + ((long long)b * ((a.N + kWaveVox - 1) / kWaveVox) + (n0 >> 6)) * mask_words(L);
  // the forward's mask layout: word [s * 4 + j], bit vq * L + cq for voxel
  // s * VPS + vq and channel 4 cq + j
  const int cq = c >> 2, j = c & 3;

  const unsigned rowb = (unsigned)(a.W * CPB);
  unsigned mm = amask;
#pragma unroll 1
  while (mm) {
    const int slot = __builtin_ctz(mm);  // relative to slot0
    mm &= mm - 1;
    const int view = slot_view(a, a.slot0 + slot);
    char* vbase = reinterpret_cast<char*>(a.ghm + (long long)frame * a.frame_stride) +
                  (long long)view * a.view_bytes + c * 4;
#pragma unroll 4
    for (int s2 = 0; s2 < NS2; s2++) {
      const int vsl = s2 * VP2 + vq2;
      if (!(lane_on && vsl < kWaveVox)) continue;
      const GroupRec r = s_rec[slot * kWaveVox + vsl];
      if (r.flags == 0u) continue;
      // d(volume)/d(masked view sum): gradient / (count + 1e-6) where the clamp
      // passes it, else 0
      const int s = vsl / VPS, vq = vsl - s * VPS;
      const unsigned long long word = mw[s * 4 + j];
      if (!((word >> (vq * L + cq)) & 1ull)) continue;
      const float gs = s_g[c * TSTR + vsl] / s_den[vsl];
      const float e = 1.0f - r.w, sth = 1.0f - r.n;
      const float wt[4] = {sth * e, sth * r.w, r.n * e, r.n * r.w};
      const unsigned off[4] = {r.base, r.base + CPB, r.base + rowb, r.base + rowb + CPB};
#pragma unroll
      for (int t = 0; t < 4; t++) {
        // column inside the map (flags) and row inside the map (byte range)
        if (!((r.flags >> (t & 1)) & 1u) || off[t] >= a.view_bytes) continue;
        const float v = gs * wt[t];
        if (v != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(vbase + off[t]), v);
      }
    }
  }
}

// ------------------------------------------- window backward, pre-aggregated
// The fine lattice's voxels are one to a few heatmap pixels apart, so the
// four-texel footprints of neighbouring voxels overlap and the per-sample
// scatter above sends several atomics to every texel it touches; the L2 retires
// about one atomic lane per clock and channel, which is what bounds that kernel.
// Here one wave (= one workgroup) owns a brick of 4 x 4 x 4 voxels and, per
// view, sums the brick's footprints into a wave-private LDS tile anchored at the
// brick's smallest (x0, y0), then flushes the tile with one global atomic per
// touched (texel, channel).  The tile holds kBrickCap texels in rows of the
// footprint box's own width, so any box of at most that area fits.
// LDS float atomics are no faster than the L2's (measured: 0.55 lanes per clock
// and CU), so the tile is accumulated with plain read-add-write: one voxel per
// step, lane = (texel of the footprint, channel), so the lanes of an instruction
// never collide and the LDS executes a wave's accesses in order.
// A footprint box larger than the tile (a near camera) is retried as the
// brick's two halves; a half that still does not fit goes to memory directly,
// as above: the tile is an accelerator, never a correctness condition.
// Config 3, ms per launch: 108.9 per-sample kernel; square 10 x 10 tile 26.9; box-shaped
// rows 24.2; 81 texels (14 waves per CU) 22.6; 12-byte voxel records (16 waves) 21.0;
// gradient loads issued together 20.2.  Tile size (round 2; config 3 / H36M slice, ms):
// 64 texels 22.9 / -, 81 20.7 / 5.96, 100 19.7 / 5.39, 121 19.8 / 5.00, 144 21.1 / 4.67 --
// a brick whose box does not fit is done as two halves, which costs more than the waves
// the larger tile displaces (12 instead of 16 per CU), up to about 1.5x the mean box.
// The tile size is chosen per launch (tempo_unproject_window_backward's tile_texels): 121
// texels suit lattices up to about two heatmap pixels per voxel (config 3: 1.5), 144 the
// coarser ones (H36M windows on 384 x 384 maps: 2.8 pixels per voxel, 5.00 -> 4.67 ms).
constexpr int kBrickCapSmall = 121, kBrickCapLarge = 144;  // texels per tile
struct BrickLds {
  int tile, stage, rec, total;
};
// CP = 16, 121 texels: 121 * 64 + 16 * 65 * 4 + 64 * 12 = 12672 bytes, 12 waves per CU
constexpr BrickLds brick_lds(int L, int cap) {
  BrickLds l{};
  l.tile = 0;
  l.stage = cap * 4 * L * 4;
  l.rec = l.stage + 4 * L * (kWaveVox + 1) * 4;
  l.total = l.rec + 3 * kWaveVox * 4;
  return l;
}

// PLN: the incoming gradient is that of the three axis-max planes (a.gplane), routed to the
// voxels that hold the planes' claims (a.claim, window_claim_kernel), instead of a gradient
// volume with the forward's clamp mask.  Most bricks then carry no gradient at all (the
// claimed voxels are a quarter of a window's, in two thirds of its bricks) and leave early.
template <int L, int UNF, int CAP, int PLN = 0>
__global__ __launch_bounds__(64) void window_backward_brick_kernel(const UnprojArgs a) {
  constexpr int MODE = MODE_WINDOW;
  constexpr int VPS = 64 / L;
  constexpr int CP = 4 * L;
  constexpr int TSTR = kWaveVox + 1;
  constexpr BrickLds LD = brick_lds(L, CAP);
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem + LD.tile);  // CAP texels x CP
  float* st = reinterpret_cast<float*>(smem + LD.stage);   // [CP][TSTR]
  // per voxel and view: 16 * (index of the footprint's top-left texel, in the tile or in
  // the map) + four bits saying which of its four texels are inside the map, and the two
  // bilinear fractions
  int* s_base = reinterpret_cast<int*>(smem + LD.rec);  // [64]
  float* s_fw = reinterpret_cast<float*>(smem + LD.rec) + kWaveVox;
  float* s_fn = s_fw + kWaveVox;

  const int bid = blockIdx.x;
  const int run = bid >> 3;
  const int item = (bid & 7) * a.chunk + run;
  if (run >= a.chunk || item >= a.total) return;
  const int b = __builtin_amdgcn_readfirstlane(item / a.tiles);
  const int lane = threadIdx.x;
  int frame = b;
  if (a.win_frame != nullptr) frame = uniform_load(a.win_frame, b);
  if ((a.live != nullptr && uniform_load_u8(a.live, b) == 0) || (unsigned)frame >= (unsigned)a.F)
    return;  // a skipped item's volume is constant zero: no gradient

  const int nb1 = (a.dim[1] + kBrickEdge - 1) / kBrickEdge;
  const int nb2 = (a.dim[2] + kBrickEdge - 1) / kBrickEdge;
  const int brick = item - b * a.tiles;
  const int bx = __builtin_amdgcn_readfirstlane(brick / (nb1 * nb2));
  const int by = __builtin_amdgcn_readfirstlane((brick - bx * nb1 * nb2) / nb2);
  const int bz = brick - (bx * nb1 + by) * nb2;
  const int i0 = bx * kBrickEdge + (lane >> 4);
  const int i1 = by * kBrickEdge + ((lane >> 2) & 3);
  const int i2 = bz * kBrickEdge + (lane & 3);
  const bool inb = i0 < a.dim[0] && i1 < a.dim[1] && i2 < a.dim[2];
  const int n = (i0 * a.dim[1] + i1) * a.dim[2] + i2;

  // incoming gradient (lane = voxel), divided by the view count and zeroed where
  // the forward's clamp blocks it, parked as [channel][voxel]
  float x = 0.f, y = 0.f, z = 0.f;
  bool have = false;
  bool hot = true;  // this lane's voxel holds a gradient in some channel
  if (PLN) {
    const float den = (float)a.V + 1e-6f;
    const int d0 = a.dim[0], d1 = a.dim[1], d2 = a.dim[2];
    // all claims first (unconditional loads, in flight together), then the few gradients
    // they select.  (A blocked claim is odd and matches no index; the sum runs z, y, x: the
    // order autograd accumulates the three torch.max backwards in does not reach fp32 here
    // -- a voxel holding two claims of one channel is rare.)
    const int j0 = inb ? i0 : 0, j1 = inb ? i1 : 0, j2 = inb ? i2 : 0;
    const long long s01 = (long long)d0 * d1, s02 = (long long)d0 * d2, s12 = (long long)d1 * d2;
    const long long p01 = ((long long)b * a.C * d0 + j0) * d1 + j1;
    const long long p02 = ((long long)b * a.C * d0 + j0) * d2 + j2;
    const long long p12 = ((long long)b * a.C * d1 + j1) * d2 + j2;
    // (claims AND gradients unconditionally, all in flight together: one round trip.  The
    // LDS tile limits this kernel to 3 waves per SIMD, so the 6 x CP registers are free.)
    int c01[CP], c02[CP], c12[CP];
    float g01[CP], g02[CP], g12[CP];
#pragma unroll
    for (int ch = 0; ch < CP; ch++) {
      const int cc = ch < a.C ? ch : 0;
      c01[ch] = a.claim[0][p01 + cc * s01];
      c02[ch] = a.claim[1][p02 + cc * s02];
      c12[ch] = a.claim[2][p12 + cc * s12];
      g01[ch] = a.gplane[0][p01 + cc * s01];
      g02[ch] = a.gplane[1][p02 + cc * s02];
      g12[ch] = a.gplane[2][p12 + cc * s12];
    }
    __builtin_amdgcn_sched_barrier(0);
    float gv[CP];
    bool any = false;
#pragma unroll
    for (int ch = 0; ch < CP; ch++) {
      gv[ch] = 0.0f;
      if (inb && ch < a.C) {
        if (c01[ch] == (i2 << 1)) gv[ch] += g01[ch];
        if (c02[ch] == (i1 << 1)) gv[ch] += g02[ch];
        if (c12[ch] == (i0 << 1)) gv[ch] += g12[ch];
        any = any || gv[ch] != 0.0f;
      }
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0) return;  // nothing to send back from this brick
    hot = any;
    for (int i = lane; i < CAP * CP; i += 64) tile[i] = 0.0f;
    have = inb && voxel_world<MODE>(a, b, n, &x, &y, &z);
    if (__builtin_amdgcn_ballot_w64(have) == 0) return;
#pragma unroll
    for (int ch = 0; ch < CP; ch++) st[ch * TSTR + lane] = gv[ch] / den;
  } else {
    const float* gout = a.cubes + (long long)b * a.C * a.N;
    const float den = (float)a.V + 1e-6f;
    // the forward's mask layout: tile n >> 6, word [s * 4 + j], bit vq * L + cq
    // for voxel s * VPS + vq of the tile and channel 4 cq + j
    const int nn = inb ? n : 0;  // lanes outside the volume read voxel 0 and drop it
    const int vs = nn & 63, s = vs / VPS, vq = vs - s * VPS;
    const unsigned long long* mw =
        a.cmask + ((long long)b * ((a.N + kWaveVox - 1) / kWaveVox) + (nn >> 6)) * mask_words(L) +
        s * 4;
    // all CP + 4 loads first (unconditional), so that they overlap
    unsigned long long word[4];
    float gv[CP];
#pragma unroll
    for (int j = 0; j < 4; j++) word[j] = mw[j];
#pragma unroll
    for (int ch = 0; ch < CP; ch++) gv[ch] = gout[(long long)(ch < a.C ? ch : 0) * a.N + nn];
    __builtin_amdgcn_sched_barrier(0);
    // (while the loads are under way: clear the tile, find the voxels' world points)
    for (int i = lane; i < CAP * CP; i += 64) tile[i] = 0.0f;
    have = inb && voxel_world<MODE>(a, b, n, &x, &y, &z);
    if (__builtin_amdgcn_ballot_w64(have) == 0) return;
    __builtin_amdgcn_sched_barrier(0);
    // (a voxel whose gradient is zero in every channel -- blocked by the clamp, or simply
    // zero: what torch.max's backward leaves in two thirds of a volume -- adds nothing and
    // stays out of the footprint box below)
    bool any = false;
#pragma unroll
    for (int ch = 0; ch < CP; ch++) {
      const int cq = ch >> 2, j = ch & 3;
      const bool open = inb && ch < a.C && ((word[j] >> (vq * L + cq)) & 1ull);
      const float g = open ? gv[ch] / den : 0.0f;
      st[ch * TSTR + lane] = g;
      any = any || g != 0.0f;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0) return;
    hot = any;
  }

  // step layout: TP footprint texels x CP channels per instruction
  constexpr int TP = 64 / CP >= 4 ? 4 : 64 / CP;
  constexpr int NP = (4 + TP - 1) / TP;
  const int tq = lane / CP;
  const int c = lane - tq * CP;

#pragma unroll 1
  for (int slot = 0; slot < a.V; slot++) {
    const int view = slot_view(a, slot);
    const TempoCam cam = load_cam(a.cam, frame * a.V + view);
    float* vbase = a.ghm + (long long)frame * a.frame_stride +
                   (long long)view * (a.view_bytes / 4);
    TempoPix p;
    p.px = p.py = 0.f;
    p.inside = 0;
    if (have) p = tempo_world_to_pixel(&cam, x, y, z, UNF);
    float gx, gy, fw, fn;
    int x0, y0;
    tempo_pixel_to_grid(&cam, p.px, p.py, &a.geom, UNF, &gx, &gy);
    tempo_bilinear_frac(gx, gy, a.W, a.H, &x0, &y0, &fw, &fn);
    // a NaN sample point adds nothing (grid_sampler's backward finds none of its
    // texels inside the map)
    const bool ok = have && gx == gx && gy == gy;
    const bool cx[2] = {ok && (unsigned)x0 < (unsigned)a.W, ok && (unsigned)(x0 + 1) < (unsigned)a.W};
    const bool ry[2] = {(unsigned)y0 < (unsigned)a.H, (unsigned)(y0 + 1) < (unsigned)a.H};
    // (fed from the planes, two thirds of a brick's voxels hold no gradient; left out, the
    // footprint box is the hot voxels' alone: fewer bricks that need halves, shorter flushes)
    const bool on = hot && (cx[0] || cx[1]) && (ry[0] || ry[1]);
    const unsigned inmap = (cx[0] && ry[0] ? 1u : 0u) | (cx[1] && ry[0] ? 2u : 0u) |
                           (cx[0] && ry[1] ? 4u : 0u) | (cx[1] && ry[1] ? 8u : 0u);
    // the whole brick first; if its footprint box is larger than the tile, the
    // brick's two halves (2 x 4 x 4 voxels = 32 lanes) one after the other
    int parts = 1;
#pragma unroll 1
    for (int part = 0; part < parts; part++) {
      const bool sel = on && (parts == 1 || (lane >> 5) == part);
      // footprint box (|x0|, |y0| < 2^15: host-checked map size, and the sample
      // grid is clamped to +-1.1)
      const unsigned k0 = wave_umax_to_lane63(sel ? (unsigned)(32767 - x0) : 0u);
      const unsigned k1 = wave_umax_to_lane63(sel ? (unsigned)(32767 - y0) : 0u);
      const unsigned k2 = wave_umax_to_lane63(sel ? (unsigned)(x0 + 32768) : 0u);
      const unsigned k3 = wave_umax_to_lane63(sel ? (unsigned)(y0 + 32768) : 0u);
      const unsigned o2 = (unsigned)__builtin_amdgcn_readlane((int)k2, 63);
      if (o2 == 0u) continue;  // no voxel of this part projects into the view
      const int ox = 32767 - __builtin_amdgcn_readlane((int)k0, 63);
      const int oy = 32767 - __builtin_amdgcn_readlane((int)k1, 63);
      const int bw = (int)o2 - 32768 + 2 - ox;
      const int bh = __builtin_amdgcn_readlane((int)k3, 63) - 32768 + 2 - oy;
      // the tile holds CAP texels in rows of the box's own width, so any box of at
      // most that area fits (oblique views give long, thin boxes)
      const bool fits = bw <= CAP && bh <= CAP && bw * bh <= CAP;  // wave-uniform
      if (!fits && parts == 1) {
        parts = 2;
        part = -1;
        continue;
      }
      const int kbeg = parts == 1 ? 0 : part * 32, kend = parts == 1 ? kWaveVox : kbeg + 32;
      // texel (x0 + i, y0 + j) sits at (base + i + j * stride) * CP, in the tile or in
      // the gradient map; its weight is (j ? fn : 1 - fn) * (i ? fw : 1 - fw)
      const int stride = fits ? bw : a.W;
      const int base = fits ? (y0 - oy) * bw + (x0 - ox) : y0 * a.W + x0;
      s_base[lane] = base * 16 + (int)(sel ? inmap : 0u);
      s_fw[lane] = fw;
      s_fn[lane] = fn;
      int toff[NP];
      float sx[NP], ax[NP], sy[NP], ay[NP];
      unsigned tbit[NP];
#pragma unroll
      for (int pp = 0; pp < NP; pp++) {
        const int t = pp * TP + tq;
        toff[pp] = ((t & 1) + (t >> 1) * stride) * CP + c;
        tbit[pp] = (tq < TP && t < 4) ? 1u << t : 0u;
        sx[pp] = (t & 1) ? 1.0f : -1.0f;  // fma(fw, sx, ax) = fw or 1 - fw, exactly
        ax[pp] = (t & 1) ? 0.0f : 1.0f;
        sy[pp] = (t >> 1) ? 1.0f : -1.0f;
        ay[pp] = (t >> 1) ? 0.0f : 1.0f;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      if (!fits) {
        // still too large (a camera much closer than usual): straight to memory,
        // as in unproject_backward_kernel
        constexpr int VP2 = 64 / CP;  // voxels per step, CP adjacent lanes = one texel's channels
#pragma unroll 4
        for (int v0 = kbeg; v0 < kend; v0 += VP2) {
          const int vsl = v0 + tq;
          if (!(tq < VP2 && vsl < kend)) continue;
          const float g = st[c * TSTR + vsl];
          const int bs = s_base[vsl];
          const float vw = s_fw[vsl], vn = s_fn[vsl];
#pragma unroll
          for (int t = 0; t < 4; t++) {
            if (!((bs >> t) & 1)) continue;
            const float wgt = ((t >> 1) ? vn : 1.0f - vn) * ((t & 1) ? vw : 1.0f - vw);
            const float v = g * wgt;
            if (v != 0.0f)
              unsafeAtomicAdd(vbase + ((bs >> 4) + (t & 1) + (t >> 1) * stride) * CP + c, v);
          }
        }
      } else {
        // accumulate: one voxel per step; operands of KB voxels are read ahead of
        // their read-add-write chain
        constexpr int KB = 8;
#pragma unroll 1
        for (int k0v = kbeg; k0v < kend; k0v += KB) {
          int of[KB][NP];
          float pv[KB][NP];
#pragma unroll
          for (int kk = 0; kk < KB; kk++) {
            const float g = st[c * TSTR + k0v + kk];
            const int bs = s_base[k0v + kk];
            const float vw = s_fw[k0v + kk], vn = s_fn[k0v + kk];
#pragma unroll
            for (int pp = 0; pp < NP; pp++) {
              of[kk][pp] = ((unsigned)bs & tbit[pp]) ? (bs >> 4) * CP + toff[pp] : -1;
              pv[kk][pp] = g * (fmaf(vn, sy[pp], ay[pp]) * fmaf(vw, sx[pp], ax[pp]));
            }
          }
#pragma unroll
          for (int kk = 0; kk < KB; kk++) {
#pragma unroll
            for (int pp = 0; pp < NP; pp++) {
              if (of[kk][pp] >= 0) tile[of[kk][pp]] += pv[kk][pp];
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // flush (and clear) the part of the tile the footprint box reaches: the box's rows
        // follow each other in the tile, so it is swept as one run, four reads in flight
        // (row by row every read was a round trip of its own: 27 instead of 5 per view)
        {
          const int rowlen = bw * CP, total = rowlen * bh;
          const unsigned magic = 0xffffffffu / (unsigned)rowlen + 1u;  // f / rowlen, f < 2^16
          constexpr int FU = 4;
#pragma unroll 1
          for (int f0 = 0; f0 < total; f0 += 64 * FU) {
            float v[FU];
#pragma unroll
            for (int u = 0; u < FU; u++) {
              const int f = f0 + 64 * u + lane;
              v[u] = f < total ? tile[f] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < FU; u++) {
              const int f = f0 + 64 * u + lane;
              if (v[u] != 0.0f) {  // (texels outside the map were never added to)
                tile[f] = 0.0f;
                const int ty = (int)__umulhi((unsigned)f, magic);
                const int rem = f - ty * rowlen;
                unsafeAtomicAdd(vbase + ((long long)(oy + ty) * a.W + ox) * CP + rem, v[u]);
              }
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// channel-last [B][V][HW][CP] -> per-view NCHW slices (inverse of the repack),
// 64-pixel tiles
struct UnrepackArgs {
  float* views[TEMPO_MAX_VIEWS];
  const float* in;
  int V, B, C_total, c_begin, C, HW, CP;
};

__global__ __launch_bounds__(kThreads) void unrepack_kernel(const UnrepackArgs a) {
  __shared__ float s[TEMPO_MAX_CHANNELS][64 + 1];
  const int p0 = blockIdx.x * 64;
  const int v = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* in = a.in + (((long long)b * a.V + v) * a.HW + p0) * a.CP;
  for (int i = tid; i < 64 * a.CP; i += kThreads) {
    const int px = i / a.CP, c = i - px * a.CP;
    if (p0 + px < a.HW) s[c][px] = in[i];
  }
  __syncthreads();
  float* out = a.views[v] + ((long long)b * a.C_total + a.c_begin) * a.HW;
  for (int c = wave; c < a.C; c += 4)
    if (p0 + lane < a.HW) out[(long long)c * a.HW + p0 + lane] = s[c][lane];
}

// ------------------------------------------------------------------ repack
struct RepackArgs {
  const void* views[TEMPO_MAX_VIEWS];  // fp32, or fp16 / bf16 for the DT instantiations
  float* out;
  int V, B, C_total, c_begin, C, HW, CP;
  int b0;  // first frame of the source tensors to take (out is numbered from 0)
};

constexpr int kRepackPix = 64;

// 16-bit maps (mixed-precision heads): DT = TEMPO_DTYPE_F16 / TEMPO_DTYPE_BF16.  The reference
// converts them with `.float()` before sampling (@force_fp32, project_layer.py:65, :242); both
// conversions are exact, so widening inside the repack gives the same texels without the
// intermediate fp32 tensor (a third of the repack's traffic).
template <int DT>
__device__ __forceinline__ float widen16(unsigned short h) {
  if (DT == TEMPO_DTYPE_F16) {
    _Float16 f;
    __builtin_memcpy(&f, &h, 2);
    return (float)f;
  }
  return __uint_as_float((unsigned)h << 16);
}

// Generic tile: 64 pixels, scalar loads (any HW, any alignment).
template <int DT = TEMPO_DTYPE_F32>
__global__ __launch_bounds__(kThreads) void repack_kernel(const RepackArgs a) {
  __shared__ float s[TEMPO_MAX_CHANNELS][kRepackPix + 1];
  const int p0 = blockIdx.x * kRepackPix;
  const int v = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long first = ((long long)(b + a.b0) * a.C_total + a.c_begin) * a.HW;
  const float* in = static_cast<const float*>(a.views[v]) + first;
  const unsigned short* in16 = static_cast<const unsigned short*>(a.views[v]) + first;
  for (int c = wave; c < a.CP; c += 4) {
    float val = 0.0f;
    if (c < a.C && p0 + lane < a.HW) {
      const long long e = (long long)c * a.HW + p0 + lane;
      val = DT == TEMPO_DTYPE_F32 ? in[e] : widen16<DT>(in16[e]);
    }
    s[c][lane] = val;
  }
  __syncthreads();
  const int L = a.CP >> 2;
  float4* out = reinterpret_cast<float4*>(
      a.out + (((long long)b * a.V + v) * a.HW + p0) * a.CP);
  for (int i = tid; i < kRepackPix * L; i += kThreads) {
    const int p = i / L, cq = i - p * L;
    if (p0 + p < a.HW)
      out[i] = make_float4(s[4 * cq][p], s[4 * cq + 1][p], s[4 * cq + 2][p],
                           s[4 * cq + 3][p]);
  }
}

// Wide tile: 256 pixels, 16-byte loads and stores (HW % 4 == 0, 16-byte aligned
// planes).  A wave reads 1 KB of one channel plane per load and keeps CP/4 such
// loads in flight; the LDS tile is channel-major with a row stride of 258 words,
// so the b128 writes are contiguous and the transposed b32 reads of 8 pixels x 4
// channel quads land in 32 different banks.  A block writes 256*CP*4 contiguous
// bytes.
constexpr int kWidePix = 256;
constexpr int kWideStr = kWidePix + 2;

template <int L, int DT = TEMPO_DTYPE_F32>
__device__ __forceinline__ void repack_wide_block(const RepackArgs& a, int bx, int v, int b,
                                                  float* s) {
  const int p0 = bx * kWidePix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long first = ((long long)(b + a.b0) * a.C_total + a.c_begin) * a.HW;
  const float* in = static_cast<const float*>(a.views[v]) + first;
  const unsigned short* in16 = static_cast<const unsigned short*>(a.views[v]) + first;
  const int p = p0 + 4 * lane;
  float4 r[L];
#pragma unroll
  for (int k = 0; k < L; k++) {
    const int c = wave + 4 * k;
    r[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (c < a.C && p < a.HW) {
      // read once: streaming policy, so the planes do not displace the chunk being gathered
      if (DT == TEMPO_DTYPE_F32) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 t = __builtin_nontemporal_load(
            reinterpret_cast<const f32x4*>(in + (long long)c * a.HW + p));
        r[k] = make_float4(t.x, t.y, t.z, t.w);
      } else {
        // 4 pixels = 8 bytes per lane (host-checked: 8-byte aligned planes)
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t t = __builtin_nontemporal_load(
            reinterpret_cast<const u32x2_t*>(in16 + (long long)c * a.HW + p));
        r[k] = make_float4(widen16<DT>((unsigned short)(t.x & 0xffffu)),
                           widen16<DT>((unsigned short)(t.x >> 16)),
                           widen16<DT>((unsigned short)(t.y & 0xffffu)),
                           widen16<DT>((unsigned short)(t.y >> 16)));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < L; k++) {
    // row stride 258 words: rows start 8-byte aligned, so two b64 writes
    float* row = s + (wave + 4 * k) * kWideStr + 4 * lane;
    *reinterpret_cast<float2*>(row) = make_float2(r[k].x, r[k].y);
    *reinterpret_cast<float2*>(row + 2) = make_float2(r[k].z, r[k].w);
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(
      a.out + (((long long)b * a.V + v) * a.HW + p0) * (4 * L));
#pragma unroll
  for (int j = 0; j < L; j++) {
    // item i = tid + 256 j  ->  pixel i / L, channel quad i % L
    const int i = tid + kThreads * j;
    const int px = i / L, cq = i - px * L;
    if (p0 + px < a.HW) {
      const float* col = s + 4 * cq * kWideStr + px;
      out[i] = make_float4(col[0], col[kWideStr], col[2 * kWideStr], col[3 * kWideStr]);
    }
  }
}

template <int L, int DT = TEMPO_DTYPE_F32>
__global__ __launch_bounds__(kThreads) void repack_wide_kernel(const RepackArgs a) {
  __shared__ __attribute__((aligned(16))) float s[4 * L * kWideStr];
  repack_wide_block<L, DT>(a, blockIdx.x, blockIdx.y, blockIdx.z, s);
}

template <int L, int DT = TEMPO_DTYPE_F32>
void launch_repack_wide(const RepackArgs& a, hipStream_t stream) {
  dim3 grid((a.HW + kWidePix - 1) / kWidePix, a.V, a.B);
  hipLaunchKernelGGL((repack_wide_kernel<L, DT>), grid, dim3(kThreads), 0, stream, a);
}

// The root tile kernel with the NEXT chunk's repack as its last workgroups (tempo_root_step):
// workgroups [0, unproj_blocks) unproject chunk c out of one half of the workspace, the
// rest repack chunk c + 1 into the other half.  Workgroups are dispatched in index order, so
// the repack starts when the last unprojection workgroups have been handed out and fills the
// launch's tail (a quarter of the wave slots idle on average over a launch otherwise) instead
// of paying a ramp and a tail of its own.  No dependency between the two roles inside a
// launch.  Results are those of the separate launches.
template <int L, int SB, int UNF>
__global__ __launch_bounds__(kThreads) void unproject_repack_kernel(const UnprojArgs a,
                                                                    const RepackArgs r,
                                                                    int unproj_blocks,
                                                                    int pix_tiles) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int bid = blockIdx.x;
  if (bid < unproj_blocks) {
    unproject_block<MODE_ROOT, L, SB, UNF>(a, bid, smem);
    return;
  }
  const int t = bid - unproj_blocks;
  const int bx = t % pix_tiles;
  const int vb = t / pix_tiles;
  repack_wide_block<L>(r, bx, vb % r.V, vb / r.V, reinterpret_cast<float*>(smem));
}

constexpr long long kFuseBytes = 220ll << 20;  // both halves of a fused step's workspace

constexpr int fused_lds_bytes(int L) {
  return kTileWaves * wave_lds_bytes(L) > 4 * L * kWideStr * 4 ? kTileWaves * wave_lds_bytes(L)
                                                                : 4 * L * kWideStr * 4;
}

// The inverse with the same tile: 256 * CP * 4 contiguous bytes in (16-byte
// loads, transposed b32 writes into the channel-major LDS tile), 1 KB runs of one
// channel plane out.  Same conditions as the wide repack.
template <int L>
__global__ __launch_bounds__(kThreads) void unrepack_wide_kernel(const UnrepackArgs a) {
  __shared__ __attribute__((aligned(16))) float s[4 * L * kWideStr];
  const int p0 = blockIdx.x * kWidePix;
  const int v = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* in = reinterpret_cast<const float4*>(
      a.in + (((long long)b * a.V + v) * a.HW + p0) * (4 * L));
  float4 r[L];
#pragma unroll
  for (int j = 0; j < L; j++) {
    const int i = tid + kThreads * j;
    r[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p0 + i / L < a.HW) r[j] = in[i];
  }
#pragma unroll
  for (int j = 0; j < L; j++) {
    const int i = tid + kThreads * j;
    const int px = i / L, cq = i - px * L;
    float* col = s + 4 * cq * kWideStr + px;
    col[0] = r[j].x;
    col[kWideStr] = r[j].y;
    col[2 * kWideStr] = r[j].z;
    col[3 * kWideStr] = r[j].w;
  }
  __syncthreads();
  float* out = a.views[v] + ((long long)b * a.C_total + a.c_begin) * a.HW;
  const int p = p0 + 4 * lane;
#pragma unroll
  for (int k = 0; k < L; k++) {
    const int c = wave + 4 * k;
    if (c < a.C && p < a.HW) {
      const float* row = s + c * kWideStr + 4 * lane;
      const float2 lo = *reinterpret_cast<const float2*>(row);
      const float2 hi = *reinterpret_cast<const float2*>(row + 2);
      *reinterpret_cast<float4*>(out + (long long)c * a.HW + p) =
          make_float4(lo.x, lo.y, hi.x, hi.y);
    }
  }
}

template <int L>
void launch_unrepack_wide(const UnrepackArgs& a, hipStream_t stream) {
  dim3 grid((a.HW + kWidePix - 1) / kWidePix, a.V, a.B);
  hipLaunchKernelGGL((unrepack_wide_kernel<L>), grid, dim3(kThreads), 0, stream, a);
}

// ------------------------------------------- 1x1 convolution, channel-last out
// The 2-D head's last layer in front of the unprojection is a 1x1 convolution
// (rnn_multiview_pose.py:61-66, :321: process_layer = nn.Conv2d(256, num_features, 1) on the
// backbone map of all T*V*B images, then chunked into feature_maps[t][v]).  This kernel is
// that layer with the unprojection's own input layout as its output: out[frame][view][pixel]
// [CP channels], frames ordered (t, b), so no NCHW tensor is written and no repack reads it
// back.  A [pixels x K] x [K x channels] contraction per image: f32-input MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains), A = weights (channels x k), B = input
// (k x pixels).  One wave owns 64 consecutive pixels of one image; per k-step of 4 planes a
// lane loads 4 consecutive pixels of plane k0 + lane / 16 (16 lanes = 256 contiguous bytes
// per plane), which are the B operands of four MFMAs whose column j stands for pixel
// 4 j + t.  The accumulators then hold, per lane, 4 consecutive channels of one pixel: one
// 16-byte store each, 64 contiguous bytes (CP = 16) per pixel.  HBM-bound: the input is
// K * 4 bytes per pixel, read once.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;      // [T*V*B][K][HW]
  const float* weight;  // [C][K]
  const float* bias;    // [C] or nullptr
  float* out;           // [T*B][V][HW][CP]
  int V, B, K, C, CP, HW;
  int images;           // T * V * B, ordered (t, v, b)
  int tiles;            // 64-pixel tiles per image
  int vec;              // HW % 4 == 0 and 16-byte aligned input: 128-bit loads
};

#ifndef TEMPO_CONV_STEP
#define TEMPO_CONV_STEP 8
#endif
constexpr int kConvStep = TEMPO_CONV_STEP;  // k-steps (of 4 planes) whose loads are in flight together

template <int NT>  // channel tiles of 16: CP = 16 NT
__global__ __launch_bounds__(kThreads) void conv1x1_nhwc_kernel(const ConvArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);  // [K][16 NT], zero rows for channels >= C
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < a.K * 16 * NT; i += kThreads) {
    const int k = i / (16 * NT), c = i - k * (16 * NT);
    wl[i] = c < a.C ? a.weight[(long long)c * a.K + k] : 0.0f;
  }
  __syncthreads();
  const long long item = (long long)blockIdx.x * (kThreads / 64) + wave;
  const int img = (int)(item / a.tiles);
  if (img >= a.images) return;
  const int p0 = (int)(item - (long long)img * a.tiles) * 64;
  // image m = (t * V + v) * B + b  ->  frame t * B + b, view v
  const int t_ = img / (a.V * a.B), v_ = (img / a.B) % a.V, b_ = img % a.B;
  const float* src = a.in + (long long)img * a.K * a.HW;
  const int kq = lane >> 4, jj = lane & 15;
  const int px = p0 + 4 * jj;  // this lane's four pixels: px .. px + 3

  auto load = [&](int k0) -> f32x4 {
    const float* q = src + (long long)(k0 + kq) * a.HW + px;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (k0 + kq < a.K) {
      if (a.vec) {
        // (read once, 5 GB at config 2's shape: streaming policy, so the maps being
        // written stay cached for the unprojection that follows)
        if (px < a.HW) x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (px + e < a.HW) x[e] = q[e];
      }
    }
    return x;
  };

  f32x4 acc[NT][4];
#pragma unroll
  for (int n = 0; n < NT; n++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[n][e] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 cur[kConvStep], nxt[kConvStep];
#pragma unroll
  for (int u = 0; u < kConvStep; u++) cur[u] = load(4 * u);
#pragma unroll 1
  for (int k0 = 0; k0 < a.K; k0 += 4 * kConvStep) {
#pragma unroll
    for (int u = 0; u < kConvStep; u++) nxt[u] = load(k0 + 4 * (kConvStep + u));
#pragma unroll
    for (int u = 0; u < kConvStep; u++) {
      const int k = k0 + 4 * u + kq;
#pragma unroll
      for (int n = 0; n < NT; n++) {
        const float w = k < a.K ? wl[k * (16 * NT) + 16 * n + jj] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++)
          acc[n][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, cur[u][e], acc[n][e], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < kConvStep; u++) cur[u] = nxt[u];
  }

  // lane: channels 16 n + 4 kq .. + 3 of pixel px + e
  float* dst = a.out + (((long long)(t_ * a.B + b_) * a.V + v_) * a.HW) * a.CP;
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const int c0 = 16 * n + 4 * kq;
    if (c0 >= a.CP) continue;
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (c0 + r < a.C) bs[r] = a.bias[c0 + r];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (px + e < a.HW)
        *reinterpret_cast<f32x4*>(dst + (long long)(px + e) * a.CP + c0) = acc[n][e] + bs;
    }
  }
}

// ------------------------------------------------------------ z-column gather
// The root detector's second stage reads, for each of its K best (x, y) cells, the
// z-column of every channel (tempo_detector.py:204-206: torch.gather on the flattened,
// permuted volume with a [B,K,C,Z] index tensor).  One workgroup per (frame, cell).
__global__ __launch_bounds__(kThreads) void zcolumns_kernel(const float* __restrict__ cubes,
                                                            const long long* __restrict__ index,
                                                            float* __restrict__ out, int K,
                                                            int C, long long XY, int Z) {
  const int bk = blockIdx.x;
  const int b = bk / K;
  const long long cell = index[bk];
  const bool ok = cell >= 0 && cell < XY;
  const float* src = cubes + ((long long)b * C * XY + (ok ? cell : 0)) * Z;
  float* dst = out + (long long)bk * C * Z;
  for (int i = threadIdx.x; i < C * Z; i += kThreads) {
    const int c = i / Z, z = i - c * Z;
    dst[i] = ok ? src[(long long)c * XY * Z + z] : 0.0f;
  }
}

// ------------------------------------------------------------ z-slab assembly
// Multi-GPU dense grids: rank r computes planes [r * Zr, (r + 1) * Zr) of every (b, c, x, y)
// column, one all-gather leaves [R][rows][Zr] on every rank (rows = B * C * X * Y), and the
// volume [rows][R * Zr] is that with the rank index moved inside the row.  One thread per
// output quad (or float when Zr is not a multiple of 4); the gathered buffer is read once
// (streaming loads), the writes are contiguous.
template <int VEC>
__global__ __launch_bounds__(kThreads) void assemble_zslabs_kernel(const float* __restrict__ in,
                                                                   float* __restrict__ out,
                                                                   int R, long long rows,
                                                                   int Zr) {
  const int per_row = R * Zr / VEC;  // output elements (of VEC floats) per row
  const long long total = rows * per_row;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long row = i / per_row;
    const int q = (int)(i - row * per_row) * VEC;
    const int r = q / Zr, zr = q - r * Zr;
    const float* src = in + ((long long)r * rows + row) * Zr + zr;
    if (VEC == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src));
      *reinterpret_cast<f4*>(out + i * 4) = v;
    } else {
      out[i] = __builtin_nontemporal_load(src);
    }
  }
}

// ---------------------------------------------------------- window geometry
// project_layer.py:299-324 for all windows of a call in one launch (the reference spends
// about fifteen tiny tensor ops on it, per candidate and timestep).  One thread per window;
// every fp32 step is the separately rounded torch op of the reference (this file is built
// with -ffp-contract=off; divisions are correctly rounded), so the integers and the offset
// are bit-identical to the tensor-op form (tests/test_gpu_layouts.py).
struct GeomArgs {
  const float* gc;     // [B][cols] proposals (x, y, z, id, conf, bbox_w, bbox_h)
  const float* scale;  // [3] (fine - 1) / space          (set_space_size)
  const float* bias;   // [3]
  const float* space;  // [3] whole_space_size
  const float* sub;    // [3] ind_space_size
  int fine[3], vox[3];
  int B, cols;
  int32_t* tl;
  int32_t* start;
  int32_t* end;
  float* offset;
  uint8_t* live;  // or nullptr
};

__global__ __launch_bounds__(kThreads) void window_geometry_kernel(const GeomArgs a) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= a.B) return;
  const float* g = a.gc + (long long)b * a.cols;
  int tl[3], mask[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float t = g[i] * a.scale[i];
    t = t + a.bias[i];
    tl[i] = (int)rintf(t);  // torch.round: half to even
    float o = (float)tl[i] / (float)(a.fine[i] - 1);
    o = o * a.space[i];
    o = o - a.space[i] / 2.0f;
    o = o + a.sub[i] / 2.0f;
    a.tl[3 * b + i] = tl[i];
    a.offset[3 * b + i] = o;
  }
  if (a.cols >= 7) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      float m = 1.0f - g[5 + j];
      m = m / 2.0f;
      m = m * (float)(a.vox[j] - 1);
      const int mi = (int)m;  // .int(): towards zero
      mask[j] = mi < 0 ? 0 : mi;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int lo = tl[i] + mask[i];
    const int st = lo >= 0 ? lo : 0;
    const int hi = tl[i] + a.vox[i] - mask[i];
    int en = hi <= a.fine[i] ? hi : a.fine[i];
    en = en > st + 1 ? en : st + 1;
    a.start[3 * b + i] = st;
    a.end[3 * b + i] = en;
  }
  if (a.live != nullptr) a.live[b] = (a.cols > 3 && !(g[3] >= 0.0f)) ? 0 : 1;
}

// ---------------------------------------------------------- project points
__global__ __launch_bounds__(kThreads) void project_points_kernel(
    const float* __restrict__ pts, const float* __restrict__ cam, int N,
    const TempoGeom geom, int unfused, float* __restrict__ pix,
    float* __restrict__ grid, uint8_t* __restrict__ inside) {
  const int m = blockIdx.y;
  const int n = blockIdx.x * kThreads + threadIdx.x;
  if (n >= N) return;
  const TempoCam c = load_cam(cam, m);
  const TempoPix p =
      tempo_world_to_pixel(&c, pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], unfused);
  float gx, gy;
  tempo_pixel_to_grid(&c, p.px, p.py, &geom, unfused, &gx, &gy);
  const long long o = (long long)m * N + n;
  pix[2 * o] = p.px;
  pix[2 * o + 1] = p.py;
  grid[2 * o] = gx;
  grid[2 * o + 1] = gy;
  inside[o] = (uint8_t)p.inside;
}

// ------------------------------------------------------------ host helpers
template <int MODE, int L, int UNF>
int launch_one(const UnprojArgs& a, hipStream_t stream, const RepackArgs* tail = nullptr) {
  // sub-steps whose fetches are in flight together (8 = all of a view's)
  // (wider texels: two sub-steps keep the fetch registers at 32 and the occupancy up;
  // 20 channels 0.38 -> 0.31 ms, 32 channels 0.53 -> 0.49 ms at config 2's shape)
  // (window, 28-32 channels: two sub-steps, 8.07 -> 7.58 ms at config 3's shape)
#ifndef TEMPO_SB_ROOT_WIDE
#define TEMPO_SB_ROOT_WIDE 2
#endif
#ifndef TEMPO_SB_WINDOW_WIDE
#define TEMPO_SB_WINDOW_WIDE 2
#endif
  constexpr int SB = MODE == MODE_ROOT ? (L == 4 ? TEMPO_SB_ROOT : TEMPO_SB_ROOT_WIDE)
                                       : (L >= 7 ? TEMPO_SB_WINDOW_WIDE : TEMPO_SB_WINDOW);
  UnprojArgs k = a;
  fill_order_magic(k.order, k.magic, a.V, a.dim);
  const int waves = kThreads / 64;
  k.wave_vox = kWaveVox;
  k.tile_vox = waves * kWaveVox;
  k.tiles = (a.N + k.tile_vox - 1) / k.tile_vox;
  const bool planes = a.pl01 != nullptr || a.pl02 != nullptr || a.pl12 != nullptr;
  if (MODE == MODE_WINDOW && planes && a.dim[2] == kWaveVox && a.dim[1] % kPlaneWaves == 0 &&
      plane_lds_bytes(L) <= 65536) {
    const long long wtotal = (long long)a.NB * (a.dim[1] / kPlaneWaves) *
                             ((a.dim[0] + kPlaneXR - 1) / kPlaneXR);
    if (wtotal > 0x7fffffffLL / 8) return TEMPO_E_DIM;
    k.total = (int)wtotal;
    k.chunk = (k.total + 7) / 8;
    hipLaunchKernelGGL((unproject_planes_kernel<L, UNF>), dim3(8 * k.chunk),
                       dim3(kPlaneWaves * 64), (size_t)plane_lds_bytes(L), stream, k);
    return (int)hipGetLastError();
  }
  // grouped launch when the records of all views fit the default LDS limit;
  // otherwise the group table is only a hint and every item projects for itself
  if (a.groups != nullptr && waves * group_wave_lds(a.V, L) <= 65536) {
    const long long gtotal = (long long)a.NG * k.tiles;
    if (gtotal > 0x7fffffffLL / 8) return TEMPO_E_DIM;
    k.total = (int)gtotal;
    k.chunk = (k.total + 7) / 8;
    k.wave_lds = group_wave_lds(a.V, L);
    hipLaunchKernelGGL((unproject_group_kernel<MODE, L, UNF>), dim3(8 * k.chunk),
                       dim3(kThreads), (size_t)waves * k.wave_lds, stream, k);
    return (int)hipGetLastError();
  }
  // root z-max (the only plane asked for): waves of whole z-columns, so that every column's
  // maximum is complete inside one wave and is stored instead of max-accumulated with
  // atomics (config 2: 60 of 64 lanes busy; the fused z-max then costs +0.03 instead of
  // +0.07 ms per step).  Full waves in workgroups of 5 (320 voxels = 16 whole columns, the
  // maxima read across the waves' tiles after one workgroup barrier) were measured at
  // 0.261 against 0.200 ms: the barrier takes away the waves' autonomy, which costs far more
  // than the 4 idle lanes.
  if (MODE == MODE_ROOT && a.pl01 != nullptr && a.pl02 == nullptr && a.pl12 == nullptr &&
      a.cmask == nullptr) {
    k.wave_vox = zmax_wave_vox(a.dim[2]);
  }
  k.tile_vox = kTileWaves * k.wave_vox;
  k.tiles = (a.N + k.tile_vox - 1) / k.tile_vox;
  const long long total = (long long)a.NB * k.tiles;
  if (total > 0x7fffffffLL / 8) return TEMPO_E_DIM;
  k.total = (int)total;
  k.chunk = (k.total + 7) / 8;
  if (MODE == MODE_ROOT && tail != nullptr && kTileWaves * 64 == kThreads) {
    // the next chunk's repack rides on this launch (unproject_repack_kernel)
    const int pix_tiles = (tail->HW + kWidePix - 1) / kWidePix;
    const long long rb = (long long)pix_tiles * tail->V * tail->B;
    if (8LL * k.chunk + rb > 0x7fffffffLL) return TEMPO_E_DIM;
    hipLaunchKernelGGL((unproject_repack_kernel<L, SB, UNF>), dim3((unsigned)(8 * k.chunk + rb)),
                       dim3(kThreads), (size_t)fused_lds_bytes(L), stream, k, *tail,
                       8 * k.chunk, pix_tiles);
    return (int)hipGetLastError();
  }
#ifndef TEMPO_TILE_PAD_LDS  // measurement builds: extra LDS per workgroup caps the occupancy
#define TEMPO_TILE_PAD_LDS 0
#endif
  hipLaunchKernelGGL((unproject_kernel<MODE, L, SB, UNF>), dim3(8 * k.chunk),
                     dim3(kTileWaves * 64),
                     (size_t)kTileWaves * wave_lds_bytes(L) + TEMPO_TILE_PAD_LDS, stream, k);
  return (int)hipGetLastError();
}

template <int MODE, int L, int UNF>
int launch_backward_one(const UnprojArgs& a, hipStream_t stream) {
  UnprojArgs k = a;
  fill_order_magic(k.order, k.magic, a.V, a.dim);
  const int waves = kThreads / 64;
  k.tile_vox = waves * kWaveVox;
  k.tiles = (a.N + k.tile_vox - 1) / k.tile_vox;
  const long long total = (long long)a.NB * k.tiles;
  if (total > 0x7fffffffLL / 8) return TEMPO_E_DIM;
  k.total = (int)total;
  k.chunk = (k.total + 7) / 8;
  // the footprint records of a launch's views live in LDS (64 KB per workgroup): more
  // views than fit (11 at CP = 16, 7 at CP = 32) go through in several launches, each of
  // which projects every view (for the denominator) and scatters its own subset
  int per = a.V;
  while (per > 1 && waves * backward_wave_lds(per, L) > 65536) per--;
  if (waves * backward_wave_lds(per, L) > 65536) return TEMPO_E_DIM;
  for (int s0 = 0; s0 < a.V; s0 += per) {
    k.slot0 = s0;
    k.slot1 = s0 + per < a.V ? s0 + per : a.V;
    k.wave_lds = backward_wave_lds(k.slot1 - k.slot0, L);
    hipLaunchKernelGGL((unproject_backward_kernel<MODE, L, UNF>), dim3(8 * k.chunk),
                       dim3(kThreads), (size_t)waves * k.wave_lds, stream, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return TEMPO_OK;
}

// window backward through the LDS tile (window_backward_brick_kernel)
template <int L, int UNF, int CAP, int PLN = 0>
int launch_backward_brick(const UnprojArgs& a, hipStream_t stream) {
  UnprojArgs k = a;
  fill_order_magic(k.order, k.magic, a.V, a.dim);
  const long long bricks = (long long)((a.dim[0] + kBrickEdge - 1) / kBrickEdge) *
                           ((a.dim[1] + kBrickEdge - 1) / kBrickEdge) *
                           ((a.dim[2] + kBrickEdge - 1) / kBrickEdge);
  const long long total = (long long)a.NB * bricks;
  if (total > 0x7fffffffLL / 8) return TEMPO_E_DIM;
  k.tile_vox = kWaveVox;
  k.tiles = (int)bricks;
  k.total = (int)total;
  k.chunk = (k.total + 7) / 8;
  hipLaunchKernelGGL((window_backward_brick_kernel<L, UNF, CAP, PLN>), dim3(8 * k.chunk), dim3(64),
                     (size_t)brick_lds(L, CAP).total, stream, k);
  return (int)hipGetLastError();
}

// the planes-gradient backward (window_backward_brick_kernel<.., PLN = 1>): the brick kernel
// only -- there is no gradient volume the per-sample kernel could read.  Its footprint boxes
// are those of a brick's hot voxels alone, so smaller tiles pay (more waves per CU): config 3,
// ms per 320 windows, tile of 64 / 81 / 100 / 121 texels: 11.0 / 10.8 / 11.9 / 13.4.
constexpr int kBrickCapPlanes = 81;
template <int L, int UNF>
int launch_backward_planes_l(const UnprojArgs& a, hipStream_t stream, int cap) {
  switch (cap) {
    case kBrickCapPlanes: return launch_backward_brick<L, UNF, kBrickCapPlanes, 1>(a, stream);
    case kBrickCapSmall: return launch_backward_brick<L, UNF, kBrickCapSmall, 1>(a, stream);
    case kBrickCapLarge: return launch_backward_brick<L, UNF, kBrickCapLarge, 1>(a, stream);
    default: return TEMPO_E_ARG;
  }
}

int launch_backward_planes(const UnprojArgs& a, int mm_unfused, hipStream_t stream, int tile) {
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  if (a.W > 16384 || a.H > 16384) return TEMPO_E_SIZE;  // 16-bit footprint boxes
  const int cap = tile == 0 ? kBrickCapPlanes : tile;
#define TEMPO_PBRICK_CASE(LL)                                                    \
  case LL:                                                                       \
    return mm_unfused ? launch_backward_planes_l<LL, 1>(a, stream, cap)          \
                      : launch_backward_planes_l<LL, 0>(a, stream, cap);
  switch (a.CP / 4) {
    TEMPO_PBRICK_CASE(4)
    TEMPO_PBRICK_CASE(5)
    TEMPO_PBRICK_CASE(6)
    TEMPO_PBRICK_CASE(7)
    TEMPO_PBRICK_CASE(8)
    default: return TEMPO_E_PAD;
  }
#undef TEMPO_PBRICK_CASE
}

template <int L, int UNF>
int launch_claims_one(const UnprojArgs& a, hipStream_t stream) {
  UnprojArgs k = a;
  fill_order_magic(k.order, k.magic, a.V, a.dim);
  k.wave_vox = kWaveVox;
  k.tile_vox = kTileWaves * kWaveVox;
  k.tiles = (a.N + k.tile_vox - 1) / k.tile_vox;
  const long long total = (long long)a.NB * k.tiles;
  if (total > 0x7fffffffLL / 8) return TEMPO_E_DIM;
  k.total = (int)total;
  k.chunk = (k.total + 7) / 8;
  hipLaunchKernelGGL((window_claim_kernel<L, UNF>), dim3(8 * k.chunk), dim3(kTileWaves * 64),
                     (size_t)kTileWaves * wave_lds_bytes(L), stream, k);
  return (int)hipGetLastError();
}

int launch_claims(const UnprojArgs& a, int mm_unfused, hipStream_t stream) {
#define TEMPO_CLAIM_CASE(LL)                                                    \
  case LL:                                                                      \
    return mm_unfused ? launch_claims_one<LL, 1>(a, stream) : launch_claims_one<LL, 0>(a, stream);
  switch (a.CP / 4) {
    TEMPO_CLAIM_CASE(4)
    TEMPO_CLAIM_CASE(5)
    TEMPO_CLAIM_CASE(6)
    TEMPO_CLAIM_CASE(7)
    TEMPO_CLAIM_CASE(8)
    default: return TEMPO_E_PAD;
  }
#undef TEMPO_CLAIM_CASE
}

template <int MODE>
int launch_backward(const UnprojArgs& a, int mm_unfused, hipStream_t stream, int tile = 0) {
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  if (tile != 0 && tile != kBrickCapSmall && tile != kBrickCapLarge) return TEMPO_E_ARG;
  const bool large = tile == kBrickCapLarge;
#ifndef TEMPO_NO_BRICK
  // the footprint box is kept in 16-bit texel coordinates
  if (MODE == MODE_WINDOW && a.W <= 16384 && a.H <= 16384) {
#define TEMPO_BRICK_CASE(LL)                                                            \
  case LL:                                                                              \
    if (large)                                                                          \
      return mm_unfused ? launch_backward_brick<LL, 1, kBrickCapLarge>(a, stream)       \
                        : launch_backward_brick<LL, 0, kBrickCapLarge>(a, stream);      \
    return mm_unfused ? launch_backward_brick<LL, 1, kBrickCapSmall>(a, stream)         \
                      : launch_backward_brick<LL, 0, kBrickCapSmall>(a, stream);
    switch (a.CP / 4) {
      TEMPO_BRICK_CASE(4)
      TEMPO_BRICK_CASE(5)
      TEMPO_BRICK_CASE(6)
      TEMPO_BRICK_CASE(7)
      TEMPO_BRICK_CASE(8)
      default: return TEMPO_E_PAD;
    }
#undef TEMPO_BRICK_CASE
  }
#endif
#define TEMPO_BWD_CASE(LL)                                          \
  case LL:                                                          \
    return mm_unfused ? launch_backward_one<MODE, LL, 1>(a, stream) \
                      : launch_backward_one<MODE, LL, 0>(a, stream);
  switch (a.CP / 4) {
    TEMPO_BWD_CASE(4)
    TEMPO_BWD_CASE(5)
    TEMPO_BWD_CASE(6)
    TEMPO_BWD_CASE(7)
    TEMPO_BWD_CASE(8)
    default: return TEMPO_E_PAD;
  }
#undef TEMPO_BWD_CASE
}

template <int MODE, int UNF>
int launch_l(const UnprojArgs& a, hipStream_t stream, const RepackArgs* tail) {
  switch (a.CP / 4) {
    case 4: return launch_one<MODE, 4, UNF>(a, stream, tail);
    case 5: return launch_one<MODE, 5, UNF>(a, stream, tail);
    case 6: return launch_one<MODE, 6, UNF>(a, stream, tail);
    case 7: return launch_one<MODE, 7, UNF>(a, stream, tail);
    case 8: return launch_one<MODE, 8, UNF>(a, stream, tail);
    default: return TEMPO_E_PAD;
  }
}

template <int MODE>
int launch_mode(const UnprojArgs& a, int mm_unfused, hipStream_t stream,
                const RepackArgs* tail = nullptr) {
  // the three matmul sites share one flavour on every host measured so far;
  // mixed flavours are not compiled in
  if (mm_unfused == 0) return launch_l<MODE, 0>(a, stream, tail);
  if (mm_unfused == TEMPO_UNFUSED_ALL) return launch_l<MODE, 1>(a, stream, tail);
  return TEMPO_E_ARG;
}

template <int L, int UNF>
int launch_columns_one(const UnprojArgs& a, hipStream_t stream) {
  UnprojArgs k = a;
  fill_order_magic(k.order, k.magic, a.V, a.dim);
  k.wave_vox = 16;
  k.tile_vox = kTileWaves * k.wave_vox;
  k.tiles = (a.N + k.tile_vox - 1) / k.tile_vox;
  const long long total = (long long)a.NB * k.tiles;
  if (total > 0x7fffffffLL) return TEMPO_E_DIM;
  k.total = (int)total;
  hipLaunchKernelGGL((unproject_columns_kernel<L, UNF>), dim3((unsigned)k.total),
                     dim3(kTileWaves * 64), (size_t)kTileWaves * wave_lds_bytes(L), stream, k);
  return (int)hipGetLastError();
}

int launch_columns(const UnprojArgs& a, int mm_unfused, hipStream_t stream) {
#define TEMPO_COL_CASE(LL)                                                       \
  case LL:                                                                       \
    return mm_unfused ? launch_columns_one<LL, 1>(a, stream) : launch_columns_one<LL, 0>(a, stream);
  switch (a.CP / 4) {
    TEMPO_COL_CASE(4)
    TEMPO_COL_CASE(5)
    TEMPO_COL_CASE(6)
    TEMPO_COL_CASE(7)
    TEMPO_COL_CASE(8)
    default: return TEMPO_E_PAD;
  }
#undef TEMPO_COL_CASE
}

// The fused window planes of a call whose texels are 32 channels wide, as two launches of 16
// channels each (the same texels, the second group 64 bytes further on; outputs of group g
// start 16 g channels further on and keep the full channel count as their stride,
// UnprojArgs::Cs).  The 32-channel instantiation of the column walk needs 175 VGPRs (2 waves
// per SIMD) and 66 KB of LDS per workgroup; config 3's shape at 32 channels, planes only:
// 16.8 ms in one launch, 9.8 ms in two.  Per-voxel results do not depend on the grouping
// (channels are independent).  The VOLUME kernels stay with one 32-channel launch: a 128-byte
// texel is exactly one cache line, and two 16-channel launches fetch every line twice
// (measured: root 0.369 -> 0.631 ms of kernel time per step, window 6.41 -> 8.35 ms).
template <int MODE>
int launch_forward(const UnprojArgs& a, int mm_unfused, hipStream_t stream,
                   const RepackArgs* tail = nullptr) {
#ifndef TEMPO_NO_CHANNEL_GROUPS
  const bool planes = a.pl01 != nullptr || a.pl02 != nullptr || a.pl12 != nullptr;
  if (MODE == MODE_WINDOW && planes && a.CP == 32 && a.dim[2] == kWaveVox &&
      a.dim[1] % kPlaneWaves == 0 && a.groups == nullptr && a.cmask == nullptr &&
      a.ghm == nullptr) {
    for (int g = 0; g < 2; g++) {
      const int cs = a.C - 16 * g < 16 ? a.C - 16 * g : 16;
      if (cs <= 0) break;
      UnprojArgs k = a;
      k.C = cs;
      k.CP = 16;
      for (int v = 0; v < a.V; v++) k.vbase[v] = a.vbase[v] + 64 * g;
      k.map_bytes = (unsigned)((long long)a.H * a.row_bytes - a.texel_bytes + 4ll * cs);
      const long long n = a.N;
      if (k.cubes != nullptr) k.cubes += 16ll * g * n;
      if (k.pl01 != nullptr) k.pl01 += 16ll * g * a.dim[0] * a.dim[1];
      if (k.pl02 != nullptr) k.pl02 += 16ll * g * a.dim[0] * a.dim[2];
      if (k.pl12 != nullptr) k.pl12 += 16ll * g * a.dim[1] * a.dim[2];
      if (g > 0) {  // the channel-independent outputs come from the first group
        k.grids = nullptr;
        k.mask = nullptr;
        k.mask_bytes = 0;
      }
      const bool last = a.C - 16 * (g + 1) <= 0;
      const int rc = launch_mode<MODE>(k, mm_unfused, stream, last ? tail : nullptr);
      if (rc != TEMPO_OK) return rc;
    }
    return TEMPO_OK;
  }
#endif
  return launch_mode<MODE>(a, mm_unfused, stream, tail);
}

// 16-byte alignment of a device pointer the kernels access with 128-bit loads / stores
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_common(int V, int C, int CP, int H, int W) {
  if (V <= 0 || V > TEMPO_MAX_VIEWS || C <= 0 || C > TEMPO_MAX_CHANNELS || H <= 0 ||
      W <= 0)
    return TEMPO_E_DIM;
  if (CP % 4 != 0 || CP < C || CP < 16 || CP > TEMPO_MAX_CHANNELS) return TEMPO_E_PAD;
  if ((long long)H * W * CP * 4 >= 0x80000000LL) return TEMPO_E_SIZE;
  return TEMPO_OK;
}

struct Extra {
  const RepackArgs* tail = nullptr;     // root forward: the next chunk's repack rides along
  float* zmax = nullptr;                // root forward: max over z, [B][C][X][Y]
  unsigned long long* cmask = nullptr;  // forward: record / backward: read the clamp mask
  float* ghm = nullptr;                 // backward: gradient of the heatmaps (accumulated)
  int tile = 0;                         // window backward: texels per LDS tile (0 = default)
  // trainable window planes: claims [xy, xz, yz] (written when `planes_in`, else read by the
  // backward together with the planes' gradients)
  int32_t* claim[3] = {nullptr, nullptr, nullptr};
  const float* gplane[3] = {nullptr, nullptr, nullptr};
  bool planes_in = false;               // window_impl: the planes are INPUTS (claim pass)
};

// Where the heatmaps are: `hm` = the repacked [F][V][H][W][CP] buffer (all kernels), or
// `views` = V channel-last tensors [F][H][W][texel_bytes / 4] (forward kernels only).
struct Layout {
  const float* hm = nullptr;
  const void* const* views = nullptr;
  long long frame_bytes = 0;
  int texel_bytes = 0;
  // planar: `views` are V NCHW tensors [F][C_total][H][W], channels [c_begin, c_begin + C)
  // (the columns kernel only)
  bool planar = false;
  int C_total = 0, c_begin = 0;
};

// fills the forward kernels' layout fields of `a` (a.V, a.H, a.W, a.CP set); TEMPO_OK or an
// error code
int apply_layout(UnprojArgs& a, const Layout& lay) {
  for (int v = 0; v < TEMPO_MAX_VIEWS; v++) a.vbase[v] = nullptr;
  if (lay.hm != nullptr) {
    if (!aligned16(lay.hm)) return TEMPO_E_ARG;  // texels are fetched with 128-bit loads
    a.hm = lay.hm;
    a.texel_bytes = a.CP * 4;
    a.row_bytes = a.W * a.texel_bytes;
    a.frame_bytes = (long long)a.V * a.H * a.row_bytes;
    for (int v = 0; v < a.V; v++)
      a.vbase[v] = reinterpret_cast<const char*>(lay.hm) + (long long)v * a.H * a.row_bytes;
  } else if (lay.planar) {
    if (lay.views == nullptr) return TEMPO_E_NULL;
    if (lay.c_begin < 0 || lay.c_begin + a.C > lay.C_total) return TEMPO_E_DIM;
    const long long plane = (long long)a.H * a.W * 4;
    if (plane * a.C >= 0x80000000LL) return TEMPO_E_SIZE;
    a.texel_bytes = 4;
    a.row_bytes = a.W * 4;
    a.plane_bytes = (unsigned)plane;
    a.frame_bytes = plane * lay.C_total;
    for (int v = 0; v < a.V; v++) {
      if (lay.views[v] == nullptr) return TEMPO_E_NULL;
      if (reinterpret_cast<uintptr_t>(lay.views[v]) % 4 != 0) return TEMPO_E_ARG;
      a.vbase[v] = static_cast<const char*>(lay.views[v]) + plane * lay.c_begin;
    }
    if (a.row_bytes >= (1 << 23) || a.H >= (1 << 23)) return TEMPO_E_SIZE;
    a.map_bytes = (unsigned)(plane * a.C);  // the slice's C planes, nothing beyond
    return TEMPO_OK;
  } else {
    if (lay.views == nullptr) return TEMPO_E_NULL;
    // texels may be narrower than the CP channels a fetch covers (the tail of the last
    // quad then belongs to the next texel and is never used), not narrower than C
    if (lay.texel_bytes % 4 != 0 || lay.texel_bytes < a.C * 4) return TEMPO_E_ARG;
    a.texel_bytes = lay.texel_bytes;
    a.row_bytes = a.W * a.texel_bytes;
    if (lay.frame_bytes < (long long)a.H * a.row_bytes) return TEMPO_E_ARG;
    a.frame_bytes = lay.frame_bytes;
    for (int v = 0; v < a.V; v++) {
      if (lay.views[v] == nullptr) return TEMPO_E_NULL;
      if (reinterpret_cast<uintptr_t>(lay.views[v]) % 4 != 0) return TEMPO_E_ARG;
      a.vbase[v] = static_cast<const char*>(lay.views[v]);
    }
  }
  // texel offsets are built with 24-bit multiplies and kept in 31 bits
  if ((long long)a.H * a.row_bytes >= 0x80000000LL || a.row_bytes >= (1 << 23) ||
      a.H >= (1 << 23))
    return TEMPO_E_SIZE;
  // The buffer range of one map ends with the C-th channel of its last texel: the caller
  // vouches for the C channels of every texel and for nothing behind them (a channel slice
  // of a wider tensor is handed over as a shifted pointer, so H * row_bytes from there
  // would reach past the tensor's end).  The fetch lanes cover CP channels; 16-byte buffer
  // loads are range-checked dword by dword, so what lies past the range reads as 0 and
  // causes no memory access (tools/range_check.hip).
  a.map_bytes = (unsigned)((long long)a.H * a.row_bytes - a.texel_bytes + 4ll * a.C);
  return TEMPO_OK;
}

int window_impl(const Layout& lay, const float* cam, const int32_t* win_frame, const float* fx,
                const float* fy, const float* fz, const int32_t* tl, const int32_t* start,
                const int32_t* end, const uint8_t* live, float* cubes, float* plane_xy,
                float* plane_xz, float* plane_yz, int NW, int F, int V, int C, int CP, int H,
                int W, int FX, int FY, int FZ, int VX, int VY, int VZ, float hm_w, float hm_h,
                float img_w, float img_h, int mm_unfused, const int32_t* groups,
                int num_groups, const Extra& ex, void* stream);
int window_impl(const float* hm, const float* cam, const int32_t* win_frame, const float* fx,
                const float* fy, const float* fz, const int32_t* tl, const int32_t* start,
                const int32_t* end, const uint8_t* live, float* cubes, float* plane_xy,
                float* plane_xz, float* plane_yz, int NW, int F, int V, int C, int CP, int H,
                int W, int FX, int FY, int FZ, int VX, int VY, int VZ, float hm_w, float hm_h,
                float img_w, float img_h, int mm_unfused, const int32_t* groups,
                int num_groups, const Extra& ex, void* stream) {
  if (!hm) return TEMPO_E_NULL;
  Layout lay;
  lay.hm = hm;
  return window_impl(lay, cam, win_frame, fx, fy, fz, tl, start, end, live, cubes, plane_xy,
                     plane_xz, plane_yz, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ, hm_w,
                     hm_h, img_w, img_h, mm_unfused, groups, num_groups, ex, stream);
}

int root_impl(const Layout& lay, const float* cam, const float* ax_x, const float* ax_y,
              const float* ax_z, int ax_batched, const uint8_t* live, float* cubes,
              float* grids, void* mask, int mask_bytes, int B, int V, int C, int CP, int H,
              int W, int X, int Y, int Z, float hm_w, float hm_h, float img_w, float img_h,
              int mm_unfused, const int32_t* groups, int num_groups, const Extra& ex,
              void* stream);
int root_impl(const float* hm, const float* cam, const float* ax_x, const float* ax_y,
              const float* ax_z, int ax_batched, const uint8_t* live, float* cubes,
              float* grids, void* mask, int mask_bytes, int B, int V, int C, int CP, int H,
              int W, int X, int Y, int Z, float hm_w, float hm_h, float img_w, float img_h,
              int mm_unfused, const int32_t* groups, int num_groups, const Extra& ex,
              void* stream) {
  if (!hm) return TEMPO_E_NULL;
  Layout lay;
  lay.hm = hm;
  return root_impl(lay, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes, grids, mask, mask_bytes,
                   B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, groups,
                   num_groups, ex, stream);
}

// tempo_repack_nhwc / tempo_repack_nhwc_half: element type DT, everything else alike
template <int DT>
int repack_impl(const void* const* views, int V, int B, int C_total, int c_begin, int C, int H,
                int W, int CP, float* out, void* stream) {
  if (views == nullptr || out == nullptr) return TEMPO_E_NULL;
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (B <= 0 || B > 65535 || c_begin < 0 || c_begin + C > C_total) return TEMPO_E_DIM;
  if (!aligned16(out)) return TEMPO_E_ARG;  // written with 128-bit stores
  constexpr uintptr_t elem = DT == TEMPO_DTYPE_F32 ? 4 : 2;
  RepackArgs a;
  for (int v = 0; v < V; v++) {
    if (views[v] == nullptr) return TEMPO_E_NULL;
    if (reinterpret_cast<uintptr_t>(views[v]) % elem != 0) return TEMPO_E_ARG;
    a.views[v] = views[v];
  }
  for (int v = V; v < TEMPO_MAX_VIEWS; v++) a.views[v] = nullptr;
  a.out = out;
  a.V = V; a.B = B; a.C_total = C_total; a.c_begin = c_begin; a.C = C;
  a.HW = H * W; a.CP = CP; a.b0 = 0;
  // the wide tile reads 4 pixels per lane in one load (16 bytes of fp32, 8 of fp16 / bf16):
  // every channel plane of every frame has to start on such a boundary
  bool wide = a.HW % 4 == 0 && (((long long)C_total * a.HW) % 4 == 0) &&
              (((long long)c_begin * a.HW) % 4 == 0);
  for (int v = 0; v < V; v++)
    wide = wide && (reinterpret_cast<uintptr_t>(views[v]) % (4 * elem) == 0);
  if (wide) {
    switch (CP / 4) {
      case 4: launch_repack_wide<4, DT>(a, (hipStream_t)stream); break;
      case 5: launch_repack_wide<5, DT>(a, (hipStream_t)stream); break;
      case 6: launch_repack_wide<6, DT>(a, (hipStream_t)stream); break;
      case 7: launch_repack_wide<7, DT>(a, (hipStream_t)stream); break;
      case 8: launch_repack_wide<8, DT>(a, (hipStream_t)stream); break;
      default: return TEMPO_E_PAD;
    }
  } else {
    dim3 grid((a.HW + kRepackPix - 1) / kRepackPix, V, B);
    hipLaunchKernelGGL(repack_kernel<DT>, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int tempo_version(void) { return TEMPO_ABI_VERSION; }

const char* tempo_error_string(int code) {
  switch (code) {
    case TEMPO_OK: return "ok";
    case TEMPO_E_NULL: return "tempo: a required pointer is NULL";
    case TEMPO_E_DIM: return "tempo: a dimension is <= 0 or exceeds the library limit";
    case TEMPO_E_PAD: return "tempo: padded channel count must be a multiple of 4 in [max(C,16), 32]";
    case TEMPO_E_SIZE:
      return "tempo: one view's map or one volume exceeds 2^31 bytes, or one frame's maps "
             "exceed the static scratch of tempo_root_step_static";
    case TEMPO_E_ARG: return "tempo: inconsistent argument";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "tempo: unknown error";
  }
}

int tempo_padded_channels(int channels) {
  if (channels <= 0 || channels > TEMPO_MAX_CHANNELS) return TEMPO_E_DIM;
  const int cp = (channels + 3) / 4 * 4;
  return cp < 16 ? 16 : cp;
}

int tempo_repack_nhwc(const void* const* views, int V, int B, int C_total, int c_begin,
                      int C, int H, int W, int CP, float* out, void* stream) {
  return repack_impl<TEMPO_DTYPE_F32>(views, V, B, C_total, c_begin, C, H, W, CP, out, stream);
}

int tempo_repack_nhwc_half(const void* const* views, int dtype, int V, int B, int C_total,
                           int c_begin, int C, int H, int W, int CP, float* out,
                           void* stream) {
  if (dtype == TEMPO_DTYPE_F16)
    return repack_impl<TEMPO_DTYPE_F16>(views, V, B, C_total, c_begin, C, H, W, CP, out, stream);
  if (dtype == TEMPO_DTYPE_BF16)
    return repack_impl<TEMPO_DTYPE_BF16>(views, V, B, C_total, c_begin, C, H, W, CP, out,
                                         stream);
  return TEMPO_E_ARG;
}

int tempo_unproject_root(const float* hm, const float* cam, const float* ax_x,
                         const float* ax_y, const float* ax_z, int ax_batched,
                         const uint8_t* live, float* cubes, float* grids, void* mask,
                         int mask_bytes, int B, int V, int C, int CP, int H, int W, int X,
                         int Y, int Z, float hm_w, float hm_h, float img_w, float img_h,
                         int mm_unfused, void* stream) {
  return tempo_unproject_root_grouped(hm, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes,
                                      grids, mask, mask_bytes, B, V, C, CP, H, W, X, Y, Z,
                                      hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0,
                                      stream);
}

int tempo_unproject_root_grouped(const float* hm, const float* cam, const float* ax_x,
                                 const float* ax_y, const float* ax_z, int ax_batched,
                                 const uint8_t* live, float* cubes, float* grids,
                                 void* mask, int mask_bytes, int B, int V, int C, int CP,
                                 int H, int W, int X, int Y, int Z, float hm_w, float hm_h,
                                 float img_w, float img_h, int mm_unfused,
                                 const int32_t* groups, int num_groups, void* stream) {
  return root_impl(hm, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes, grids, mask, mask_bytes,
                   B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, groups,
                   num_groups, Extra(), stream);
}

int tempo_unproject_root_zmax(const float* hm, const float* cam, const float* ax_x,
                              const float* ax_y, const float* ax_z, int ax_batched,
                              const uint8_t* live, float* cubes, float* grids, void* mask,
                              int mask_bytes, int B, int V, int C, int CP, int H, int W,
                              int X, int Y, int Z, float hm_w, float hm_h, float img_w,
                              float img_h, int mm_unfused, float* zmax, void* stream) {
  if (!zmax) return TEMPO_E_NULL;
  Extra ex;
  ex.zmax = zmax;
  return root_impl(hm, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes, grids, mask, mask_bytes,
                   B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0,
                   ex, stream);
}

int tempo_clamp_mask_words(int num_voxels, int CP) {
  if (num_voxels <= 0) return TEMPO_E_DIM;
  if (CP % 4 != 0 || CP < 16 || CP > TEMPO_MAX_CHANNELS) return TEMPO_E_PAD;
  return (num_voxels + kWaveVox - 1) / kWaveVox * mask_words(CP / 4);
}

int tempo_unproject_root_train(const float* hm, const float* cam, const float* ax_x,
                               const float* ax_y, const float* ax_z, int ax_batched,
                               const uint8_t* live, float* cubes, float* grids, void* mask,
                               int mask_bytes, int B, int V, int C, int CP, int H, int W,
                               int X, int Y, int Z, float hm_w, float hm_h, float img_w,
                               float img_h, int mm_unfused, uint64_t* clamp_mask,
                               void* stream) {
  if (!clamp_mask) return TEMPO_E_NULL;
  Extra ex;
  ex.cmask = reinterpret_cast<unsigned long long*>(clamp_mask);
  return root_impl(hm, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes, grids, mask, mask_bytes,
                   B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0,
                   ex, stream);
}

int tempo_unproject_root_backward(const float* grad_cubes, const float* cam,
                                  const float* ax_x, const float* ax_y, const float* ax_z,
                                  int ax_batched, const uint8_t* live,
                                  const uint64_t* clamp_mask, float* grad_hm, int B, int V,
                                  int C, int CP, int H, int W, int X, int Y, int Z,
                                  float hm_w, float hm_h, float img_w, float img_h,
                                  int mm_unfused, void* stream) {
  if (!grad_cubes || !clamp_mask || !grad_hm) return TEMPO_E_NULL;
  Extra ex;
  ex.cmask = reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(clamp_mask));
  ex.ghm = grad_hm;
  return root_impl(grad_hm, cam, ax_x, ax_y, ax_z, ax_batched, live,
                   const_cast<float*>(grad_cubes), nullptr, nullptr, 0, B, V, C, CP, H, W, X, Y,
                   Z, hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0, ex, stream);
}

int tempo_unproject_window_train(const float* hm, const float* cam, const int32_t* win_frame,
                                 const float* fx, const float* fy, const float* fz,
                                 const int32_t* tl, const int32_t* start, const int32_t* end,
                                 const uint8_t* live, float* cubes, int NW, int F, int V,
                                 int C, int CP, int H, int W, int FX, int FY, int FZ, int VX,
                                 int VY, int VZ, float hm_w, float hm_h, float img_w,
                                 float img_h, int mm_unfused, uint64_t* clamp_mask,
                                 void* stream) {
  if (!cubes || !clamp_mask) return TEMPO_E_NULL;
  Extra ex;
  ex.cmask = reinterpret_cast<unsigned long long*>(clamp_mask);
  return window_impl(hm, cam, win_frame, fx, fy, fz, tl, start, end, live, cubes, nullptr,
                     nullptr, nullptr, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ, hm_w,
                     hm_h, img_w, img_h, mm_unfused, nullptr, 0, ex, stream);
}

int tempo_unproject_window_backward(const float* grad_cubes, const float* cam,
                                    const int32_t* win_frame, const float* fx,
                                    const float* fy, const float* fz, const int32_t* tl,
                                    const int32_t* start, const int32_t* end,
                                    const uint8_t* live, const uint64_t* clamp_mask,
                                    float* grad_hm, int NW, int F, int V, int C, int CP, int H,
                                    int W, int FX, int FY, int FZ, int VX, int VY, int VZ,
                                    float hm_w, float hm_h, float img_w, float img_h,
                                    int mm_unfused, int tile_texels, void* stream) {
  if (!grad_cubes || !clamp_mask || !grad_hm) return TEMPO_E_NULL;
  if (tile_texels != 0 && tile_texels != kBrickCapSmall && tile_texels != kBrickCapLarge)
    return TEMPO_E_ARG;
  Extra ex;
  ex.cmask = reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(clamp_mask));
  ex.ghm = grad_hm;
  ex.tile = tile_texels;
  return window_impl(grad_hm, cam, win_frame, fx, fy, fz, tl, start, end, live,
                     const_cast<float*>(grad_cubes), nullptr, nullptr, nullptr, NW, F, V, C, CP,
                     H, W, FX, FY, FZ, VX, VY, VZ, hm_w, hm_h, img_w, img_h, mm_unfused,
                     nullptr, 0, ex, stream);
}

int tempo_window_planes_claims(const float* hm, const float* cam, const int32_t* win_frame,
                               const float* fx, const float* fy, const float* fz,
                               const int32_t* tl, const int32_t* start, const int32_t* end,
                               const uint8_t* live, const float* plane_xy,
                               const float* plane_xz, const float* plane_yz, int32_t* claim_xy,
                               int32_t* claim_xz, int32_t* claim_yz, int NW, int F, int V,
                               int C, int CP, int H, int W, int FX, int FY, int FZ, int VX,
                               int VY, int VZ, float hm_w, float hm_h, float img_w,
                               float img_h, int mm_unfused, void* stream) {
  if (!hm || !plane_xy || !plane_xz || !plane_yz || !claim_xy || !claim_xz || !claim_yz)
    return TEMPO_E_NULL;
  // a claim word is 2 * index + blocked: keep it well inside 31 bits
  if (VX >= (1 << 23) || VY >= (1 << 23) || VZ >= (1 << 23)) return TEMPO_E_DIM;
  Extra ex;
  ex.planes_in = true;
  ex.claim[0] = claim_xy; ex.claim[1] = claim_xz; ex.claim[2] = claim_yz;
  return window_impl(hm, cam, win_frame, fx, fy, fz, tl, start, end, live, nullptr,
                     const_cast<float*>(plane_xy), const_cast<float*>(plane_xz),
                     const_cast<float*>(plane_yz), NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY,
                     VZ, hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0, ex, stream);
}

int tempo_unproject_window_planes_backward(
    const float* grad_xy, const float* grad_xz, const float* grad_yz, const int32_t* claim_xy,
    const int32_t* claim_xz, const int32_t* claim_yz, const float* cam,
    const int32_t* win_frame, const float* fx, const float* fy, const float* fz,
    const int32_t* tl, const int32_t* start, const int32_t* end, const uint8_t* live,
    float* grad_hm, int NW, int F, int V, int C, int CP, int H, int W, int FX, int FY, int FZ,
    int VX, int VY, int VZ, float hm_w, float hm_h, float img_w, float img_h, int mm_unfused,
    int tile_texels, void* stream) {
  if (!grad_xy || !grad_xz || !grad_yz || !claim_xy || !claim_xz || !claim_yz || !grad_hm)
    return TEMPO_E_NULL;
  if (tile_texels != 0 && tile_texels != kBrickCapPlanes && tile_texels != kBrickCapSmall &&
      tile_texels != kBrickCapLarge)
    return TEMPO_E_ARG;
  Extra ex;
  ex.ghm = grad_hm;
  ex.tile = tile_texels;
  ex.gplane[0] = grad_xy; ex.gplane[1] = grad_xz; ex.gplane[2] = grad_yz;
  ex.claim[0] = const_cast<int32_t*>(claim_xy);
  ex.claim[1] = const_cast<int32_t*>(claim_xz);
  ex.claim[2] = const_cast<int32_t*>(claim_yz);
  return window_impl(grad_hm, cam, win_frame, fx, fy, fz, tl, start, end, live, nullptr,
                     nullptr, nullptr, nullptr, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ,
                     hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0, ex, stream);
}

int tempo_unrepack_nchw(const float* in, void* const* views, int V, int B, int C_total,
                        int c_begin, int C, int H, int W, int CP, void* stream) {
  if (in == nullptr || views == nullptr) return TEMPO_E_NULL;
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (B <= 0 || B > 65535 || c_begin < 0 || c_begin + C > C_total) return TEMPO_E_DIM;
  if (!aligned16(in)) return TEMPO_E_ARG;
  UnrepackArgs a;
  for (int v = 0; v < TEMPO_MAX_VIEWS; v++) a.views[v] = nullptr;
  for (int v = 0; v < V; v++) {
    if (views[v] == nullptr) return TEMPO_E_NULL;
    a.views[v] = static_cast<float*>(views[v]);
  }
  a.in = in;
  a.V = V; a.B = B; a.C_total = C_total; a.c_begin = c_begin; a.C = C;
  a.HW = H * W; a.CP = CP;
  bool wide = a.HW % 4 == 0 && (((long long)C_total * a.HW) % 4 == 0) &&
              (((long long)c_begin * a.HW) % 4 == 0) &&
              reinterpret_cast<uintptr_t>(in) % 16 == 0;
  for (int v = 0; v < V; v++) wide = wide && (reinterpret_cast<uintptr_t>(views[v]) % 16 == 0);
  if (wide) {
    switch (CP / 4) {
      case 4: launch_unrepack_wide<4>(a, (hipStream_t)stream); break;
      case 5: launch_unrepack_wide<5>(a, (hipStream_t)stream); break;
      case 6: launch_unrepack_wide<6>(a, (hipStream_t)stream); break;
      case 7: launch_unrepack_wide<7>(a, (hipStream_t)stream); break;
      case 8: launch_unrepack_wide<8>(a, (hipStream_t)stream); break;
      default: return TEMPO_E_PAD;
    }
  } else {
    dim3 grid((a.HW + 63) / 64, V, B);
    hipLaunchKernelGGL(unrepack_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

}  // extern "C"

namespace {

int root_impl(const Layout& lay, const float* cam, const float* ax_x, const float* ax_y,
              const float* ax_z, int ax_batched, const uint8_t* live, float* cubes,
              float* grids, void* mask, int mask_bytes, int B, int V, int C, int CP, int H,
              int W, int X, int Y, int Z, float hm_w, float hm_h, float img_w, float img_h,
              int mm_unfused, const int32_t* groups, int num_groups, const Extra& ex,
              void* stream) {
  if (!cam || !ax_x || !ax_y || !ax_z) return TEMPO_E_NULL;
  // the volume may be left out only where another output is asked for: the z-max planes
  // (plain forward launch: not grouped, not training, not backward)
  if (!cubes && (ex.zmax == nullptr || groups != nullptr || ex.cmask != nullptr ||
                 ex.ghm != nullptr))
    return TEMPO_E_NULL;
  if ((groups == nullptr) != (num_groups == 0) || num_groups < 0 || num_groups > B)
    return TEMPO_E_ARG;
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0) return TEMPO_E_DIM;
  const long long N = (long long)X * Y * Z;
  if (N * 4 >= 0x80000000LL) return TEMPO_E_SIZE;
  if ((mask == nullptr) != (mask_bytes == 0)) return TEMPO_E_ARG;
  if (mask && !((mask_bytes == 1 && V <= 8) || (mask_bytes == 2 && V <= 16) ||
                mask_bytes == 4))
    return TEMPO_E_ARG;
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  // the grouped launch and the backward work on the repacked buffer only
  if (lay.hm == nullptr && (groups != nullptr || ex.ghm != nullptr)) return TEMPO_E_ARG;
  UnprojArgs a = {};
  a.cam = cam;
  a.ax[0] = ax_x; a.ax[1] = ax_y; a.ax[2] = ax_z;
  a.ax_stride[0] = ax_batched ? X : 0;
  a.ax_stride[1] = ax_batched ? Y : 0;
  a.ax_stride[2] = ax_batched ? Z : 0;
  a.live = live; a.cubes = cubes; a.grids = grids; a.mask = mask;
  a.mask_bytes = mask_bytes;
  a.frame_stride = (long long)V * H * W * CP;
  a.view_bytes = (unsigned)((long long)H * W * CP * 4);
  a.NB = B; a.F = B; a.V = V; a.C = C; a.Cs = C; a.CP = CP; a.H = H; a.W = W;
  a.dim[0] = X; a.dim[1] = Y; a.dim[2] = Z; a.N = (int)N;
  a.geom = tempo_make_geom(hm_w, hm_h, img_w, img_h);
  a.groups = groups; a.NG = num_groups;
  a.cmask = ex.cmask;
  const int lrc = apply_layout(a, lay);
  if (lrc != TEMPO_OK) return lrc;
  if (ex.ghm != nullptr) {  // backward: `cubes` is the incoming gradient
    a.ghm = ex.ghm;
    const hipError_t e = hipMemsetAsync(ex.ghm, 0, (size_t)B * a.frame_stride * sizeof(float),
                                        (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return launch_backward<MODE_ROOT>(a, mm_unfused, (hipStream_t)stream);
  }
  if (ex.zmax != nullptr) {
    // max-accumulated with atomics: starts at +0.0, the smallest clamped value; waves of
    // whole columns (launch_one) store every entry themselves, dead items included
    if (groups != nullptr || a.cmask != nullptr || zmax_wave_vox(Z) == kWaveVox) {
      const hipError_t e = hipMemsetAsync(ex.zmax, 0, (size_t)B * C * X * Y * sizeof(float),
                                          (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
    }
    a.pl01 = reinterpret_cast<unsigned*>(ex.zmax);
  }
  return launch_forward<MODE_ROOT>(a, mm_unfused, (hipStream_t)stream,
                                   groups == nullptr && a.cmask == nullptr ? ex.tail : nullptr);
}

}  // namespace

extern "C" {

long long tempo_root_step_workspace(int B, int V, int H, int W, int CP, int chunk) {
  if (B <= 0 || V <= 0 || H <= 0 || W <= 0 || CP <= 0 || chunk <= 0) return TEMPO_E_DIM;
  // one chunk's channel-last maps; two of them when the batch takes more than one chunk (the
  // next chunk is repacked into the other half while the current one is unprojected)
  return (long long)(B < chunk ? B : chunk) * V * H * W * CP * (B > chunk ? 2 : 1);
}

}  // extern "C"

namespace {

int root_step_impl(const void* const* views, int C_total, int c_begin, const float* cam,
                   const float* ax_x, const float* ax_y, const float* ax_z, int ax_batched,
                   const uint8_t* live, float* cubes, float* grids, void* mask,
                   int mask_bytes, int B, int V, int C, int CP, int H, int W, int X, int Y,
                   int Z, float hm_w, float hm_h, float img_w, float img_h, int mm_unfused,
                   float* zmax, float* workspace, int chunk, bool two_halves, void* stream) {
  if (views == nullptr || workspace == nullptr || cam == nullptr ||
      (cubes == nullptr && zmax == nullptr))
    return TEMPO_E_NULL;
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (B <= 0 || B > 65535 || chunk <= 0 || c_begin < 0 || c_begin + C > C_total ||
      X <= 0 || Y <= 0 || Z <= 0)
    return TEMPO_E_DIM;
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  if (!aligned16(workspace)) return TEMPO_E_ARG;
  for (int v = 0; v < V; v++)
    if (views[v] == nullptr) return TEMPO_E_NULL;
  const long long HW = (long long)H * W;
  const long long N = (long long)X * Y * Z;
  const long long half = (long long)(B < chunk ? B : chunk) * V * HW * CP;  // floats per chunk
  // With a workspace of two chunk-sized halves the repack of chunk c + 1 rides on the launch
  // that unprojects chunk c (unproject_repack_kernel); it needs the wide repack's conditions
  // (16-byte aligned planes).  Otherwise the chunks go through one buffer, one after the other.
  // Both halves must stay in the 256 MB Infinity Cache together, or the rider evicts the maps
  // being gathered: config 2, ms per 32-frame step, fused / one buffer: 8-frame chunks (2 x
  // 79 MB) 0.277 / 0.297, 11 frames (2 x 108 MB) 0.280 / 0.299, 16 frames (2 x 158 MB)
  // 0.301 / 0.270 -- so the default 16-frame chunks go through one buffer.
  bool fuse = two_halves && B > chunk && half * 4 * 2 <= kFuseBytes && HW % 4 == 0 &&
              ((long long)C_total * HW) % 4 == 0 && ((long long)c_begin * HW) % 4 == 0 &&
              (half * 4) % 16 == 0;
  for (int v = 0; v < V; v++) fuse = fuse && reinterpret_cast<uintptr_t>(views[v]) % 16 == 0;
#ifdef TEMPO_NO_TAIL_FUSION
  fuse = false;
#endif
  const void* shifted[TEMPO_MAX_VIEWS];
  int err = TEMPO_OK;
  int ci = 0;
  for (int b0 = 0; b0 < B && err == TEMPO_OK; b0 += chunk, ci++) {
    const int n = b0 + chunk <= B ? chunk : B - b0;
    float* buf = workspace + (fuse && (ci & 1) ? half : 0);
    if (!fuse || b0 == 0) {
      for (int v = 0; v < V; v++)
        shifted[v] = static_cast<const float*>(views[v]) + (long long)b0 * C_total * HW;
      err = tempo_repack_nhwc(shifted, V, n, C_total, c_begin, C, H, W, CP, buf, stream);
      if (err != TEMPO_OK) break;
    }
    Extra ex;
    ex.zmax = zmax ? zmax + (long long)b0 * C * X * Y : nullptr;
    RepackArgs tail;
    if (fuse && b0 + chunk < B) {  // the next chunk, into the other half
      const int b1 = b0 + chunk;
      for (int v = 0; v < TEMPO_MAX_VIEWS; v++)
        tail.views[v] = v < V ? views[v] : nullptr;
      tail.out = workspace + ((ci & 1) ? 0 : half);
      tail.V = V; tail.B = b1 + chunk <= B ? chunk : B - b1;
      tail.C_total = C_total; tail.c_begin = c_begin; tail.C = C;
      tail.HW = (int)HW; tail.CP = CP; tail.b0 = b1;
      ex.tail = &tail;
    }
    err = root_impl(buf, cam + (long long)b0 * V * CAM_FLOATS,
                    ax_x + (ax_batched ? (long long)b0 * X : 0),
                    ax_y + (ax_batched ? (long long)b0 * Y : 0),
                    ax_z + (ax_batched ? (long long)b0 * Z : 0), ax_batched,
                    live ? live + b0 : nullptr,
                    cubes ? cubes + (long long)b0 * C * N : nullptr,
                    grids ? grids + (long long)b0 * N * 3 : nullptr,
                    mask ? static_cast<char*>(mask) + (long long)b0 * N * mask_bytes : nullptr,
                    mask_bytes, n, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h,
                    mm_unfused, nullptr, 0, ex, stream);
  }
  return err;
}

}  // namespace

extern "C" {

int tempo_root_step(const void* const* views, int C_total, int c_begin, const float* cam,
                    const float* ax_x, const float* ax_y, const float* ax_z, int ax_batched,
                    const uint8_t* live, float* cubes, float* grids, void* mask,
                    int mask_bytes, int B, int V, int C, int CP, int H, int W, int X, int Y,
                    int Z, float hm_w, float hm_h, float img_w, float img_h, int mm_unfused,
                    float* workspace, int chunk, void* stream) {
  return root_step_impl(views, C_total, c_begin, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes,
                        grids, mask, mask_bytes, B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w,
                        img_h, mm_unfused, nullptr, workspace, chunk, true, stream);
}

int tempo_root_step_zmax(const void* const* views, int C_total, int c_begin, const float* cam,
                         const float* ax_x, const float* ax_y, const float* ax_z,
                         int ax_batched, const uint8_t* live, float* cubes, float* grids,
                         void* mask, int mask_bytes, int B, int V, int C, int CP, int H, int W,
                         int X, int Y, int Z, float hm_w, float hm_h, float img_w, float img_h,
                         int mm_unfused, float* zmax, float* workspace, int chunk,
                         void* stream) {
  if (zmax == nullptr) return TEMPO_E_NULL;  // (cubes may be NULL: z-max alone)
  return root_step_impl(views, C_total, c_begin, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes,
                        grids, mask, mask_bytes, B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w,
                        img_h, mm_unfused, zmax, workspace, chunk, true, stream);
}

}  // extern "C"

// ------------------------------------------- entry point that allocates nothing
// For callers that cannot call hipMalloc (HIP graph capture, serving loops, harnesses that
// forbid it): tempo_root_step with its channel-last staging buffer in a file-scope device
// array instead of a caller-allocated workspace.  Every GPU the library runs on holds its own
// copy of the array at its own address until the process ends; the host-side name `scratch`
// is only a handle for hipGetSymbolAddress, never a device address.
namespace {

static __device__ __align__(256) unsigned char scratch[TEMPO_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices];  // per device, filled at the first call on that device
std::mutex g_scratch_mutex;

// device address of `scratch` on the current device (looked up once per device)
int static_scratch(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return TEMPO_E_DIM;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (p == nullptr) return TEMPO_E_NULL;
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_scratch_ptr[dev];
  return TEMPO_OK;
}

}  // namespace

extern "C" {

int tempo_root_step_static(const void* const* views, int C_total, int c_begin,
                           const float* cam, const float* ax_x, const float* ax_y,
                           const float* ax_z, int ax_batched, const uint8_t* live,
                           float* cubes, float* grids, void* mask, int mask_bytes, int B,
                           int V, int C, int CP, int H, int W, int X, int Y, int Z,
                           float hm_w, float hm_h, float img_w, float img_h, int mm_unfused,
                           int chunk, void* stream) {
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (chunk <= 0) return TEMPO_E_DIM;
  // one frame's channel-last maps must fit the static array; the chunk is cut to what fits
  const long long frame_bytes = (long long)V * H * W * CP * 4;
  if (frame_bytes > (long long)TEMPO_STATIC_SCRATCH_BYTES) return TEMPO_E_SIZE;
  const long long fit = (long long)TEMPO_STATIC_SCRATCH_BYTES / frame_bytes;
  if (chunk > fit) chunk = (int)fit;
  float* ws = nullptr;
  const int e = static_scratch(&ws);
  if (e != TEMPO_OK) return e;
  // (nothing to clear: the repack overwrites every byte of a chunk's region, padding
  // channels included, before the unprojection of that chunk reads it)
  // (one chunk-sized region of the array, chunks one after the other: two halves would halve
  // the chunk that fits)
  return root_step_impl(views, C_total, c_begin, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes,
                        grids, mask, mask_bytes, B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w,
                        img_h, mm_unfused, nullptr, ws, chunk, false, stream);
}

// ------------------------------------------- heatmaps read in place (channel-last views)
int tempo_unproject_root_views(const void* const* views, long long frame_bytes,
                               int texel_bytes, const float* cam, const float* ax_x,
                               const float* ax_y, const float* ax_z, int ax_batched,
                               const uint8_t* live, float* cubes, float* grids, void* mask,
                               int mask_bytes, int B, int V, int C, int CP, int H, int W, int X,
                               int Y, int Z, float hm_w, float hm_h, float img_w, float img_h,
                               int mm_unfused, void* stream) {
  if (!views) return TEMPO_E_NULL;
  Layout lay;
  lay.views = views; lay.frame_bytes = frame_bytes; lay.texel_bytes = texel_bytes;
  return root_impl(lay, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes, grids, mask, mask_bytes,
                   B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0,
                   Extra(), stream);
}

int tempo_unproject_root_views_zmax(const void* const* views, long long frame_bytes,
                                    int texel_bytes, const float* cam, const float* ax_x,
                                    const float* ax_y, const float* ax_z, int ax_batched,
                                    const uint8_t* live, float* cubes, float* grids, void* mask,
                                    int mask_bytes, int B, int V, int C, int CP, int H, int W,
                                    int X, int Y, int Z, float hm_w, float hm_h, float img_w,
                                    float img_h, int mm_unfused, float* zmax, void* stream) {
  if (!views || !zmax) return TEMPO_E_NULL;
  Layout lay;
  lay.views = views; lay.frame_bytes = frame_bytes; lay.texel_bytes = texel_bytes;
  Extra ex;
  ex.zmax = zmax;
  return root_impl(lay, cam, ax_x, ax_y, ax_z, ax_batched, live, cubes, grids, mask, mask_bytes,
                   B, V, C, CP, H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0,
                   ex, stream);
}

int tempo_unproject_window_views(const void* const* views, long long frame_bytes,
                                 int texel_bytes, const float* cam, const int32_t* win_frame,
                                 const float* fx, const float* fy, const float* fz,
                                 const int32_t* tl, const int32_t* start, const int32_t* end,
                                 const uint8_t* live, float* cubes, int NW, int F, int V, int C,
                                 int CP, int H, int W, int FX, int FY, int FZ, int VX, int VY,
                                 int VZ, float hm_w, float hm_h, float img_w, float img_h,
                                 int mm_unfused, void* stream) {
  if (!views || !cubes) return TEMPO_E_NULL;
  Layout lay;
  lay.views = views; lay.frame_bytes = frame_bytes; lay.texel_bytes = texel_bytes;
  return window_impl(lay, cam, win_frame, fx, fy, fz, tl, start, end, live, cubes, nullptr,
                     nullptr, nullptr, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ, hm_w, hm_h,
                     img_w, img_h, mm_unfused, nullptr, 0, Extra(), stream);
}

int tempo_unproject_window_planes_views(const void* const* views, long long frame_bytes,
                                        int texel_bytes, const float* cam,
                                        const int32_t* win_frame, const float* fx,
                                        const float* fy, const float* fz, const int32_t* tl,
                                        const int32_t* start, const int32_t* end,
                                        const uint8_t* live, float* cubes, float* plane_xy,
                                        float* plane_xz, float* plane_yz, int NW, int F, int V,
                                        int C, int CP, int H, int W, int FX, int FY, int FZ,
                                        int VX, int VY, int VZ, float hm_w, float hm_h,
                                        float img_w, float img_h, int mm_unfused,
                                        void* stream) {
  if (!views) return TEMPO_E_NULL;
  if (!cubes && !plane_xy && !plane_xz && !plane_yz) return TEMPO_E_NULL;
  Layout lay;
  lay.views = views; lay.frame_bytes = frame_bytes; lay.texel_bytes = texel_bytes;
  return window_impl(lay, cam, win_frame, fx, fy, fz, tl, start, end, live, cubes, plane_xy,
                     plane_xz, plane_yz, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ, hm_w,
                     hm_h, img_w, img_h, mm_unfused, nullptr, 0, Extra(), stream);
}

int tempo_unproject_window(const float* hm, const float* cam, const int32_t* win_frame,
                           const float* fx, const float* fy, const float* fz,
                           const int32_t* tl, const int32_t* start, const int32_t* end,
                           const uint8_t* live, float* cubes, int NW, int F, int V, int C,
                           int CP, int H, int W, int FX, int FY, int FZ, int VX, int VY,
                           int VZ, float hm_w, float hm_h, float img_w, float img_h,
                           int mm_unfused, void* stream) {
  return tempo_unproject_window_grouped(hm, cam, win_frame, fx, fy, fz, tl, start, end, live,
                                        cubes, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ,
                                        hm_w, hm_h, img_w, img_h, mm_unfused, nullptr, 0,
                                        stream);
}

int tempo_unproject_window_grouped(const float* hm, const float* cam,
                                   const int32_t* win_frame, const float* fx,
                                   const float* fy, const float* fz, const int32_t* tl,
                                   const int32_t* start, const int32_t* end,
                                   const uint8_t* live, float* cubes, int NW, int F, int V,
                                   int C, int CP, int H, int W, int FX, int FY, int FZ,
                                   int VX, int VY, int VZ, float hm_w, float hm_h,
                                   float img_w, float img_h, int mm_unfused,
                                   const int32_t* groups, int num_groups, void* stream) {
  if (!cubes) return TEMPO_E_NULL;
  return window_impl(hm, cam, win_frame, fx, fy, fz, tl, start, end, live, cubes, nullptr,
                     nullptr, nullptr, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ, hm_w,
                     hm_h, img_w, img_h, mm_unfused, groups, num_groups, Extra(), stream);
}

int tempo_unproject_window_planes(const float* hm, const float* cam,
                                  const int32_t* win_frame, const float* fx,
                                  const float* fy, const float* fz, const int32_t* tl,
                                  const int32_t* start, const int32_t* end,
                                  const uint8_t* live, float* cubes, float* plane_xy,
                                  float* plane_xz, float* plane_yz, int NW, int F, int V,
                                  int C, int CP, int H, int W, int FX, int FY, int FZ, int VX,
                                  int VY, int VZ, float hm_w, float hm_h, float img_w,
                                  float img_h, int mm_unfused, void* stream) {
  if (!cubes && !plane_xy && !plane_xz && !plane_yz) return TEMPO_E_NULL;
  return window_impl(hm, cam, win_frame, fx, fy, fz, tl, start, end, live, cubes, plane_xy,
                     plane_xz, plane_yz, NW, F, V, C, CP, H, W, FX, FY, FZ, VX, VY, VZ, hm_w,
                     hm_h, img_w, img_h, mm_unfused, nullptr, 0, Extra(), stream);
}

}  // extern "C"

namespace {

int window_impl(const Layout& lay, const float* cam, const int32_t* win_frame, const float* fx,
                const float* fy, const float* fz, const int32_t* tl, const int32_t* start,
                const int32_t* end, const uint8_t* live, float* cubes, float* plane_xy,
                float* plane_xz, float* plane_yz, int NW, int F, int V, int C, int CP, int H,
                int W, int FX, int FY, int FZ, int VX, int VY, int VZ, float hm_w, float hm_h,
                float img_w, float img_h, int mm_unfused, const int32_t* groups,
                int num_groups, const Extra& ex, void* stream) {
  if (!cam || !fx || !fy || !fz || !tl || !start || !end) return TEMPO_E_NULL;
  if ((groups == nullptr) != (num_groups == 0) || num_groups < 0 || num_groups > NW)
    return TEMPO_E_ARG;
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (NW <= 0 || F <= 0 || FX <= 0 || FY <= 0 || FZ <= 0 || VX <= 0 || VY <= 0 || VZ <= 0)
    return TEMPO_E_DIM;
  if (win_frame == nullptr && NW != F) return TEMPO_E_ARG;
  const long long N = (long long)VX * VY * VZ;
  if (N * 4 >= 0x80000000LL) return TEMPO_E_SIZE;
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  if (lay.hm == nullptr && (groups != nullptr || ex.ghm != nullptr)) return TEMPO_E_ARG;
  UnprojArgs a = {};
  a.cam = cam;
  a.ax[0] = fx; a.ax[1] = fy; a.ax[2] = fz;
  a.live = live; a.win_frame = win_frame; a.tl = tl; a.start = start; a.end = end;
  a.cubes = cubes;
  a.frame_stride = (long long)V * H * W * CP;
  a.view_bytes = (unsigned)((long long)H * W * CP * 4);
  a.NB = NW; a.F = F; a.V = V; a.C = C; a.Cs = C; a.CP = CP; a.H = H; a.W = W;
  a.dim[0] = VX; a.dim[1] = VY; a.dim[2] = VZ; a.N = (int)N;
  a.fdim[0] = FX; a.fdim[1] = FY; a.fdim[2] = FZ;
  a.geom = tempo_make_geom(hm_w, hm_h, img_w, img_h);
  a.groups = groups; a.NG = num_groups;
  const int lrc = apply_layout(a, lay);
  if (lrc != TEMPO_OK) return lrc;
  const size_t nc = (size_t)NW * C * sizeof(float);
  float* const pl[3] = {plane_xy, plane_xz, plane_yz};
  const size_t pln[3] = {nc * VX * VY, nc * VX * VZ, nc * VY * VZ};
  if (ex.planes_in) {
    // claim pass: the finished planes come in, the claims start at a large positive word
    // (any real claim, 2 * index + blocked < 2^24, is smaller)
    if (lay.hm == nullptr || groups != nullptr) return TEMPO_E_ARG;
    for (int i = 0; i < 3; i++) {
      if (pl[i] == nullptr || ex.claim[i] == nullptr) return TEMPO_E_NULL;
      const hipError_t e = hipMemsetAsync(ex.claim[i], 0x7f, pln[i], (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
      a.claim[i] = ex.claim[i];
    }
    a.pl01 = reinterpret_cast<unsigned*>(plane_xy);
    a.pl02 = reinterpret_cast<unsigned*>(plane_xz);
    a.pl12 = reinterpret_cast<unsigned*>(plane_yz);
    return launch_claims(a, mm_unfused, (hipStream_t)stream);
  }
  if (ex.gplane[0] != nullptr) {
    // backward from the planes' gradients (claims filed by the pass above)
    if (ex.ghm == nullptr) return TEMPO_E_NULL;
    for (int i = 0; i < 3; i++) {
      if (ex.gplane[i] == nullptr || ex.claim[i] == nullptr) return TEMPO_E_NULL;
      a.gplane[i] = ex.gplane[i];
      a.claim[i] = ex.claim[i];
    }
    a.ghm = ex.ghm;
    const hipError_t e = hipMemsetAsync(ex.ghm, 0, (size_t)F * a.frame_stride * sizeof(float),
                                        (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return launch_backward_planes(a, mm_unfused, (hipStream_t)stream, ex.tile);
  }
  // the planes are max-accumulated: they start at +0.0, the smallest clamped value
  for (int i = 0; i < 3; i++) {
    if (pl[i] == nullptr) continue;
    const hipError_t e = hipMemsetAsync(pl[i], 0, pln[i], (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  a.pl01 = reinterpret_cast<unsigned*>(plane_xy);
  a.pl02 = reinterpret_cast<unsigned*>(plane_xz);
  a.pl12 = reinterpret_cast<unsigned*>(plane_yz);
  a.cmask = ex.cmask;
  if (ex.ghm != nullptr) {  // backward: `cubes` is the incoming gradient
    a.ghm = ex.ghm;
    const hipError_t e = hipMemsetAsync(ex.ghm, 0, (size_t)F * a.frame_stride * sizeof(float),
                                        (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return launch_backward<MODE_WINDOW>(a, mm_unfused, (hipStream_t)stream, ex.tile);
  }
  return launch_forward<MODE_WINDOW>(a, mm_unfused, (hipStream_t)stream);
}

}  // namespace

namespace {

int columns_impl(const Layout& lay, const float* cam, const float* ax_x, const float* ax_y,
                 const float* ax_z, int ax_batched, const uint8_t* live, const int64_t* index,
                 float* out, int B, int K, int V, int C, int CP, int H, int W, int X, int Y,
                 int Z, float hm_w, float hm_h, float img_w, float img_h, int mm_unfused,
                 void* stream) {
  if (!cam || !ax_x || !ax_y || !ax_z || !index || !out) return TEMPO_E_NULL;
  const int rc = check_common(V, C, CP, H, W);
  if (rc != TEMPO_OK) return rc;
  if (B <= 0 || K <= 0 || X <= 0 || Y <= 0 || Z <= 0) return TEMPO_E_DIM;
  if ((long long)X * Y >= 0x7fffffffLL || (long long)K * Z >= 0x7fffffffLL / 4 ||
      (long long)K * Z * C * 4 >= 0x80000000LL)
    return TEMPO_E_SIZE;
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  UnprojArgs a = {};
  a.cam = cam;
  a.ax[0] = ax_x; a.ax[1] = ax_y; a.ax[2] = ax_z;
  a.ax_stride[0] = ax_batched ? X : 0;
  a.ax_stride[1] = ax_batched ? Y : 0;
  a.ax_stride[2] = ax_batched ? Z : 0;
  a.live = live;
  a.frame_stride = (long long)V * H * W * CP;
  a.view_bytes = (unsigned)((long long)H * W * CP * 4);
  a.NB = B; a.F = B; a.V = V; a.C = C; a.Cs = C; a.CP = CP; a.H = H; a.W = W;
  a.dim[0] = 1; a.dim[1] = K; a.dim[2] = Z; a.N = K * Z;  // the virtual grid of the columns
  a.fdim[0] = X; a.fdim[1] = Y; a.fdim[2] = Z;            // the root grid the cells index
  a.geom = tempo_make_geom(hm_w, hm_h, img_w, img_h);
  a.col_index = reinterpret_cast<const long long*>(index);
  a.col_out = out;
  a.K = K;
  const int lrc = apply_layout(a, lay);
  if (lrc != TEMPO_OK) return lrc;
  return launch_columns(a, mm_unfused, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int tempo_unproject_root_columns(const float* hm, const float* cam, const float* ax_x,
                                 const float* ax_y, const float* ax_z, int ax_batched,
                                 const uint8_t* live, const int64_t* index, float* out, int B,
                                 int K, int V, int C, int CP, int H, int W, int X, int Y, int Z,
                                 float hm_w, float hm_h, float img_w, float img_h,
                                 int mm_unfused, void* stream) {
  if (!hm) return TEMPO_E_NULL;
  Layout lay;
  lay.hm = hm;
  return columns_impl(lay, cam, ax_x, ax_y, ax_z, ax_batched, live, index, out, B, K, V, C, CP,
                      H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, stream);
}

int tempo_unproject_root_columns_views(const void* const* views, long long frame_bytes,
                                       int texel_bytes, const float* cam, const float* ax_x,
                                       const float* ax_y, const float* ax_z, int ax_batched,
                                       const uint8_t* live, const int64_t* index, float* out,
                                       int B, int K, int V, int C, int CP, int H, int W, int X,
                                       int Y, int Z, float hm_w, float hm_h, float img_w,
                                       float img_h, int mm_unfused, void* stream) {
  if (!views) return TEMPO_E_NULL;
  Layout lay;
  lay.views = views; lay.frame_bytes = frame_bytes; lay.texel_bytes = texel_bytes;
  return columns_impl(lay, cam, ax_x, ax_y, ax_z, ax_batched, live, index, out, B, K, V, C, CP,
                      H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, stream);
}

int tempo_root_step_columns(const void* const* views, int C_total, int c_begin,
                            const float* cam, const float* ax_x, const float* ax_y,
                            const float* ax_z, int ax_batched, const uint8_t* live,
                            const int64_t* index, float* out, int B, int K, int V, int C,
                            int CP, int H, int W, int X, int Y, int Z, float hm_w, float hm_h,
                            float img_w, float img_h, int mm_unfused, void* stream) {
  if (!views) return TEMPO_E_NULL;
  Layout lay;
  lay.views = views; lay.planar = true; lay.C_total = C_total; lay.c_begin = c_begin;
  return columns_impl(lay, cam, ax_x, ax_y, ax_z, ax_batched, live, index, out, B, K, V, C, CP,
                      H, W, X, Y, Z, hm_w, hm_h, img_w, img_h, mm_unfused, stream);
}

int tempo_conv1x1_nhwc(const float* in, const float* weight, const float* bias, float* out,
                       int T, int V, int B, int K, int C, int CP, int H, int W, void* stream) {
  if (!in || !weight || !out) return TEMPO_E_NULL;
  if (T <= 0 || V <= 0 || V > TEMPO_MAX_VIEWS || B <= 0 || K <= 0 || C <= 0 ||
      C > TEMPO_MAX_CHANNELS || H <= 0 || W <= 0 || (long long)T * V * B > 0x7fffffffLL)
    return TEMPO_E_DIM;
  if (CP % 4 != 0 || CP < C || CP < 16 || CP > TEMPO_MAX_CHANNELS) return TEMPO_E_PAD;
  if (!aligned16(out)) return TEMPO_E_ARG;
  const long long HW = (long long)H * W;
  if (HW * CP * 4 >= 0x80000000LL || HW * K >= 0x7fffffffLL) return TEMPO_E_SIZE;
  const int nt = (CP + 15) / 16;
  if ((size_t)K * 16 * nt * 4 > 65536) return TEMPO_E_DIM;  // the weights live in LDS
  ConvArgs a = {};
  a.in = in; a.weight = weight; a.bias = bias; a.out = out;
  a.V = V; a.B = B; a.K = K; a.C = C; a.CP = CP; a.HW = (int)HW;
  a.images = T * V * B;
  a.tiles = (int)((HW + 63) / 64);
  a.vec = (HW % 4 == 0) && aligned16(in);
  const long long waves = (long long)T * V * B * a.tiles;
  const long long blocks = (waves + kThreads / 64 - 1) / (kThreads / 64);
  if (blocks > 0x7fffffffLL) return TEMPO_E_DIM;
  const size_t lds = (size_t)K * 16 * nt * 4;
  const dim3 grid((unsigned)blocks);
  if (nt == 1)
    hipLaunchKernelGGL((conv1x1_nhwc_kernel<1>), grid, dim3(kThreads), lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv1x1_nhwc_kernel<2>), grid, dim3(kThreads), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int tempo_window_geometry(const float* grid_center, int B, int cols, const float* scale,
                          const float* bias, const float* space_size,
                          const float* sub_space_size, int FX, int FY, int FZ, int VX, int VY,
                          int VZ, int32_t* tl, int32_t* start, int32_t* end, float* offset,
                          uint8_t* live, void* stream) {
  if (!grid_center || !scale || !bias || !space_size || !sub_space_size || !tl || !start ||
      !end || !offset)
    return TEMPO_E_NULL;
  if (B <= 0 || (cols != 3 && cols < 7) || FX <= 1 || FY <= 1 || FZ <= 1 || VX <= 0 ||
      VY <= 0 || VZ <= 0)
    return TEMPO_E_DIM;
  GeomArgs a = {};
  a.gc = grid_center; a.scale = scale; a.bias = bias; a.space = space_size;
  a.sub = sub_space_size;
  a.fine[0] = FX; a.fine[1] = FY; a.fine[2] = FZ;
  a.vox[0] = VX; a.vox[1] = VY; a.vox[2] = VZ;
  a.B = B; a.cols = cols;
  a.tl = tl; a.start = start; a.end = end; a.offset = offset; a.live = live;
  hipLaunchKernelGGL(window_geometry_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads),
                     0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int tempo_assemble_zslabs(const float* slabs, float* out, int R, long long rows, int Zr,
                          void* stream) {
  if (!slabs || !out) return TEMPO_E_NULL;
  if (R <= 0 || rows <= 0 || Zr <= 0 || (long long)R * Zr > 0x7fffffffLL) return TEMPO_E_DIM;
  if (slabs == out) return TEMPO_E_ARG;
  const bool vec = Zr % 4 == 0 && aligned16(slabs) && aligned16(out);
  const long long total = rows * ((long long)R * Zr / (vec ? 4 : 1));
  long long blocks = (total + kThreads - 1) / kThreads;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 workgroups per CU
  if (vec)
    hipLaunchKernelGGL((assemble_zslabs_kernel<4>), dim3((unsigned)blocks), dim3(kThreads), 0,
                       (hipStream_t)stream, slabs, out, R, rows, Zr);
  else
    hipLaunchKernelGGL((assemble_zslabs_kernel<1>), dim3((unsigned)blocks), dim3(kThreads), 0,
                       (hipStream_t)stream, slabs, out, R, rows, Zr);
  return (int)hipGetLastError();
}

int tempo_gather_zcolumns(const float* cubes, const int64_t* index, float* out, int B, int K,
                          int C, int X, int Y, int Z, void* stream) {
  if (!cubes || !index || !out) return TEMPO_E_NULL;
  if (B <= 0 || K <= 0 || C <= 0 || X <= 0 || Y <= 0 || Z <= 0) return TEMPO_E_DIM;
  if ((long long)B * K > 0x7fffffffLL || (long long)C * Z > 0x7fffffffLL) return TEMPO_E_SIZE;
  hipLaunchKernelGGL(zcolumns_kernel, dim3((unsigned)(B * K)), dim3(kThreads), 0,
                     (hipStream_t)stream, cubes, reinterpret_cast<const long long*>(index), out,
                     K, C, (long long)X * Y, Z);
  return (int)hipGetLastError();
}

int tempo_project_points(const float* pts, const float* cam, int M, int N, float hm_w,
                         float hm_h, float img_w, float img_h, int mm_unfused, float* pix,
                         float* grid, uint8_t* inside, void* stream) {
  if (!pts || !cam || !pix || !grid || !inside) return TEMPO_E_NULL;
  if (M <= 0 || M > 65535 || N <= 0) return TEMPO_E_DIM;
  if (mm_unfused != 0 && mm_unfused != TEMPO_UNFUSED_ALL) return TEMPO_E_ARG;
  dim3 g((N + kThreads - 1) / kThreads, M);
  hipLaunchKernelGGL(project_points_kernel, g, dim3(kThreads), 0, (hipStream_t)stream, pts,
                     cam, N, tempo_make_geom(hm_w, hm_h, img_w, img_h), mm_unfused != 0, pix, grid, inside);
  return (int)hipGetLastError();
}

}  // extern "C"

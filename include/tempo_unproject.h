/* tempo_unproject.h -- C ABI of the MI355X (gfx950) heatmap -> voxel unprojection.
 *
 * One shared library, libtempo_unproject.so, plain C symbols, no C++/torch types.
 * The reference is 100 % Python and has no FFI of its own; each entry point
 * below names the reference code it replaces (paths relative to the
 * reference's root, rccchoudhury/tempo @ v1).
 *
 * Conventions
 *  - every buffer is a caller-allocated DEVICE pointer (hipMalloc / torch
 *    tensor .data_ptr()); no ownership transfer, no hidden allocation, no
 *    host synchronisation; work is enqueued on `stream` (a hipStream_t passed
 *    as void*, NULL = the legacy default stream);
 *  - return value: 0 = ok, < 0 = TEMPO_E_* (bad argument, nothing launched),
 *    > 0 = a hipError_t from the launch.  Nothing throws or exits;
 *  - re-entrant: no mutable globals; concurrent calls on different streams
 *    are safe as long as their output buffers do not overlap (one exception,
 *    stated there: tempo_root_step_static);
 *  - heatmap buffers in channel-last layout (hm, grad_hm, the repack's out, the
 *    root step's workspace) must be 16-byte aligned: they are accessed with
 *    128-bit loads and stores (TEMPO_E_ARG otherwise);
 *  - all floating point is IEEE fp32; the operation order of the per-voxel
 *    arithmetic is fixed (tempo_amd/csrc/unproject_math.h) so that results do
 *    not depend on the launch shape.
 *
 * Packed camera record: 32 floats (128 B) per (frame, view):
 *   [0..8]   R[k][j] row-major; points are row vectors: Xc = Xw @ R + T
 *   [9..11]  T_w2c = -(T^T @ R)
 *   [12..17] K[0][0..2], K[1][0..2]        (pixel = [x y 1] @ K^T)
 *   [18..20] radial k0 k1 k2   [21..22] tangential p0 p1
 *   [23]     != 0 when the distortion terms apply
 *   [24..29] 2x3 image affine (mmpose get_affine_transform(center, scale/200, 0, image_size))
 *   [30] width = center[0]*2   [31] height = center[1]*2
 * (what the reference rebuilds per (batch, view) at project_layer.py:97-110 and
 *  :240-254 through mmpose's SimpleCameraTorch / get_affine_transform.)
 *
 * mm_unfused: rounding flavour of the path's three K=3 matmuls (X @ R,
 * [x y 1] @ K^T, affine @ [x y 1]^T), which the reference runs through the
 * host BLAS: 0 = multiply + two FMAs ascending in k (oneMKL on Intel),
 * TEMPO_MM_UNFUSED = three products and two adds ascending in k (oneMKL on AMD
 * EPYC).  tempo_amd/numerics.py measures the host and picks the matching one.
 */
#ifndef TEMPO_UNPROJECT_H
#define TEMPO_UNPROJECT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEMPO_ABI_VERSION 14
#define TEMPO_CAM_FLOATS 32
#define TEMPO_MAX_VIEWS 32
#define TEMPO_MAX_CHANNELS 32 /* per call; wider tensors go in channel slices */
/* bytes of the per-GPU static staging array of tempo_root_step_static */
#define TEMPO_STATIC_SCRATCH_BYTES (64u << 20)
#define TEMPO_MM_FMA 0
#define TEMPO_MM_UNFUSED 7
/* element types of tempo_repack_nhwc_half's inputs */
#define TEMPO_DTYPE_F32 0
#define TEMPO_DTYPE_F16 1
#define TEMPO_DTYPE_BF16 2

enum {
  TEMPO_OK = 0,
  TEMPO_E_NULL = -1,      /* a required pointer is NULL */
  TEMPO_E_DIM = -2,       /* a dimension is <= 0 or exceeds a limit above */
  TEMPO_E_PAD = -3,       /* padded channel count not a multiple of 4 / < C */
  TEMPO_E_SIZE = -4,      /* one view's map or one volume exceeds 2^31 bytes */
  TEMPO_E_ARG = -5        /* any other inconsistent argument */
};

/* ABI version of the loaded library (== TEMPO_ABI_VERSION of its build). */
int tempo_version(void);

/* Static text for a return code of this library (TEMPO_E_* or hipError_t). */
const char* tempo_error_string(int code);

/* Smallest supported padded channel count for `channels` (a multiple of 4). */
int tempo_padded_channels(int channels);

/* NCHW -> channel-last repack of the V per-view heatmap tensors.
 *   views   : HOST array of V device pointers, each fp32 [B, C_total, H, W]
 *   c_begin : first channel to take; channels [c_begin, c_begin + C)
 *   out     : fp32 [B, V, H, W, CP], channels C..CP-1 zero-filled
 * Replaces nothing in the reference (it samples NCHW with F.grid_sample,
 * project_layer.py:128-133 / :278-284); it exists so that one 16-byte fetch
 * per lane serves four joints of one texel in the kernels below. */
int tempo_repack_nhwc(const void* const* views, int V, int B, int C_total,
                      int c_begin, int C, int H, int W, int CP, float* out,
                      void* stream);

/* The same repack from 16-bit maps: `views` point to [B, C_total, H, W] tensors of
 * IEEE half (dtype = TEMPO_DTYPE_F16) or bfloat16 (TEMPO_DTYPE_BF16); `out` is fp32 as above.
 * Under mixed precision the reference's get_voxel converts its inputs with `.float()` first
 * (@force_fp32(apply_to=('feature_maps', 'centers')), project_layer.py:65, :242); both
 * conversions are exact, so the texels equal those of tempo_repack_nhwc on the converted
 * tensors bit for bit and the fp32 copy is never made.  View pointers must be 2-byte
 * aligned (TEMPO_E_ARG otherwise; 8-byte aligned planes take the wide path); any other
 * dtype value: TEMPO_E_ARG. */
int tempo_repack_nhwc_half(const void* const* views, int dtype, int V, int B,
                           int C_total, int c_begin, int C, int H, int W, int CP,
                           float* out, void* stream);

/* Root-grid unprojection: ProjectLayer.get_voxel, project_layer.py:65-152
 * (agg_method == 'mean').
 *   hm      : [B, V, H, W, CP] from tempo_repack_nhwc
 *   cam     : [B, V, 32] packed camera records
 *   ax_x/y/z: voxel-centre coordinates per axis, fp32 [GB, X] / [GB, Y] / [GB, Z]
 *             with GB = 1 (shared, ax_batched = 0) or GB = B (ax_batched = 1);
 *             = torch.linspace(-S/2, S/2, n) + center of compute_grid(), :43-63
 *   live    : [B] bytes, 0 = batch item skipped (volume, grids, mask stay 0),
 *             the rule of project_layer.py:81; NULL = all live
 *   cubes   : out fp32 [B, C, X, Y, Z]
 *   grids   : out fp32 [B, X*Y*Z, 3] or NULL (the reference's second return)
 *   mask    : out valid-view bitmap [B, X*Y*Z] of mask_bytes-wide words
 *             (1, 2 or 4; bit v = view v passes the bounds test of :113-115),
 *             or NULL with mask_bytes = 0
 *   hm_w,hm_h / img_w,img_h : the module's heatmap_size / image_size */
int tempo_unproject_root(const float* hm, const float* cam, const float* ax_x,
                         const float* ax_y, const float* ax_z, int ax_batched,
                         const uint8_t* live, float* cubes, float* grids,
                         void* mask, int mask_bytes, int B, int V, int C,
                         int CP, int H, int W, int X, int Y, int Z, float hm_w,
                         float hm_h, float img_w, float img_h, int mm_unfused,
                         void* stream);

/* tempo_unproject_root that also emits the volume's maximum over z, the first
 * thing the root detector computes from it (center_net.py:230, torch.max(x,
 * dim=4)).
 *   zmax  : out fp32 [B, C, X, Y], bit-identical to torch.max over the volume
 *           (NaN propagates); cleared on `stream` before the launch where needed.
 *   cubes : out fp32 [B, C, X, Y, Z], or NULL: the volume is then never written.
 *           Its only other consumer in the reference, the K z-columns gathered at
 *           tempo_detector.py:204-206, can be computed without it
 *           (tempo_unproject_root_columns). */
int tempo_unproject_root_zmax(const float* hm, const float* cam, const float* ax_x,
                              const float* ax_y, const float* ax_z, int ax_batched,
                              const uint8_t* live, float* cubes, float* grids,
                              void* mask, int mask_bytes, int B, int V, int C, int CP,
                              int H, int W, int X, int Y, int Z, float hm_w,
                              float hm_h, float img_w, float img_h, int mm_unfused,
                              float* zmax, void* stream);

/* Whole root step for a batch, NCHW maps in, volume out: tempo_repack_nhwc +
 * tempo_unproject_root over chunks of `chunk` frames through a small channel-last
 * `workspace`: choose chunk so that a chunk's maps take <= ~190 MB (16 frames at the
 * Panoptic sizes) and they stay in the MI355X's 256 MB Infinity Cache between the repack
 * that writes them and the gather that reads them.  A batch of more than one chunk uses the
 * workspace as two halves: the repack of chunk c + 1 runs as the last workgroups of the
 * launch that unprojects chunk c (it fills that launch's tail instead of paying its own).
 * Results are those of the two separate calls on the whole batch.
 *   views     : HOST array of V device pointers, fp32 [B, C_total, H, W]
 *   workspace : device fp32 [tempo_root_step_workspace(B, V, H, W, CP, chunk)]
 *               (= min(B, chunk) frames' maps, twice that when B > chunk)
 *   cam, ax_*, live, cubes, grids, mask : as tempo_unproject_root, for all B items */
long long tempo_root_step_workspace(int B, int V, int H, int W, int CP, int chunk);
int tempo_root_step(const void* const* views, int C_total, int c_begin,
                    const float* cam, const float* ax_x, const float* ax_y,
                    const float* ax_z, int ax_batched, const uint8_t* live,
                    float* cubes, float* grids, void* mask, int mask_bytes, int B,
                    int V, int C, int CP, int H, int W, int X, int Y, int Z,
                    float hm_w, float hm_h, float img_w, float img_h,
                    int mm_unfused, float* workspace, int chunk, void* stream);

/* tempo_root_step that also returns the volume's maximum over z (what CenterNet
 * computes first, center_net.py:230), as tempo_unproject_root_zmax does per chunk.
 *   zmax  : out fp32 [B, C, X, Y], bit-identical to torch.max(cubes, dim=4)
 *   cubes : may be NULL (z-max alone, the volume is never written) */
int tempo_root_step_zmax(const void* const* views, int C_total, int c_begin,
                         const float* cam, const float* ax_x, const float* ax_y,
                         const float* ax_z, int ax_batched, const uint8_t* live,
                         float* cubes, float* grids, void* mask, int mask_bytes, int B,
                         int V, int C, int CP, int H, int W, int X, int Y, int Z,
                         float hm_w, float hm_h, float img_w, float img_h,
                         int mm_unfused, float* zmax, float* workspace, int chunk,
                         void* stream);

/* tempo_root_step that allocates nothing and takes no workspace.  NO COUNTERPART IN
 * THE REFERENCE: it exists for callers that cannot call hipMalloc or bring a
 * workspace -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation.  The channel-last staging buffer is a file-scope __device__
 * array of TEMPO_STATIC_SCRATCH_BYTES (64 MiB) that every GPU the library runs on
 * holds until the process ends; its device address is looked up once per device
 * (hipGetSymbolAddress) at the first call made with that device current.  The
 * call itself performs no hipMalloc / hipFree / allocating library call.
 *   chunk : frames per repack+unproject chunk, cut to what fits the array
 *           (6 frames at the Panoptic cam5 sizes, 1 at H36M 4 x 384 x 384 x 20)
 * Limit: one frame's maps, V*H*W*CP*4 bytes, must fit the array (TEMPO_E_SIZE
 * otherwise).  Results are bit-identical to tempo_root_step.  Every call on a GPU
 * shares that GPU's array (it keeps the last call's maps; nothing needs to start at
 * zero, every chunk is fully rewritten before it is read): calls that overlap in
 * time on ONE GPU -- different streams or threads -- race on it and must be
 * serialised by the caller.  All other arguments as tempo_root_step. */
int tempo_root_step_static(const void* const* views, int C_total, int c_begin,
                           const float* cam, const float* ax_x, const float* ax_y,
                           const float* ax_z, int ax_batched, const uint8_t* live,
                           float* cubes, float* grids, void* mask, int mask_bytes,
                           int B, int V, int C, int CP, int H, int W, int X, int Y,
                           int Z, float hm_w, float hm_h, float img_w, float img_h,
                           int mm_unfused, int chunk, void* stream);

/* ---- heatmaps in place: the forward kernels on V channel-last tensors.
 * The reference samples V separate [B, C, H, W] tensors (project_layer.py:128-133,
 * :278-284).  When the 2-D head already produces them channel-last
 * (torch.channels_last: memory order [B][H][W][C]) the repack is unnecessary:
 *   views       : HOST array of V device pointers to frame 0's map of each view
 *   frame_bytes : bytes between consecutive frames of one view
 *   texel_bytes : bytes between consecutive texels = 4 * (channels in memory); the C
 *                 channels read start at the texel's first byte (offset the pointers
 *                 to take a channel slice); rows are W * texel_bytes apart
 *   CP          : tempo_padded_channels(C); the fetch lanes cover CP channels per texel,
 *                 channels C..CP-1 are discarded, so texel_bytes may be as small as 4 * C
 * The caller vouches for the C channels of every texel and nothing else: a map's buffer
 * range ends with channel C - 1 of its last texel, so a channel slice at the end of a
 * tensor is never read past (what the fetch lanes cover beyond it reads as 0 without a
 * memory access).
 * Pointers 4-byte aligned (16 for full fetch speed).  The repacked buffer is the special
 * case views[v] = hm + v*H*W*CP, frame_bytes = V*H*W*CP*4, texel_bytes = CP*4.
 * Everything else as tempo_unproject_root / tempo_unproject_window. */
int tempo_unproject_root_views(const void* const* views, long long frame_bytes,
                               int texel_bytes, const float* cam, const float* ax_x,
                               const float* ax_y, const float* ax_z, int ax_batched,
                               const uint8_t* live, float* cubes, float* grids,
                               void* mask, int mask_bytes, int B, int V, int C, int CP,
                               int H, int W, int X, int Y, int Z, float hm_w,
                               float hm_h, float img_w, float img_h, int mm_unfused,
                               void* stream);
/* tempo_unproject_root_views that also returns the volume's maximum over z, as
 * tempo_unproject_root_zmax does (center_net.py:230):
 *   zmax  : out fp32 [B, C, X, Y], bit-identical to torch.max(cubes, dim=4)
 *   cubes : may be NULL (z-max alone) */
int tempo_unproject_root_views_zmax(const void* const* views, long long frame_bytes,
                                    int texel_bytes, const float* cam, const float* ax_x,
                                    const float* ax_y, const float* ax_z, int ax_batched,
                                    const uint8_t* live, float* cubes, float* grids,
                                    void* mask, int mask_bytes, int B, int V, int C, int CP,
                                    int H, int W, int X, int Y, int Z, float hm_w,
                                    float hm_h, float img_w, float img_h, int mm_unfused,
                                    float* zmax, void* stream);
int tempo_unproject_window_views(const void* const* views, long long frame_bytes,
                                 int texel_bytes, const float* cam,
                                 const int32_t* win_frame, const float* fx,
                                 const float* fy, const float* fz, const int32_t* tl,
                                 const int32_t* start, const int32_t* end,
                                 const uint8_t* live, float* cubes, int NW, int F,
                                 int V, int C, int CP, int H, int W, int FX, int FY,
                                 int FZ, int VX, int VY, int VZ, float hm_w,
                                 float hm_h, float img_w, float img_h, int mm_unfused,
                                 void* stream);
/* tempo_unproject_window_planes on maps read in place (arguments as
 * tempo_unproject_window_views + the planes of tempo_unproject_window_planes; cubes may be
 * NULL): the pose regressor's three axis-max planes (tempo_pose.py:172-176) straight from
 * channel-last maps, no repack and no volume. */
int tempo_unproject_window_planes_views(const void* const* views, long long frame_bytes,
                                        int texel_bytes, const float* cam,
                                        const int32_t* win_frame, const float* fx,
                                        const float* fy, const float* fz,
                                        const int32_t* tl, const int32_t* start,
                                        const int32_t* end, const uint8_t* live,
                                        float* cubes, float* plane_xy, float* plane_xz,
                                        float* plane_yz, int NW, int F, int V, int C,
                                        int CP, int H, int W, int FX, int FY, int FZ,
                                        int VX, int VY, int VZ, float hm_w, float hm_h,
                                        float img_w, float img_h, int mm_unfused,
                                        void* stream);

/* tempo_unproject_root for a batch in which runs of consecutive items share one
 * calibration and one grid -- the frames of one recording; the reference
 * re-projects the grid for every item (project_layer.py:80-111).
 *   groups : DEVICE int32 [num_groups, 2] = (first item, count); the groups
 *            partition [0, B).  The caller guarantees that the items of a group
 *            have bit-identical camera records cam[item] and, with ax_batched,
 *            identical axes; only their heatmaps (and live flags) differ.  The
 *            projection then runs once per group and voxel; results are
 *            bit-identical to tempo_unproject_root.  NULL / 0 = no sharing. */
int tempo_unproject_root_grouped(const float* hm, const float* cam,
                                 const float* ax_x, const float* ax_y,
                                 const float* ax_z, int ax_batched,
                                 const uint8_t* live, float* cubes, float* grids,
                                 void* mask, int mask_bytes, int B, int V, int C,
                                 int CP, int H, int W, int X, int Y, int Z,
                                 float hm_w, float hm_h, float img_w, float img_h,
                                 int mm_unfused, const int32_t* groups,
                                 int num_groups, void* stream);

/* Per-person window unprojection: ProjectLayerWithMask.get_voxel +
 * project_grid, project_layer.py:225-378 (agg_method == 'mean').  Only the
 * window's own voxels are projected (the reference projects the whole fine
 * grid and slices it, :255-276).
 *   hm        : [F, V, H, W, CP]; cam : [F, V, 32]
 *   win_frame : [NW] int32 frame of each window, or NULL (window i uses frame
 *               i, NW == F: the reference's one proposal per batch item)
 *   fx/fy/fz  : fine-grid axes over the whole space, fp32 [FX] [FY] [FZ]
 *               (= compute_grid(space_size, space_center, fine_voxels_per_axis))
 *   tl,start,end : int32 [NW,3]  centers_tl, start, end of :299-324
 *   live      : [NW] bytes or NULL, the rule of :232
 *   cubes     : out fp32 [NW, C, VX, VY, VZ]; voxels outside [start,end) are 0 */
int tempo_unproject_window(const float* hm, const float* cam,
                           const int32_t* win_frame, const float* fx,
                           const float* fy, const float* fz, const int32_t* tl,
                           const int32_t* start, const int32_t* end,
                           const uint8_t* live, float* cubes, int NW, int F,
                           int V, int C, int CP, int H, int W, int FX, int FY,
                           int FZ, int VX, int VY, int VZ, float hm_w,
                           float hm_h, float img_w, float img_h, int mm_unfused,
                           void* stream);

/* tempo_unproject_window for runs of consecutive windows that share tl/start/end
 * and the camera records of their frames -- the T timesteps of one proposal
 * (tempo_pose.py:145-156 calls the layer once per timestep with the same
 * candidate).  groups as in tempo_unproject_root_grouped, partitioning [0, NW). */
int tempo_unproject_window_grouped(const float* hm, const float* cam,
                                   const int32_t* win_frame, const float* fx,
                                   const float* fy, const float* fz,
                                   const int32_t* tl, const int32_t* start,
                                   const int32_t* end, const uint8_t* live,
                                   float* cubes, int NW, int F, int V, int C,
                                   int CP, int H, int W, int FX, int FY, int FZ,
                                   int VX, int VY, int VZ, float hm_w, float hm_h,
                                   float img_w, float img_h, int mm_unfused,
                                   const int32_t* groups, int num_groups,
                                   void* stream);

/* tempo_unproject_window with the pose regressor's axis-max fused in: the only
 * consumer of the window volume takes its maximum along z, y and x
 * (tempo_pose.py:172-176: torch.max(cubes, dim=5 / 4 / 3)).
 *   cubes    : out fp32 [NW, C, VX, VY, VZ], or NULL to skip the volume
 *   plane_xy : out fp32 [NW, C, VX, VY] = max over z, or NULL
 *   plane_xz : out fp32 [NW, C, VX, VZ] = max over y, or NULL
 *   plane_yz : out fp32 [NW, C, VY, VZ] = max over x, or NULL
 * Values are bit-identical to torch.max over the volume tempo_unproject_window
 * writes; a NaN voxel makes its three plane entries NaN (as torch.max does; the
 * NaN's payload bits are not specified).  The planes are cleared on `stream`
 * before the launch. */
int tempo_unproject_window_planes(const float* hm, const float* cam,
                                  const int32_t* win_frame, const float* fx,
                                  const float* fy, const float* fz,
                                  const int32_t* tl, const int32_t* start,
                                  const int32_t* end, const uint8_t* live,
                                  float* cubes, float* plane_xy, float* plane_xz,
                                  float* plane_yz, int NW, int F, int V, int C,
                                  int CP, int H, int W, int FX, int FY, int FZ,
                                  int VX, int VY, int VZ, float hm_w, float hm_h,
                                  float img_w, float img_h, int mm_unfused,
                                  void* stream);

/* ---- training: forward that records the clamp mask, and the backward pass.
 * The reference trains through this layer (freeze_2d = False,
 * configs/.../resnet_rnn_panoptic_cam5.py:133): autograd differentiates
 * project_layer.py:128-151 (grid_sample -> * bounding -> sum / (count + 1e-6) ->
 * clamp) and :278-378 with respect to the feature maps; cameras and grids carry
 * no gradient.
 *
 * tempo_clamp_mask_words: 64-bit words of clamp mask per batch item / window for
 * a grid of num_voxels voxels and CP padded channels (< 0: TEMPO_E_*).  The mask
 * says where min <= x <= max held at the final clamp (torch.clamp's backward
 * rule); its layout is private to the library. */
int tempo_clamp_mask_words(int num_voxels, int CP);

/* tempo_unproject_root / _window that also write clamp_mask
 * [B or NW][tempo_clamp_mask_words(X*Y*Z, CP)] (device). */
int tempo_unproject_root_train(const float* hm, const float* cam, const float* ax_x,
                               const float* ax_y, const float* ax_z, int ax_batched,
                               const uint8_t* live, float* cubes, float* grids,
                               void* mask, int mask_bytes, int B, int V, int C, int CP,
                               int H, int W, int X, int Y, int Z, float hm_w,
                               float hm_h, float img_w, float img_h, int mm_unfused,
                               uint64_t* clamp_mask, void* stream);
int tempo_unproject_window_train(const float* hm, const float* cam,
                                 const int32_t* win_frame, const float* fx,
                                 const float* fy, const float* fz, const int32_t* tl,
                                 const int32_t* start, const int32_t* end,
                                 const uint8_t* live, float* cubes, int NW, int F,
                                 int V, int C, int CP, int H, int W, int FX, int FY,
                                 int FZ, int VX, int VY, int VZ, float hm_w,
                                 float hm_h, float img_w, float img_h, int mm_unfused,
                                 uint64_t* clamp_mask, void* stream);

/* Backward of the two calls above with respect to the heatmaps.
 *   grad_cubes : fp32 [B, C, X, Y, Z] / [NW, C, VX, VY, VZ], gradient of the loss
 *                with respect to the volume
 *   clamp_mask : as written by the matching *_train call (same geometry)
 *   grad_hm    : out fp32 [B or F, V, H, W, CP] channel-last; cleared on `stream`,
 *                then accumulated with fp32 atomics (windows of one frame add
 *                up), so sums agree with autograd to rounding, not bit for bit.
 *                tempo_unrepack_nchw turns it into per-view NCHW gradients.
 * All other arguments as in the forward calls.  Any number of views up to
 * TEMPO_MAX_VIEWS: the root backward (and the window backward for maps wider or
 * taller than 16384 texels) keeps the per-view footprints of a tile in LDS, 11
 * views per launch at CP = 16 (10 at 20, 9 at 24, 8 at 28, 7 at 32); more views go
 * through in several launches on `stream`.  The default window backward keeps no
 * per-view records; it sums the footprints of a 4x4x4 brick of voxels in an LDS tile of
 * tile_texels texels before they go to memory: 0 (= 121, for lattices up to about two
 * heatmap pixels per voxel), 121 or 144 (coarser lattices, e.g. H36M windows on 384 x 384
 * maps at 2.8 pixels per voxel); a speed choice only, TEMPO_E_ARG for other values. */
int tempo_unproject_root_backward(const float* grad_cubes, const float* cam,
                                  const float* ax_x, const float* ax_y,
                                  const float* ax_z, int ax_batched,
                                  const uint8_t* live, const uint64_t* clamp_mask,
                                  float* grad_hm, int B, int V, int C, int CP, int H,
                                  int W, int X, int Y, int Z, float hm_w, float hm_h,
                                  float img_w, float img_h, int mm_unfused,
                                  void* stream);
int tempo_unproject_window_backward(const float* grad_cubes, const float* cam,
                                    const int32_t* win_frame, const float* fx,
                                    const float* fy, const float* fz,
                                    const int32_t* tl, const int32_t* start,
                                    const int32_t* end, const uint8_t* live,
                                    const uint64_t* clamp_mask, float* grad_hm,
                                    int NW, int F, int V, int C, int CP, int H, int W,
                                    int FX, int FY, int FZ, int VX, int VY, int VZ,
                                    float hm_w, float hm_h, float img_w, float img_h,
                                    int mm_unfused, int tile_texels, void* stream);

/* Training THROUGH the fused window planes: the pose regressor takes torch.max of the window
 * volume over z, y and x (tempo_pose.py:172-176) inside forward_train (freeze_2d = False),
 * so autograd sends each plane entry's gradient to one voxel of the volume -- the first one
 * along the reduced axis that attains the maximum (CPU torch.max; a NaN, once met, is the
 * maximum) -- through torch.clamp's backward (gradient passes where 0 <= x <= 1) and the
 * masked mean into the heatmaps (project_layer.py:278-378).  These two calls do that without
 * the [NW, C, VX, VY, VZ] volume and its gradient volume ever existing:
 *
 * tempo_window_planes_claims: given the FINAL planes of tempo_unproject_window_planes on the
 *   same inputs (plane_xy [NW,C,VX,VY], plane_xz [NW,C,VX,VZ], plane_yz [NW,C,VY,VZ]),
 *   recompute every voxel and record per plane entry the claim of the voxel its gradient
 *   goes to: int32 claim_* in the planes' layouts, 2 * (index along the reduced axis) + 1 if
 *   the clamp blocks the gradient there.  Entries of skipped windows keep 0x7f7f7f7f.
 *   hm is the repacked [F, V, H, W, CP] buffer; every other argument as
 *   tempo_unproject_window_planes.  All six plane / claim pointers are required.
 *
 * tempo_unproject_window_planes_backward: grad_xy / grad_xz / grad_yz (the planes'
 *   incoming gradients, fp32, planes' layouts) + those claims -> grad_hm [F, V, H, W, CP]
 *   (zeroed, then accumulated with fp32 atomics like tempo_unproject_window_backward, whose
 *   kernel it shares; tile_texels 0 (= 81), 81, 121 or 144: the footprint boxes here are
 *   those of a brick's claimed voxels alone, so the small tile is the fast one).  Maps wider
 *   or taller than 16384 texels:
 *   TEMPO_E_SIZE.  Follow with tempo_unrepack_nchw for per-view NCHW gradients. */
int tempo_window_planes_claims(const float* hm, const float* cam,
                               const int32_t* win_frame, const float* fx,
                               const float* fy, const float* fz, const int32_t* tl,
                               const int32_t* start, const int32_t* end,
                               const uint8_t* live, const float* plane_xy,
                               const float* plane_xz, const float* plane_yz,
                               int32_t* claim_xy, int32_t* claim_xz, int32_t* claim_yz,
                               int NW, int F, int V, int C, int CP, int H, int W,
                               int FX, int FY, int FZ, int VX, int VY, int VZ,
                               float hm_w, float hm_h, float img_w, float img_h,
                               int mm_unfused, void* stream);
int tempo_unproject_window_planes_backward(
    const float* grad_xy, const float* grad_xz, const float* grad_yz,
    const int32_t* claim_xy, const int32_t* claim_xz, const int32_t* claim_yz,
    const float* cam, const int32_t* win_frame, const float* fx, const float* fy,
    const float* fz, const int32_t* tl, const int32_t* start, const int32_t* end,
    const uint8_t* live, float* grad_hm, int NW, int F, int V, int C, int CP, int H,
    int W, int FX, int FY, int FZ, int VX, int VY, int VZ, float hm_w, float hm_h,
    float img_w, float img_h, int mm_unfused, int tile_texels, void* stream);

/* Inverse of tempo_repack_nhwc: channel-last [B, V, H, W, CP] -> channels
 * [c_begin, c_begin + C) of V per-view fp32 [B, C_total, H, W] tensors
 * (views: HOST array of V device pointers).  Other channels are not touched. */
int tempo_unrepack_nchw(const float* in, void* const* views, int V, int B,
                        int C_total, int c_begin, int C, int H, int W, int CP,
                        void* stream);

/* The 2-D head's 1x1 convolution with the unprojection's input layout as its output:
 * rnn_multiview_pose.py:61-66 (process_layer = nn.Conv2d(256, num_features, 1)) applied
 * at :321 to the backbone map of all T*V*B images, then chunked into
 * feature_maps[t][v] = [B, C, H, W] at :324-326.  Writing channel-last directly removes
 * the NCHW tensor and the repack (tempo_repack_nhwc) that would read it back.
 *   in     : fp32 [T*V*B, K, H, W], images ordered (t, v, b) as torch.cat builds them
 *   weight : fp32 [C, K] (the conv's weight[:, :, 0, 0]);  bias : fp32 [C] or NULL
 *   out    : fp32 [T*B, V, H, W, CP], frame = t * B + b; channels C..CP-1 are zero;
 *            16-byte aligned.  The result of frames [t*B, (t+1)*B) is the `hm` argument of
 *            tempo_unproject_root / _window for timestep t.
 * fp32 MFMA (v_mfma_f32_16x16x4_f32): each output is an fp32 FMA chain over k ascending,
 * then + bias -- the same precision class as F.conv2d in fp32, not bit-identical to it
 * (libraries sum in their own order).  K * 16 * ceil(CP / 16) * 4 bytes of weights must
 * fit 64 KB of LDS (K <= 1024 at CP = 16, 512 at CP = 32: TEMPO_E_DIM beyond). */
int tempo_conv1x1_nhwc(const float* in, const float* weight, const float* bias, float* out,
                       int T, int V, int B, int K, int C, int CP, int H, int W,
                       void* stream);

/* Integer window geometry of ProjectLayerWithMask.get_voxel, project_layer.py:299-324, for
 * all windows of a call in one launch: centers_tl = round(gc[:, :3] * scale + bias).int(),
 * offset = centers_tl / (fine - 1) * space - space / 2 + sub / 2, the bbox mask
 * ((1 - gc[:, 5:7]) / 2 * (vox[:2] - 1)).int() clamped at 0, start / end (:316-324), and the
 * skip flag gc[:, 3] >= 0 (:232).  Every fp32 step is rounded like the reference's separate
 * tensor ops: results are bit-identical to them.
 *   grid_center : fp32 [B, cols], cols == 3 (no bbox, no skip rule) or >= 7
 *   scale, bias, space_size, sub_space_size : DEVICE fp32 [3] (the module's own tensors,
 *                 set_space_size :184-201)
 *   tl, start, end : int32 [B, 3];  offset : fp32 [B, 3];  live : uint8 [B] or NULL */
int tempo_window_geometry(const float* grid_center, int B, int cols, const float* scale,
                          const float* bias, const float* space_size,
                          const float* sub_space_size, int FX, int FY, int FZ, int VX, int VY,
                          int VZ, int32_t* tl, int32_t* start, int32_t* end, float* offset,
                          uint8_t* live, void* stream);

/* z-columns of the root volume at K (x, y) cells per frame -- what the detector's 1-D
 * stage reads (tempo_detector.py:204-206: torch.gather(torch.flatten(cubes, 2, 3)
 * .permute(0, 2, 1, 3), dim=1, index=topk_2d_flatten_index...repeat(1, 1, C, Z))).
 *   cubes : fp32 [B, C, X, Y, Z];  index : int64 [B, K], cell = x * Y + y
 *   out   : fp32 [B, K, C, Z];  a cell outside [0, X*Y) gives zeros (torch raises) */
int tempo_gather_zcolumns(const float* cubes, const int64_t* index, float* out, int B,
                          int K, int C, int X, int Y, int Z, void* stream);

/* The same z-columns computed directly from the heatmaps: only the K * Z voxels of the K
 * cells are unprojected (per-voxel arithmetic of tempo_unproject_root, hence bit-identical
 * to tempo_gather_zcolumns on the volume that call writes), so the root detector
 * (tempo_detector.py:187-206: volume -> z-max -> 2-D net -> top-K cells -> z-columns -> 1-D
 * net) never needs the [B, C, X, Y, Z] volume in memory: tempo_*_zmax with cubes = NULL
 * first, this call once the cells are known.
 *   index : int64 [B, K], cell = x * Y + y; a cell outside [0, X*Y) gives zeros
 *   out   : fp32 [B, K, C, Z];  live : [B] bytes or NULL (a skipped item's columns are 0)
 * Three forms by where the maps are: the repacked buffer hm [B, V, H, W, CP]; channel-last
 * tensors in place (arguments as tempo_unproject_root_views); the reference's own V NCHW
 * tensors [B, C_total, H, W] (arguments as tempo_root_step, no workspace: the few texels
 * needed are read plane by plane). */
int tempo_unproject_root_columns(const float* hm, const float* cam, const float* ax_x,
                                 const float* ax_y, const float* ax_z, int ax_batched,
                                 const uint8_t* live, const int64_t* index, float* out, int B,
                                 int K, int V, int C, int CP, int H, int W, int X, int Y, int Z,
                                 float hm_w, float hm_h, float img_w, float img_h,
                                 int mm_unfused, void* stream);
int tempo_unproject_root_columns_views(const void* const* views, long long frame_bytes,
                                       int texel_bytes, const float* cam, const float* ax_x,
                                       const float* ax_y, const float* ax_z, int ax_batched,
                                       const uint8_t* live, const int64_t* index, float* out,
                                       int B, int K, int V, int C, int CP, int H, int W, int X,
                                       int Y, int Z, float hm_w, float hm_h, float img_w,
                                       float img_h, int mm_unfused, void* stream);
int tempo_root_step_columns(const void* const* views, int C_total, int c_begin,
                            const float* cam, const float* ax_x, const float* ax_y,
                            const float* ax_z, int ax_batched, const uint8_t* live,
                            const int64_t* index, float* out, int B, int K, int V, int C,
                            int CP, int H, int W, int X, int Y, int Z, float hm_w, float hm_h,
                            float img_w, float img_h, int mm_unfused, void* stream);

/* Multi-GPU dense grids (BASELINE config 5; no counterpart in the reference, whose only
 * parallelism is DDP, tools/dist_train.sh:12-20): rank r unprojects planes [r*Zr, (r+1)*Zr)
 * of the grid (tempo_unproject_root with a shorter z axis), ONE all-gather over RCCL leaves
 * slabs [R][rows][Zr] on every rank (rows = B*C*X*Y), and this call moves the rank index
 * inside the row:  out [rows][R*Zr] = the volume [B, C, X, Y, Z].  slabs != out. */
int tempo_assemble_zslabs(const float* slabs, float* out, int R, long long rows, int Zr,
                          void* stream);

/* Projection only, for parity checks of the coordinate chain
 * (SimpleCameraTorch.world_to_pixel + project_layer.py:113-126):
 *   pts [N,3] world points, cam [M,32]  ->  pix [M,N,2] source-image pixels,
 *   grid [M,N,2] grid_sample coordinates, inside [M,N] bytes (bounds test). */
int tempo_project_points(const float* pts, const float* cam, int M, int N,
                         float hm_w, float hm_h, float img_w, float img_h,
                         int mm_unfused, float* pix, float* grid, uint8_t* inside,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TEMPO_UNPROJECT_H */

// LDS float atomic add rate on gfx950: ds_add_f32 (no return), 64 distinct consecutive words per
// wave-instruction (no bank conflict, no same-address lanes), 16 waves per CU hammering one
// shared tile -- against plain ds_read + add + ds_write of the same words.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_rate.hip -o build_dev/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int tile_words) {
  extern __shared__ float tile[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < tile_words; i += blockDim.x) tile[i] = 0.0f;
  __syncthreads();
  unsigned pos = (unsigned)(wave * 977 + blockIdx.x * 131) * 64u;
  const float v = 1.0f + lane * 0.001f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const unsigned a = (pos + (unsigned)lane) % (unsigned)tile_words;
      if (MODE == 0) {
        __hip_atomic_fetch_add(&tile[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 1) {
        tile[a] += v;  // plain read-add-write (races between waves: timing only)
      } else if (MODE == 3) {  // integer max (what the fused axis-max planes use)
        __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(&tile[a]), __float_as_uint(v) + u,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 4) {  // plain store
        tile[a] = v;
      } else {  // every lane of the instruction on ONE word per 16 lanes (same-address lanes)
        __hip_atomic_fetch_add(&tile[(a & ~63u) + (lane >> 4)], v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      pos += 64u * 7u;
    }
  }
  __syncthreads();
  if (tid < 64) out[blockIdx.x * 64 + tid] = tile[tid];
}

template <int MODE>
void run(const char* name, int waves, int tile_words) {
  float* out;
  hipMalloc(&out, 256 * 8 * 64 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(256), block(waves * 64);
  hipLaunchKernelGGL(k<MODE>, grid, block, tile_words * 4, 0, out, 10, tile_words);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, block, tile_words * 4, 0, out, iters, tile_words);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 8 * waves;  // wave-instructions per CU
  printf("%-28s %2d waves/CU, tile %6d B: %.3f ms -> %.1f cycles per 64-lane instruction per CU (at 2.4 GHz)\n",
         name, waves, tile_words * 4, ms, ms * 1e-3 * 2.4e9 / instr);
  hipFree(out);
}

int main() {
  for (int waves : {4, 8, 16}) {
    run<0>("ds_add_f32 distinct words", waves, 32768);
    run<1>("plain read+add+write", waves, 32768);
    run<2>("ds_add_f32 16 lanes per word", waves, 32768);
    run<3>("ds_max_u32 distinct words", waves, 32768);
    run<4>("plain ds_write_b32", waves, 32768);
  }
  return 0;
}

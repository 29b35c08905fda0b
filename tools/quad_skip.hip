// Micro-benchmark: does the texture path's time per 16-byte-per-lane buffer load depend on how
// many of the wave's 16 quads (4 lanes = one 64-byte texel) actually fetch?  Quads are switched
// off either by an out-of-range offset (the buffer descriptor's range check returns 0 without a
// memory access -- how the kernels mark texels they do not need) or by the EXEC mask.
//   If time scales with the ACTIVE quads, skipping texels a neighbouring voxel already holds
//   (window grid: adjacent voxels share rows) would pay; if it does not, it cannot.
// Build: hipcc --offload-arch=gfx950 -O3 tools/quad_skip.hip -o build_dev/quad_skip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: all quads fetch; 1: quads with (quad & mask) != 0 get an out-of-range offset;
// 2: those quads are EXEC-masked off
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void gather(const char* base, unsigned bytes, unsigned span_mask,
                                              int iters, unsigned off_mask, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned quad = (unsigned)lane >> 2;
  const unsigned grp = (blockIdx.x * 256u + threadIdx.x) >> 2;
  const unsigned sub = (unsigned)(lane & 3) * 16u;
  const bool off = (quad & off_mask) != 0;
  unsigned s = grp * 2654435761u + 12345u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(base), 0, (int)bytes, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      s = s * 1664525u + 1013904223u;
      unsigned o = (((s >> 4) * 64u) & span_mask) + sub;
      if (MODE == 1 && off) o = 0xfffffff0u;
      if (MODE == 2) {
        v[u] = u32x4{0, 0, 0, 0};
        if (!off) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
      } else {
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = 1;
}

template <int MODE, int UNR>
void run(const char* name, const char* buf, size_t span, int iters, unsigned off_mask, int active,
         unsigned* sink, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned sm = (unsigned)(span - 1) & ~63u;
  gather<MODE, UNR><<<blocks, 256>>>(buf, (unsigned)span, sm, 4, off_mask, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  gather<MODE, UNR><<<blocks, 256>>>(buf, (unsigned)span, sm, iters, off_mask, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double)blocks * 4 * iters * UNR;  // wave-level load instructions
  printf("%-34s span %6zu KB  active quads %2d/16  %7.3f ms  %6.3f instr/clk/CU  %6.3f texels/clk/CU\n",
         name, span >> 10, active, ms, instr / (ms * 1e-3) / 2.4e9 / 256,
         instr * active / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
  const size_t cap = 64ull << 20;
  char* buf;
  unsigned* sink;
  hipMalloc(&buf, cap);
  hipMalloc(&sink, 4);
  hipMemset(buf, 1, cap);
  const int blocks = 256 * 16;
  for (size_t span : {(size_t)16 << 10, (size_t)8 << 20}) {
    run<0, 8>("all quads", buf, span, 32, 0, 16, sink, blocks);
    run<1, 8>("odd quads out of range", buf, span, 32, 1, 8, sink, blocks);
    run<2, 8>("odd quads EXEC-masked", buf, span, 32, 1, 8, sink, blocks);
    run<1, 8>("3 of 4 quads out of range", buf, span, 32, 3, 4, sink, blocks);
    run<2, 8>("3 of 4 quads EXEC-masked", buf, span, 32, 3, 4, sink, blocks);
    run<1, 8>("upper half out of range", buf, span, 32, 8, 8, sink, blocks);
    run<2, 8>("upper half EXEC-masked", buf, span, 32, 8, 8, sink, blocks);
  }
  return 0;
}

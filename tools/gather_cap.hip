// Micro-benchmark: how many random L1-missing texel reads per cycle and CU does
// gfx950 sustain?  (Is the root kernel's gather at the hardware's rate?)
//   lanes per access: 4 x 16 B = one 64-byte texel, or 8 x 16 B = one aligned
//   128-byte texel pair; addresses pseudo-random over `span` bytes (L2-resident
//   when span <= 2 MB per XCD), UNR loads in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_cap.hip -o build_dev/gather_cap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LPA, int UNR>
__global__ __launch_bounds__(256) void gather(const char* base, unsigned span_mask, int iters,
                                              unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned grp = (blockIdx.x * 256u + threadIdx.x) / LPA;
  const unsigned sub = (unsigned)(lane % LPA) * 16u;
  unsigned s = grp * 2654435761u + 12345u;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      s = s * 1664525u + 1013904223u;
      const unsigned off = ((s >> 4) * (LPA * 16u)) & span_mask;
      v[u] = *reinterpret_cast<const u32x4*>(base + off + sub);
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = 1;
}

template <int LPA, int UNR>
void run(const char* name, const char* buf, size_t span, int iters, unsigned* sink, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  gather<LPA, UNR><<<blocks, 256>>>(buf, (unsigned)(span - 1) & ~(LPA * 16u - 1), 4, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  gather<LPA, UNR><<<blocks, 256>>>(buf, (unsigned)(span - 1) & ~(LPA * 16u - 1), iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double acc = (double)blocks * 256 / LPA * iters * UNR;  // accesses of LPA*16 bytes
  printf("%-28s span %8zu KB  %7.3f ms  %7.2f G acc/s  %6.2f TB/s  %.3f acc/clk/CU (2.4 GHz, 256 CU)\n",
         name, span >> 10, ms, acc / ms / 1e6, acc * LPA * 16 / ms / 1e9,
         acc / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
  const size_t cap = 1ull << 30;
  char* buf;
  unsigned* sink;
  hipMalloc(&buf, cap);
  hipMalloc(&sink, 4);
  hipMemset(buf, 1, cap);
  const int blocks = 256 * 16;
  for (size_t span : {(size_t)16 << 10, (size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20,
                      (size_t)1 << 30}) {
    run<4, 4>("64B x4 in flight", buf, span, 64, sink, blocks);
    run<4, 16>("64B x16 in flight", buf, span, 16, sink, blocks);
    run<8, 8>("128B x8 in flight", buf, span, 32, sink, blocks);
    run<16, 8>("256B x8 in flight", buf, span, 32, sink, blocks);
  }
  return 0;
}

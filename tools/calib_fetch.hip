// Calibration of rocprofv3's FETCH_SIZE for THIS path's access shape (MI355X guide, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Four lanes read one 64-byte chunk (16 B per lane, buffer-load shaped like the
// texel fetch); every chunk of a buffer far larger than L2 + Infinity Cache is read exactly once.
//   mode 0: chunk order = a pseudo-random permutation at 64-byte granularity (line mates apart)
//   mode 1: permutation at 128-byte granularity (both halves of a line by neighbouring quads)
//   mode 2: fully sequential (the wide coalesced streaming read of the guide)
// build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o build_dev/calib_fetch
// run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- build_dev/calib_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ void gather(const uint4* __restrict__ src, uint4* __restrict__ sink, unsigned nchunks,
                       unsigned mult) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned q = t >> 2, part = t & 3;  // quad -> chunk, lane -> 16 B of it
  if (q >= nchunks) return;
  unsigned c;
  if (MODE == 0) c = (unsigned)(((unsigned long long)q * mult) % nchunks);  // mult coprime to nchunks
  else if (MODE == 1) {
    const unsigned line = (unsigned)(((unsigned long long)(q >> 1) * mult) % (nchunks >> 1));
    c = line * 2 + (q & 1);
  } else c = q;
  const uint4 v = src[(size_t)c * 4 + part];
  if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[t & 1023] = v;  // never true: keep the load
}

int main() {
  const size_t bytes = 2ull << 30;  // 2 GiB
  const unsigned nchunks = (unsigned)(bytes / 64);  // 2^25: any odd multiplier is coprime
  uint4 *src, *sink;
  hipMalloc(&src, bytes);
  hipMalloc(&sink, 1024 * sizeof(uint4));
  hipMemset(src, 0, bytes);
  const unsigned threads = 256, blocks = (unsigned)(((size_t)nchunks * 4 + threads - 1) / threads);
  const unsigned mult = 2654435761u | 1u;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(threads), 0, 0, src, sink, nchunks, mult);
    hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(threads), 0, 0, src, sink, nchunks, mult);
    hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(threads), 0, 0, src, sink, nchunks, mult);
  }
  hipDeviceSynchronize();
  printf("true bytes per launch: %zu\n", bytes);
  return 0;
}

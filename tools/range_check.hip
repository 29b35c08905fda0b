// How does gfx950 range-check a 16-byte raw buffer load whose dwords straddle the end of the
// buffer's range (num_records)?  apply_layout() in tempo_unproject.hip ends a map's range with
// the last channel the caller vouches for, which can fall inside a lane's 16-byte quad; the
// kernels rely on the check being per dword (in-range dwords load, the rest read as 0 and
// cause no memory access).  Prints what a b128 load at offset 0 returns for ranges of 0..20
// bytes over a buffer of 1, 2, 3, ... and for offsets that start inside / past the range.
//   hipcc --offload-arch=gfx950 -O2 tools/range_check.hip -o build_dev/range_check
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* buf, int range, unsigned offset, unsigned* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(buf), 0, range, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, offset, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
}

int main() {
  float host[16];
  for (int i = 0; i < 16; i++) host[i] = (float)(i + 1);
  float* buf;
  unsigned* out;
  if (hipMalloc(&buf, sizeof(host)) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 1;
  hipMemcpy(buf, host, sizeof(host), hipMemcpyHostToDevice);
  int bad = 0;
  for (int range = 0; range <= 24; range += 4) {
    for (unsigned offset = 0; offset <= 16; offset += 4) {
      probe<<<1, 64>>>(buf, range, offset, out);
      unsigned got[4];
      if (hipMemcpy(got, out, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
      printf("range %2d offset %2u:", range, offset);
      for (int j = 0; j < 4; j++) {
        float f;
        __builtin_memcpy(&f, &got[j], 4);
        const unsigned byte = offset + 4 * j;
        const float want = byte + 4 <= (unsigned)range ? host[byte / 4] : 0.0f;
        printf(" %g%s", f, f == want ? "" : "(!)");
        bad += f != want;
      }
      printf("\n");
    }
  }
  printf(bad ? "NOT per dword: %d lanes differ\n" : "per-dword range check: ok\n", bad);
  return bad != 0;
}

// Micro-probe (round 4): what does the store PATTERN of the SP16 epilogues cost?  A persistent grid writes a 256 MB buffer with 16-byte
// stores per lane in three patterns:
//   A  contiguous: a wave-instruction writes 1 KB of consecutive bytes (the ideal)
//   B  SP16 epilogue: lane (li = lane & 31, kh = lane >> 5) writes 16 B at pixel li * 64 + q * 32 + kh * 16; the two instructions q = 0, 1
//      that complete a pixel's 64-byte record are issued back to back (conv3x3_sp.hip: "the four pieces ... complete the same lines")
//   C  like B but the two halves of a record are issued `gap` other stores apart (what a spread-out epilogue would do)
// build: hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip ; run: ./write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void wr(char* out, size_t bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wg = bytes / gridDim.x;               // multiple of 64 KB
  char* base = out + (size_t)blockIdx.x * per_wg;
  const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  // a "tile" = 8 waves x 2 KB (32 pixel records per wave)
  for (size_t t = 0; t + 16384 <= per_wg; t += 16384) {
    char* w = base + t + wave * 2048;
    if (MODE == 0) {
      *reinterpret_cast<u32x4*>(w + lane * 16) = v;
      *reinterpret_cast<u32x4*>(w + 1024 + lane * 16) = v;
    } else {
      const int li = lane & 31, kh = lane >> 5;
      *reinterpret_cast<u32x4*>(w + li * 64 + 0 * 32 + kh * 16) = v;
      if (MODE == 2) __builtin_amdgcn_s_sleep(8);
      *reinterpret_cast<u32x4*>(w + li * 64 + 1 * 32 + kh * 16) = v;
    }
  }
}
int main() {
  const size_t bytes = (size_t)256 << 20;
  char* d; hipMalloc(&d, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"A contiguous 1 KB per instruction", "B SP16 halves back to back", "C SP16 halves apart (s_sleep 8)"};
  for (int rep = 0; rep < 2; ++rep)
  for (int m = 0; m < 3; ++m) {
    for (int g : {256, 512, 1024}) {
      auto k = m == 0 ? wr<0> : (m == 1 ? wr<1> : wr<2>);
      hipLaunchKernelGGL(k, dim3(g), dim3(512), 0, 0, d, bytes);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(512), 0, 0, d, bytes);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-36s grid %4d: %.1f us per 256 MB = %.2f TB/s\n", names[m], g, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
    }
  }
  return 0;
}

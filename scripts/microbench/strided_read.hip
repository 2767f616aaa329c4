// Does the lifter attention's read pattern cost HBM bandwidth?  A unit of seq_attention_mfma reads, for each of q, k, v, N = 17 rows of
// 1 KB that sit 6 KB apart (one token's [q | k | v] row is 3C floats; a unit takes one 1 KB column chunk of each).  This reads the same
// bytes (a) that way and (b) from a planar layout where a unit's 17 rows of an operand are 17 KB contiguous, and prints TB/s.
//   hipcc -O3 --offload-arch=gfx950 scripts/microbench/strided_read.hip -o /tmp/strided_read && /tmp/strided_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// units: (sequence s, chunk g in 0..5 = {q,k,v} x 2 column chunks).  row(s, t) = s * 17 + t.
template <bool PLANAR>
__global__ __launch_bounds__(256) void reader(const float* __restrict__ buf, float* __restrict__ sink, int nseq, long long M) {
  const int lane16 = threadIdx.x & 63;   // 64 lanes x 16 B = one 1 KB row per wave instruction
  const int wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long nunits = (long long)nseq * 6;
  for (long long u = blockIdx.x; u < nunits; u += gridDim.x) {
    const long long s = u / 6;
    const int g = (int)(u % 6);
    for (int t = wave; t < 17; t += 4) {
      const long long row = s * 17 + t;
      const float* p = PLANAR ? buf + ((long long)g * M + row) * 256 + lane16 * 4 : buf + row * 1536 + g * 256 + lane16 * 4;
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
      acc += v;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main() {
  const int nseq = 4096;                  // B = 256: 4096 frames of 17 tokens
  const long long M = (long long)nseq * 17;
  const size_t bytes = (size_t)M * 1536 * 4;
  float *buf, *sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 4);
  hipMemset(buf, 0, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int grid : {512, 1024, 2048}) {
    for (int planar = 0; planar < 2; ++planar) {
      float best = 1e9f;
      for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(a);
        if (planar) hipLaunchKernelGGL(reader<true>, dim3(grid), dim3(256), 0, 0, buf, sink, nseq, M);
        else hipLaunchKernelGGL(reader<false>, dim3(grid), dim3(256), 0, 0, buf, sink, nseq, M);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("grid %4d %-28s %7.1f us  %.2f TB/s\n", grid, planar ? "planar (17 KB contiguous)" : "interleaved (1 KB at 6 KB)", best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}

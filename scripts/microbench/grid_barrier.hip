// What a grid-wide barrier costs on MI355X (round 6: would the GRU recurrence as ONE persistent launch per layer beat 25 dependent launches?).
// 256 workgroups (one per CU, all co-resident), N barrier rounds each; per round a workgroup optionally WRITES a slice of a shared buffer before
// it arrives and READS the whole buffer after it leaves (the recurrence's h: every workgroup needs all of it).  Spins are BOUNDED: a barrier that
// cannot complete (fewer resident workgroups than the grid) sets an error word instead of hanging the GPU.
//   mode 0: one counter (agent-scope atomic add, every workgroup polls it)
//   mode 1: XCD-hierarchical: a counter per XCD (blockIdx & 7); the XCD's last arriver bumps the global counter everybody polls
//   mode 2: as 1, and everybody polls its OWN XCD's release word, written by that XCD's elected workgroup after it saw the global counter
//   hipcc -O3 --offload-arch=gfx950 scripts/microbench/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned long long wall() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;  // 100 MHz
}
#define LD(p) __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)

struct Sync { unsigned global, pad0[31]; unsigned xcd[8][32]; unsigned rel[8][32]; unsigned err; };

__global__ __launch_bounds__(256) void rounds(int mode, int n, int payload_floats, float* buf, Sync* s, unsigned long long* out, float* sink) {
  const int tid = threadIdx.x, G = gridDim.x, wg = blockIdx.x, x = wg & 7, per_xcd = G >> 3;
  float acc = 0.f;
  const unsigned long long w0 = wall();
  for (int it = 1; it <= n; ++it) {
    if (payload_floats > 0) {  // this workgroup's slice of the buffer (the h of its units)
      const int per = payload_floats / G;
      for (int i = tid; i < per; i += 256) buf[wg * per + i] = (float)it;
      __threadfence();
    }
    __syncthreads();
    if (tid == 0) {
      int spins = 0;
      if (mode == 0) {
        ADD(&s->global, 1u);
        while (LD(&s->global) < (unsigned)(it * G) && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
      } else {
        const unsigned old = ADD(&s->xcd[x][0], 1u);
        const bool elected = old == (unsigned)(it * per_xcd - 1);
        if (elected) ADD(&s->global, 1u);
        if (mode == 1 || elected) {
          while (LD(&s->global) < (unsigned)(it * 8) && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
          if (mode == 2) ST(&s->rel[x][0], (unsigned)it);
        } else {
          while (LD(&s->rel[x][0]) < (unsigned)it && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        }
      }
      if (spins >= (1 << 22)) s->err = 1u;
    }
    __syncthreads();
    if (payload_floats > 0) {  // everybody reads everything (coalesced, L2 / MALL resident)
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4* b4 = reinterpret_cast<const f4*>(buf);
      for (int i = tid; i < payload_floats / 4; i += 256) {
        const f4 v = __builtin_nontemporal_load(b4 + i);
        acc += v.x + v.y + v.z + v.w;
      }
    }
  }
  const unsigned long long w1 = wall();
  if (tid == 0) out[wg] = w1 - w0;
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int G = 256, N = 2000;
  Sync* s; unsigned long long* out; float *buf, *sink;
  hipMalloc(&s, sizeof(Sync)); hipMalloc(&out, G * 8); hipMalloc(&buf, 1 << 22); hipMalloc(&sink, 4);
  unsigned long long h[G];
  for (int payload : {0, 1 << 14, 1 << 16, 1 << 18}) {   // 0, 64 KB, 256 KB, 1 MB of h per round
    for (int mode = 0; mode < 3; ++mode) {
      hipMemset(s, 0, sizeof(Sync));
      hipLaunchKernelGGL(rounds, dim3(G), dim3(256), 0, 0, mode, 20, payload, buf, s, out, sink);   // warm
      hipMemset(s, 0, sizeof(Sync));
      hipDeviceSynchronize();
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(rounds, dim3(G), dim3(256), 0, 0, mode, N, payload, buf, s, out, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      Sync hs; hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost);
      hipMemcpy(h, out, G * 8, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; for (int i = 0; i < G; ++i) mx = h[i] > mx ? h[i] : mx;
      printf("payload %7d B  mode %d: %.2f us per round (event), %.2f us (slowest workgroup's own clock)%s\n", payload * 4, mode, ms * 1e3 / N, mx * 0.01 / N,
             hs.err ? "  ** BARRIER TIMED OUT **" : "");
    }
  }
  // the alternative: N empty dependent launches on one stream
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipMemset(s, 0, sizeof(Sync));
  hipEventRecord(a);
  for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(rounds, dim3(G), dim3(256), 0, 0, 0, 0, 0, buf, s, out, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("1000 empty dependent launches of 256 workgroups: %.2f us each\n", ms);
  return 0;
}

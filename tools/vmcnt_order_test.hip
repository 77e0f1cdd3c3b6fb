// Micro-test (debug): do LDS-DMA loads (global_load_lds) and ordinary global loads retire IN ORDER with respect to
// each other under one vmcnt counter on gfx950?  Each trial issues an old op from a cold (HBM) address and a
// newer op from a hot (L1/L2) address, waits with vmcnt(1) (= "all but the newest one done") and checks that the
// OLD op's data is really there.
//   case 0: old = LDS-DMA (cold), new = register load (hot)  -> LDS must hold the DMA data after vmcnt(1)
//   case 1: old = register load (cold), new = LDS-DMA (hot)  -> the register must hold the loaded data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__global__ void order_kernel(const uint32_t* cold, const uint32_t* hot, size_t cold_words, int iters, int mode,
                             unsigned long long* bad) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][64 * 4];
  const int lane = threadIdx.x & 63;
  unsigned long long nbad = 0;
  size_t pos = ((size_t)blockIdx.x * 7919 + 13) * 4096 % (cold_words - 4096);
  for (int it = 0; it < iters; ++it) {
    pos = (pos * 6364136223846793005ull + 1442695040888963407ull) % (cold_words - 4096);
    pos &= ~(size_t)3;
    const uint32_t* cp = cold + pos + lane * 4;        // 16 B per lane, cold line
    const uint32_t* hp = hot + lane * 4;               // always the same hot line
    lds[0][lane * 4] = 0xdeadbeefu;
    lds[1][lane * 4] = 0xdeadbeefu;
    __syncthreads();
    if (mode == 0) {
      uint32_t r0, r1, r2, r3;
      __builtin_amdgcn_global_load_lds((gptr_t)cp, (lptr_t)&lds[0][0], 16, 0, 0);   // old, cold -> LDS
      asm volatile("global_load_dwordx4 %0, %1, off\n" : "=v"(*(uint4*)&r0) : "v"(hp) : "memory");
      (void)r1; (void)r2; (void)r3;
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      const uint32_t got = lds[0][lane * 4];           // must be the DMA data
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (got != cp[0]) ++nbad;
    } else {
      uint4 r;
      asm volatile("global_load_dwordx4 %0, %1, off\n" : "=v"(r) : "v"(cp) : "memory");   // old, cold -> VGPR
      __builtin_amdgcn_global_load_lds((gptr_t)hp, (lptr_t)&lds[1][0], 16, 0, 0);          // new, hot -> LDS
      uint32_t got;
      asm volatile("s_waitcnt vmcnt(1)\n v_mov_b32 %0, %1" : "=v"(got) : "v"(r.x) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (got != cp[0]) ++nbad;
    }
    __syncthreads();
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  const size_t cold_words = (size_t)1 << 30;           // 4 GiB: far larger than L2 + MALL
  uint32_t *cold, *hot;
  unsigned long long* bad;
  if (hipMalloc(&cold, cold_words * 4) != hipSuccess || hipMalloc(&hot, 4096) != hipSuccess ||
      hipMalloc(&bad, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  std::vector<uint32_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;
  for (size_t off = 0; off < cold_words; off += h.size()) hipMemcpy(cold + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(hot, h.data(), 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(order_kernel, dim3(1024), dim3(64), 0, 0, cold, hot, cold_words, 2000, mode, bad);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    unsigned long long b = 0;
    hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %llu out-of-order observations in %d lane-trials\n", mode,
           mode == 0 ? "old LDS-DMA cold, new load hot" : "old load cold, new LDS-DMA hot", b, 1024 * 2000 * 64);
  }
  return 0;
}

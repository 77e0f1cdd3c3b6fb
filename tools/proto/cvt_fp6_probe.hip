// Probe of the gfx950 fp6 pack instructions: v_cvt_scalef32_pk32_fp6_f16 (32 halfs -> 32 x e2m3) and v_cvt_scalef32_2xpk16_fp6_f32
// (2 x 16 floats): is the result x / scale or x * scale, and in which order do the 32 codes sit in the 192 output bits?
// build: hipcc -O3 --offload-arch=gfx950 tools/proto/cvt_fp6_probe.hip -o tools/proto/cvt_fp6_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(const float* in, unsigned* out, float scale) {
  v16f a, b;
  v32h h;
  for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
  for (int i = 0; i < 32; ++i) h[i] = (_Float16)in[i];
  const v6u r32 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
  const v6u r16 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale);
  for (int i = 0; i < 6; ++i) { out[i] = r32[i]; out[6 + i] = r16[i]; }
}
static float e2m3_val(int code) {
  const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float h_in[32];
  for (int i = 0; i < 32; ++i) h_in[i] = e2m3_val((i * 5 + 3) & 63);       // 32 distinct-ish representable values
  float* din; unsigned* dout;
  CHECK(hipMalloc(&din, 128)); CHECK(hipMalloc(&dout, 48));
  CHECK(hipMemcpy(din, h_in, 128, hipMemcpyHostToDevice));
  for (float scale : {1.0f, 2.0f, 0.5f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, din, dout, scale);
    unsigned o[12];
    CHECK(hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost));
    for (int which = 0; which < 2; ++which) {
      unsigned char bytes[24];
      memcpy(bytes, o + 6 * which, 24);
      printf("%s scale %.1f:", which ? "pk32_fp6_f16   " : "2xpk16_fp6_f32 ", scale);
      int ok_div = 1, ok_mul = 1, ok_il = 1;
      for (int t = 0; t < 32; ++t) {
        int c = 0;
        for (int q = 0; q < 6; ++q) c |= ((bytes[(6 * t + q) >> 3] >> ((6 * t + q) & 7)) & 1) << q;
        const float v = e2m3_val(c);
        if (t < 6) printf(" %g", v);
        if (v != h_in[t] / scale) ok_div = 0;
        if (v != h_in[t] * scale) ok_mul = 0;
        const int src = (t & 1) * 16 + (t >> 1);                               // interleaved hypothesis: code 2i = a[i], 2i+1 = b[i]
        if (v != h_in[src] / scale) ok_il = 0;
      }
      printf(" ... in order & x/scale: %d   in order & x*scale: %d   interleaved & x/scale: %d   (inputs %g %g %g %g %g %g)\n", ok_div, ok_mul,
             ok_il, h_in[0], h_in[1], h_in[2], h_in[3], h_in[4], h_in[5]);
    }
  }
  return 0;
}

// Per-instruction issue rates relevant to Blake2s on gfx950 (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 4096
template <int OP> __global__ void k(uint32_t* out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = t * 2654435761u, b = t ^ 0x9e3779b9u, c = t + 77u, d = ~t;
  uint32_t e = a + 1, f = b + 2, g = c + 3, h = d + 4;
  for (int r = 0; r < REPS; ++r) {
    if (OP == 0) { a = __builtin_amdgcn_alignbit(a, a, 7); b = __builtin_amdgcn_alignbit(b, b, 7); c = __builtin_amdgcn_alignbit(c, c, 7); d = __builtin_amdgcn_alignbit(d, d, 7);
                   e = __builtin_amdgcn_alignbit(e, e, 7); f = __builtin_amdgcn_alignbit(f, f, 7); g = __builtin_amdgcn_alignbit(g, g, 7); h = __builtin_amdgcn_alignbit(h, h, 7); }
    if (OP == 1) { a ^= b; b ^= c; c ^= d; d ^= e; e ^= f; f ^= g; g ^= h; h ^= a; }
    if (OP == 2) { a += b; b += c; c += d; d += e; e += f; f += g; g += h; h += a; }
    if (OP == 3) { a = a + b + c; b = b + c + d; c = c + d + e; d = d + e + f; e = e + f + g; f = f + g + h; g = g + h + a; h = h + a + b; }
    if (OP == 4) { a = (a >> 16) | (a << 16); b = (b >> 16) | (b << 16); c = (c >> 16) | (c << 16); d = (d >> 16) | (d << 16);
                   e = (e >> 16) | (e << 16); f = (f >> 16) | (f << 16); g = (g >> 16) | (g << 16); h = (h >> 16) | (h << 16); a += r; b += r; c += r; d += r; e += r; f += r; g += r; h += r; }
    if (OP == 5) { a = a * b; b = b * c; c = c * d; d = d * e; e = e * f; f = f * g; g = g * h; h = h * a; }
    if (OP == 6) { a = __umulhi(a, b); b = __umulhi(b, c); c = __umulhi(c, d); d = __umulhi(d, e); e = __umulhi(e, f); f = __umulhi(f, g); g = __umulhi(g, h); h = __umulhi(h, a) | 1; }
  }
  out[t] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int OP> void run(const char* name, uint32_t* out, int per) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int blocks = 256 * 8;
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-12s %.3f ms  %.1f Tops/s\n", name, ms, (double)blocks * 256 * REPS * per / ms / 1e9);
}
int main() {
  uint32_t* out; hipMalloc(&out, 16u << 20);
  run<0>("alignbit", out, 8); run<1>("xor", out, 8); run<2>("add", out, 8); run<3>("add3", out, 8);
  run<4>("rot16+add", out, 16); run<5>("mul_lo", out, 8); run<6>("mul_hi", out, 8);
  return 0;
}

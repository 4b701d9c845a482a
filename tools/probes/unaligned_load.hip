// Probe: do 16-byte buffer / global loads at 2-byte-aligned addresses return the right bytes on gfx950 (unaligned access
// mode), and what do they cost?  hipcc --offload-arch=gfx950 -O3 tools/probes/unaligned_load.hip -o tools/probes/bin/unaligned_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__global__ void probe(const unsigned short* __restrict__ src, u32x4_t* __restrict__ out_buf, u32x4_t* __restrict__ out_glb,
                      int n_bytes, int shift) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, n_bytes, 0x00020000);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned off = (unsigned)(i * 16 + shift * 2);
  out_buf[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
  out_glb[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned char*>(src) + off);
}

template <int MODE>
__global__ void bw(const unsigned short* __restrict__ src, u32x4_t* __restrict__ out, size_t n_vec, int shift) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, (int)(n_vec * 16 + 64), 0x00020000);
  u32x4_t acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned off = (unsigned)(i * 16 + shift * 2);
    u32x4_t v = MODE == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0)
                          : *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned char*>(src) + off);
    acc ^= v;
  }
  if (acc.x == 0x12345678u) out[0] = acc;
}

__global__ void probe_store(const u32x4_t* __restrict__ src, unsigned short* __restrict__ dst, int shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(dst) + (size_t)i * 16 + shift * 2) = src[i];
}

int main() {
  const int N = 1 << 16;                 // vectors
  std::vector<unsigned short> h((size_t)N * 8 + 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(i * 2654435761u >> 7);
  unsigned short* d; u32x4_t *ob, *og;
  hipMalloc(&d, h.size() * 2); hipMalloc(&ob, N * 16); hipMalloc(&og, N * 16);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  std::vector<unsigned short> rb((size_t)N * 8), rg((size_t)N * 8);
  for (int shift = 0; shift < 8; ++shift) {
    hipLaunchKernelGGL(probe, dim3(N / 256), dim3(256), 0, 0, d, ob, og, (int)(h.size() * 2), shift);
    hipMemcpy(rb.data(), ob, N * 16, hipMemcpyDeviceToHost);
    hipMemcpy(rg.data(), og, N * 16, hipMemcpyDeviceToHost);
    size_t bad_b = 0, bad_g = 0;
    for (size_t i = 0; i < (size_t)N * 8; ++i) { bad_b += rb[i] != h[i + shift]; bad_g += rg[i] != h[i + shift]; }
    printf("shift %d elements (%2d bytes): buffer_load_b128 wrong %zu, global_load_b128 wrong %zu of %d\n", shift, shift * 2, bad_b, bad_g, N * 8);
  }
  {  // 16-byte stores at 2-byte-aligned addresses
    unsigned short* dd; hipMalloc(&dd, (size_t)N * 16 + 64);
    std::vector<unsigned short> back((size_t)N * 8 + 32);
    for (int shift = 0; shift < 8; ++shift) {
      hipMemset(dd, 0xEE, (size_t)N * 16 + 64);
      hipLaunchKernelGGL(probe_store, dim3(N / 256), dim3(256), 0, 0, (const u32x4_t*)d, dd, shift);
      hipMemcpy(back.data(), dd, (size_t)N * 16 + 64, hipMemcpyDeviceToHost);
      size_t bad = 0;
      for (size_t i = 0; i < (size_t)N * 8; ++i) bad += back[i + shift] != h[i];
      for (int i = 0; i < shift; ++i) bad += back[i] != 0xEEEE;
      bad += back[(size_t)N * 8 + shift] != 0xEEEE;
      printf("store shift %d elements: wrong %zu\n", shift, bad);
    }
  }
  // bandwidth: 256 MB
  const size_t NV = (size_t)16 << 20;
  unsigned short* big; hipMalloc(&big, NV * 16 + 128); hipMemset(big, 1, NV * 16 + 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
    for (int shift : {0, 1, 2, 3}) {
      float best = 1e9f;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(a, 0);
        if (mode == 0) hipLaunchKernelGGL(bw<0>, dim3(4096), dim3(256), 0, 0, big, ob, NV, shift);
        else hipLaunchKernelGGL(bw<1>, dim3(4096), dim3(256), 0, 0, big, ob, NV, shift);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      printf("%s shift %d: %.3f ms for 256 MB = %.0f GB/s\n", mode == 0 ? "buffer" : "global", shift, best, 268.4 / best);
    }
  return 0;
}

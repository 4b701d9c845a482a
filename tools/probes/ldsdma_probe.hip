// Probe of the gfx950 LDS-DMA path (`buffer_load_dwordx4 ... lds`) semantics the igemm relies on:
//   Q1  out-of-range lanes (voffset >= num_records): is ZERO written to LDS, or is the lane's slot left untouched?
//   Q2  destination = M0 base + lane * 16 (lane-linear), whatever the per-lane source offset is
//   Q3  lanes switched off by EXEC: slot untouched?
//   Q4  the builtin's immediate offset: applied to the source only, or to the LDS address as well?
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/bin/ldsdma_probe tools/probes/ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* s = (unsigned*)smem;
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) s[i] = 0xAAAAAAAAu;          // 4 KiB of sentinel
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)nbytes, 0x00020000);
  unsigned off = (unsigned)((63 - lane) * 16);                         // reversed source order: lane-linear dest check
  if (mode == 1 && (lane % 3 == 0)) off = 0xFFFFFFFFu;                 // OOB lanes
  if (mode == 4 && (lane % 3 == 0)) off = 0x80000000u;                 // OOB lanes, second marker
  if (mode == 2) {
    if (lane & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + 1024), 16, off, 0, 0, 0);
  } else if (mode == 3) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + 1024), 16, off, 0, 1024, 0);   // imm offset 1024
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + 1024), 16, off, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = s[i];
}

int main() {
  const int N = 2048;                                                  // dwords of source (8 KiB)
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x10000u + i;
  unsigned *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> r(1024);
  for (int mode = 0; mode <= 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, (unsigned)(mode == 3 ? N * 4 : 1024), o, mode);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    printf("mode %d (%s): err=%d\n", mode, mode == 0 ? "plain" : mode == 1 ? "OOB lanes 0xFFFFFFFF" : mode == 2 ? "EXEC-masked even lanes"
           : mode == 3 ? "imm offset 1024" : "OOB lanes 0x80000000", (int)e);
    // where did data land?
    int first = -1, last = -1, sentinel_in_dest = 0, zeros = 0;
    for (int i = 0; i < 1024; ++i) if (r[i] != 0xAAAAAAAAu) { if (first < 0) first = i; last = i; }
    printf("  touched dwords: first %d last %d (dest region = dwords 256..511)\n", first, last);
    for (int l = 0; l < 64; ++l) {
      unsigned v = r[(first < 0 ? 256 : (first / 256) * 256) + l * 4];
      if (v == 0xAAAAAAAAu) sentinel_in_dest++;
      else if (v == 0) zeros++;
    }
    printf("  per-lane slots: %d untouched (sentinel), %d zero\n", sentinel_in_dest, zeros);
    int base = first < 0 ? 256 : (first / 256) * 256;
    printf("  lane0..5 first dword: ");
    for (int l = 0; l < 6; ++l) printf("%08x ", r[base + l * 4]);
    printf(" (source dword index of lane l in plain mode = (63-l)*4 -> 0x%x for lane 0)\n", 0x10000 + 63 * 4);
  }
  return 0;
}

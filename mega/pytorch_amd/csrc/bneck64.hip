// Fused identity bottleneck of the backbone's full-resolution stage (layer1 blocks 1 and 2 of ResNet-50/101-C4:
// mega_core/modeling/backbone/resnet.py:324-344 with 256 -> 64 -> 64 (3x3) -> 256 channels, stride 1, identity residual,
// FrozenBN folded to scale / bias, layers/batch_norm.py:19-31):
//     y = relu( bn3(conv3( relu(bn2(conv2_3x3( relu(bn1(conv1(x))) ))) )) + x )
// in ONE persistent kernel.  Unfused, the block moves 2083 bytes per pixel through HBM in three launches (x read by
// conv1, t1 written / read with a halo, t2 written / read, x read again as the residual, y written) and every one of
// those launches is HBM-bound (4.3-4.7 TB/s measured); fused, t1 and t2 never leave the CU: x patch in (1.41 x 512 B with
// the halo), y out (512 B), the residual re-read hits L2 (it is the centre of the patch the same CU loaded microseconds
// earlier).
//
// One 512-thread block per CU walks 8 x 16-pixel output tiles:
//   conv1 on the tile's 10 x 18 halo patch (180 pixels, padded to 192 GEMM rows): the patch arrives in four 64-channel
//         K-chunks (global -> registers a whole tile ahead -> LDS, rows XOR-swizzled as in igemm8.hip), each with its
//         [64][64] slice of w1; waves 0-5 own 32 patch rows x 64 channels each (8 MFMAs per chunk).  Patch pixels outside
//         the image give t1 = 0 -- conv2's zero padding applies to t1, not to conv1(0) + bias;
//   conv2 exactly as conv64.hip: t1 in its swizzled patch layout, w2 [64][576] resident in LDS for the kernel's life,
//         taps ascending, channels ascending inside a tap; wave (mb, nb) owns 32 pixels x 32 channels (36 MFMAs);
//   conv3 on t2 [128 px][64] in LDS against w3 [256][64] (re-read from L2 every tile into the freed chunk area): wave w owns
//         output channels 32 w .. 32 w + 31 of all 128 pixels (16 MFMAs);
//   epilogue with igemm8.hip's arithmetic -- f32 acc * scale + bias, + residual (f32 add), ReLU on the rounded value -- in
//         registers (a lane's residual elements are 2-byte loads that hit L2), the finished bf16 tile staged through LDS
//         and written as whole 16-byte vectors.
// Same MFMA (v_mfma_f32_32x32x16_bf16), same ascending K order per output element, same epilogue arithmetic and the same
// bf16 roundings of t1 / t2 as the three launches it replaces: BIT-IDENTICAL results (tests/test_kernels_gpu.py::
// test_fused_bottleneck64_bit_equal_to_unfused).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BK_TY = 8, BK_TX = 16;                 // output tile
constexpr int BK_PY = BK_TY + 2, BK_PX = BK_TX + 2;  // halo patch 10 x 18
constexpr int BK_NP = BK_PY * BK_PX;                 // 180 patch pixels
constexpr int BK_M1 = 192;                           // conv1 GEMM rows (180 padded to 6 x 32)
constexpr int BK_NT = 512;
constexpr int BK_W2ROW = 1168;                       // LDS bytes per w2 row (576 bf16 + 16 pad)
constexpr int BK_WROW = 144;                         // LDS bytes per w1-chunk / w3 row (64 bf16 + 16 pad)
constexpr int BK_SBROW = 528;                        // bytes per pixel row of the output staging (256 bf16 + 16 pad)
// LDS map (bytes)
constexpr int BK_OFF_W2 = 0;
constexpr int BK_OFF_T1 = 64 * BK_W2ROW;                        // 74752
constexpr int BK_OFF_XB = BK_OFF_T1 + BK_NP * 128;              // 97792: x chunk [192][128 B] + w1 chunk, later w3 [256][144 B]
constexpr int BK_OFF_W1 = BK_OFF_XB + BK_M1 * 128;              // 122368
constexpr int BK_XB_BYTES = 256 * BK_WROW;                      // 36864 (>= 192 * 128 + 64 * 144 = 33792)
constexpr int BK_OFF_T2 = BK_OFF_XB + BK_XB_BYTES;              // 134656
constexpr int BK_LDS = BK_OFF_T2 + 128 * 128;                   // 151040
constexpr int BK_OFF_ST = BK_OFF_T1;                            // output staging [128][528 B] = 67584 B over T1 / XB / T2
static_assert(BK_OFF_ST + 128 * BK_SBROW <= BK_LDS, "staging must fit");
static_assert(BK_M1 * 128 + 64 * BK_WROW <= BK_XB_BYTES, "x chunk + w1 chunk must fit the XB region");

struct Bneck64Params {
  const bf16_t* x;      // [N][H][W][256]
  const bf16_t* w1;     // [64][256]
  const bf16_t* w2;     // [64][3][3][64]
  const bf16_t* w3;     // [256][64]
  const float *s1, *b1, *s2, *b2, *s3, *b3;
  bf16_t* out;          // [N][H][W][256]
  int N, H, W, tiles_y, tiles_x, ntiles;
};

// HT (every kernel / helper below): the 16-bit element type, bf16_t or f16_t (IEEE half, round 6).  The parameter structs keep
// raw 16-bit pointers (bf16_t = unsigned short): only the MFMA, the f32 <-> 16-bit conversions and the residual unpack differ.
template <typename HT>
__device__ __forceinline__ f32x16_t bk_mma(f32x16_t acc, const uint4& a, const uint4& b) {
  return Half16<HT>::mfma32(__builtin_bit_cast(u32x4_t, a), __builtin_bit_cast(u32x4_t, b), acc);
}

// ReLU on the ROUNDED value (the igemm / igemm8 fast epilogues' form): one packed max per pair, a negative gives +0
template <typename HT>
__device__ __forceinline__ unsigned bk_relu_pack(float a, float b) {
  const s16x2_t z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, Half16<HT>::pack2(a, b)), z));
}
template <typename HT>
__device__ __forceinline__ bf16_t bk_relu1(float a) {
  const bf16_t h = Half16<HT>::cvt(a);
  return (short)h < 0 ? (bf16_t)0 : h;
}

template <typename HT>
__global__ __launch_bounds__(BK_NT, 2) void bneck64_kernel(Bneck64Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const w2l = smem + BK_OFF_W2;
  unsigned char* const t1l = smem + BK_OFF_T1;
  unsigned char* const xbl = smem + BK_OFF_XB;
  unsigned char* const w1l = smem + BK_OFF_W1;
  unsigned char* const t2l = smem + BK_OFF_T2;
  unsigned char* const sbl = smem + BK_OFF_ST;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- w2 -> LDS once
  {
    const uint4* wg = reinterpret_cast<const uint4*>(p.w2);
    for (int e = tid; e < 64 * 72; e += BK_NT) {
      const int n = e / 72, c = e - n * 72;
      *reinterpret_cast<uint4*>(w2l + n * BK_W2ROW + c * 16) = wg[e];
    }
  }

  // x through a buffer resource: 32-bit offsets (the tensor is < 2 GiB, checked by the host), out-of-image patch pixels get
  // an out-of-range offset, which the hardware range check turns into zeros -- no 64-bit address arithmetic per load
  const unsigned xbytes = (unsigned)((size_t)p.N * p.H * p.W * 512);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)xbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    const int per_img = p.tiles_y * p.tiles_x;
    n = t / per_img;
    const int r = t - n * per_img;
    const int ty = r / p.tiles_x;
    y0 = ty * BK_TY;
    x0 = (r - ty * p.tiles_x) * BK_TX;
  };

  // ---- the next tile's x patch (12 x 16 B per thread: chunk kc = pieces 3 kc .. 3 kc + 2) and w1 (4 x 16 B: chunk kc)
  u32x4_t px[12], pw1[4];      // (plain vector types: arrays of HIP's uint4 class were left in scratch memory)
  auto prefetch = [&](int t) {
    int n, y0, x0;
    const bool live = t < p.ntiles;
    tile_origin(live ? t : 0, n, y0, x0);
    unsigned off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int v = tid + BK_NT * i;            // 0 .. 1535: row = v >> 3, 16-byte piece j = v & 7 of a 64-channel chunk
      const int row = v >> 3, j = v & 7;
      const int py = row / BK_PX, pxx = row - py * BK_PX;
      const int yy = y0 - 1 + py, xx = x0 - 1 + pxx;
      const bool ok = live && row < BK_NP && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      off[i] = ok ? (unsigned)(((n * p.H + yy) * p.W + xx) * 512 + j * 16) : OOB;
    }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        px[3 * kc + i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off[i], kc * 128, 0);
      }
      pw1[kc] = *reinterpret_cast<const u32x4_t*>(p.w1 + (tid >> 3) * 256 + kc * 64 + (tid & 7) * 8);
    }
  };

  // ---- per-lane constants (bases + swizzle terms; the per-step offsets are formed at the use: two VALU per fragment read)
  // conv1: wave w < 6 owns patch rows 32 w .. 32 w + 31; A fragment row = 32 w + l31
  const int r1 = 32 * wave + l31;
  const unsigned a1_base = (unsigned)(r1 * 128), a1_sw = (unsigned)((r1 >> 1) & 7);
  const unsigned b1_base = (unsigned)(l31 * BK_WROW + h * 16);
  // conv2: wave (mb, nb): pixels of tile rows 2 mb, 2 mb + 1; channels 32 nb ..
  const int mb2 = wave >> 1, nb2 = wave & 1;
  const int ay = 2 * mb2 + (l31 >> 4), ax = l31 & 15;
  const unsigned a2_base = (unsigned)((ay * BK_PX + ax) * 128);
  const int b2_off = (32 * nb2 + l31) * BK_W2ROW + h * 16;
  // conv3: wave w owns output channels 32 w .. 32 w + 31; A fragment row = 32 mb + l31 of t2 (the swizzle repeats every 16 rows)
  const unsigned a3_base = (unsigned)(l31 * 128), a3_sw = (unsigned)((l31 >> 1) & 7);
  const unsigned b3_base = (unsigned)((32 * wave + l31) * BK_WROW + h * 16);
  const float s1a = p.s1[l31], s1b = p.s1[32 + l31], b1a = p.b1[l31], b1b = p.b1[32 + l31];
  const float s2v = p.s2[32 * nb2 + l31], b2v = p.b2[32 * nb2 + l31];
  const float s3v = p.s3[32 * wave + l31], b3v = p.b3[32 * wave + l31];

  int t = blockIdx.x;
  prefetch(t);
  __syncthreads();                              // w2 is in LDS
  for (; t < p.ntiles; t += gridDim.x) {
    int n, y0, x0;
    tile_origin(t, n, y0, x0);
    // ================= conv1: four K-chunks of 64 channels
    f32x16_t c1a, c1b;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1a[r] = 0.f; c1b[r] = 0.f; }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int v = tid + BK_NT * i;
        const int row = v >> 3, j = v & 7;
        *reinterpret_cast<u32x4_t*>(xbl + row * 128 + ((j ^ ((row >> 1) & 7)) * 16)) = px[3 * kc + i];
      }
      *reinterpret_cast<u32x4_t*>(w1l + (tid >> 3) * BK_WROW + (tid & 7) * 16) = pw1[kc];
      __syncthreads();
      if (wave < 6) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 a = *reinterpret_cast<const uint4*>(xbl + a1_base + (((unsigned)(2 * ks + h) ^ a1_sw) << 4));
          const uint4 b0 = *reinterpret_cast<const uint4*>(w1l + b1_base + ks * 32);
          const uint4 b1 = *reinterpret_cast<const uint4*>(w1l + 32 * BK_WROW + b1_base + ks * 32);
          c1a = bk_mma<HT>(c1a, a, b0);
          c1b = bk_mma<HT>(c1b, a, b1);
        }
      }
      __syncthreads();                          // every wave is done with this chunk
    }
    // ---- w3 [256][64] (32 KB, L2-resident) for this tile's conv3: requested now, dropped into the (free) XB region below
    u32x4_t pw3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pw3[i] = reinterpret_cast<const u32x4_t*>(p.w3)[tid + BK_NT * i];   // vec id = row * 8 + j
    // ---- t1 = relu(bn1(.)) as bf16 into the conv2 patch layout; zero outside the image (conv2's padding)
    // (ho / lo: the lane's half / column, opaque to the compiler inside the tile loop -- otherwise it hoists the 16 rows' index
    //  arithmetic of this epilogue and of the staging below out of the persistent loop and keeps ~100 loop-invariant
    //  registers alive, which spill)
    int ho = h, lo = l31;
    asm volatile("" : "+v"(ho), "+v"(lo));
    if (wave < 6) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int R = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * ho;
        if (R < BK_NP) {
          const int py = R / BK_PX, pxx = R - py * BK_PX;
          const int yy = y0 - 1 + py, xx = x0 - 1 + pxx;
          const bool in = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
          // (converted unconditionally, then selected: the f16 conversion's register barrier -- common.h f16_src -- would
          //  otherwise keep it under a divergent branch per element instead of a v_cndmask)
          const bf16_t ra = bk_relu1<HT>(c1a[r] * s1a + b1a), rb = bk_relu1<HT>(c1b[r] * s1b + b1b);
          const bf16_t va = in ? ra : (bf16_t)0;
          const bf16_t vb = in ? rb : (bf16_t)0;
          unsigned char* base = t1l + R * 128;
          const int sw = (pxx >> 1) & 7;
          *reinterpret_cast<bf16_t*>(base + (((lo >> 3) ^ sw) * 16) + (lo & 7) * 2) = va;
          *reinterpret_cast<bf16_t*>(base + ((((32 + lo) >> 3) ^ sw) * 16) + (lo & 7) * 2) = vb;
        }
      }
    }
    // ---- w3 into the (free) XB region; the NEXT tile's patch starts travelling
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + BK_NT * i;            // row = v >> 3, piece j = v & 7
      *reinterpret_cast<u32x4_t*>(xbl + (v >> 3) * BK_WROW + (v & 7) * 16) = pw3[i];
    }
    prefetch(t + gridDim.x);
    __syncthreads();                            // t1 and w3 are visible
    // ================= conv2: 9 taps x 4 channel steps
    f32x16_t c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) c2[r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const unsigned sw2 = (unsigned)(((ax + kw) >> 1) & 7);
          const uint4 a = *reinterpret_cast<const uint4*>(t1l + a2_base + (kh * BK_PX + kw) * 128 +
                                                          (((unsigned)(2 * ks + h) ^ sw2) << 4));
          const uint4 b = *reinterpret_cast<const uint4*>(w2l + b2_off + (kh * 3 + kw) * 128 + ks * 32);
          c2 = bk_mma<HT>(c2, a, b);
        }
    // ---- t2 = relu(bn2(.)) as bf16 [128 px][64 ch], rows swizzled for conv3's A fragments
    {
      const int c = 32 * nb2 + lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int P = 32 * mb2 + (r & 3) + 8 * (r >> 2) + 4 * ho;
        *reinterpret_cast<bf16_t*>(t2l + P * 128 + (((c >> 3) ^ ((P >> 1) & 7)) * 16) + (c & 7) * 2) = bk_relu1<HT>(c2[r] * s2v + b2v);
      }
    }
    __syncthreads();                            // t2 is visible
    // ================= conv3 (64 -> 256) + bn3 + residual + ReLU, one 32-pixel block at a time: this wave's 32 output
    //                   channels; the residual elements of a lane (its channel, its 16 pixels) are 2-byte loads (L2 hits:
    //                   the centre of the patch this CU loaded a tile ago) requested before the block's MFMAs
    unsigned pk[4][8];                          // results as bf16 pairs: pk[mb][q] = (r = 2 q, r = 2 q + 1)
    {
      uint4 bf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const uint4*>(xbl + b3_base + ks * 32);
      const int c3ch = 32 * wave + lo;
      // byte offset of (tile pixel (0, 4 h), this lane's channel); the rest of a residual element's address is wave-uniform
      const unsigned res_base = (unsigned)(((n * p.H + y0) * p.W + x0 + 4 * ho) * 512 + c3ch * 2);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        __builtin_amdgcn_sched_barrier(0);      // (keeps the 64 residual loads of the four blocks from being hoisted together)
        float rs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {          // pixel 32 mb + (r&3) + 8 (r>>2) + 4 h of the tile = (row 2 mb + (r>>3), column ...)
          const int yy = y0 + 2 * mb + (r >> 3), xx = x0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * ho;
          const unsigned o = (yy < p.H && xx < p.W) ? res_base : OOB;
          const unsigned short q = __builtin_amdgcn_raw_buffer_load_b16(
              rs_x, o, ((2 * mb + (r >> 3)) * p.W + (r & 3) + 8 * ((r >> 2) & 1)) * 512, 0);
          rs[r] = Half16<HT>::one(q);
        }
        f32x16_t c3;
#pragma unroll
        for (int r = 0; r < 16; ++r) c3[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 a = *reinterpret_cast<const uint4*>(t2l + mb * 4096 + a3_base + (((unsigned)(2 * ks + h) ^ a3_sw) << 4));
          c3 = bk_mma<HT>(c3, a, bf[ks]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v0 = c3[2 * q] * s3v + b3v, v1 = c3[2 * q + 1] * s3v + b3v;      // (the f32 value igemm8 stages ...)
          v0 += rs[2 * q];                                                        // (... and the residual added to it)
          v1 += rs[2 * q + 1];
          pk[mb][q] = bk_relu_pack<HT>(v0, v1);
        }
      }
    }
    __syncthreads();                            // every wave is done with t2 / w3: the staging area may overwrite them
    // ================= the tile as bf16 [128 px][256 ch] through LDS (528-byte rows), out as whole 16-byte vectors
    {
      const int c3ch = 32 * wave + lo;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int P = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * ho;
          const unsigned v = pk[mb][r >> 1];
          *reinterpret_cast<bf16_t*>(sbl + P * BK_SBROW + c3ch * 2) = (bf16_t)((r & 1) ? (v >> 16) : (v & 0xffffu));
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + BK_NT * i;           // 0 .. 4095: pixel P = id >> 5, 8-channel vector cv = id & 31
      const int P = id >> 5, cv = id & 31;
      const int yy = y0 + (P >> 4), xx = x0 + (P & 15);
      const uint4 v = *reinterpret_cast<const uint4*>(sbl + P * BK_SBROW + cv * 16);
      const u32x4_t q = {v.x, v.y, v.z, v.w};
      const unsigned o = (yy < p.H && xx < p.W) ? (unsigned)(((n * p.H + yy) * p.W + xx) * 512 + cv * 16) : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(q, rs_o, o, 0, 0);
    }
    __syncthreads();                            // the staging area is free again (the next tile's conv1 chunks, t1)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The stage's FIRST block (layer1 block 0: 64 -> 64 -> 64 (3x3) -> 256 channels with the 1x1 downsample branch on the
// residual, backbone/resnet.py:266-276,:324-344):
//     y = relu( bn3(conv3(t2)) + bnd(convd(x)) ),   t2 = relu(bn2(conv2_3x3(relu(bn1(conv1 x)))))
// Unfused it is FOUR launches and 2.2 KB per pixel (x read twice, t1 / t2 written and read, the 256-channel identity
// written and read, y written); fused the 64-channel x patch is read once (1.41 x 128 B) and y written (512 B) -- the
// identity branch is computed in the kernel from the patch that is already in LDS.
// Same tile, wave roles and layouts as bneck64_kernel; differences: the patch is ONE 64-channel chunk that stays in LDS
// (the downsample GEMM reads its centre pixels as MFMA rows), t2 re-uses t1's area (one more barrier), and ONE 36 KB
// region holds w1, then wd, then w3 in turn.  The identity is rounded to bf16 before it is added (it is a tensor of its
// own in the unfused path), then the igemm8 epilogue arithmetic: bit-identical to the four launches.
constexpr int BD_OFF_W2 = 0;
constexpr int BD_OFF_T1 = 64 * BK_W2ROW;                        // 74752: t1 patch [180][128 B]; t2 [128][128 B] after conv2
constexpr int BD_OFF_XP = BD_OFF_T1 + BK_NP * 128;              // 97792: x patch [192][128 B], alive until the downsample GEMM
constexpr int BD_OFF_WR = BD_OFF_XP + BK_M1 * 128;              // 122368: w1 [64][144 B] -> wd [256][144 B] -> w3 [256][144 B]
constexpr int BD_LDS = BD_OFF_WR + 256 * BK_WROW;               // 159232
static_assert(BD_OFF_T1 + 128 * BK_SBROW <= BD_LDS, "output staging must fit over t1 / x patch / weight region");

struct Bneck64DsParams {
  const bf16_t* x;      // [N][H][W][64]
  const bf16_t* w1;     // [64][64]
  const bf16_t* w2;     // [64][3][3][64]
  const bf16_t* w3;     // [256][64]
  const bf16_t* wd;     // [256][64]
  const float *s1, *b1, *s2, *b2, *s3, *b3, *sd, *bd;
  bf16_t* out;          // [N][H][W][256]
  int N, H, W, tiles_y, tiles_x, ntiles;
};

template <typename HT>
__global__ __launch_bounds__(BK_NT, 2) void bneck64_ds_kernel(Bneck64DsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const w2l = smem + BD_OFF_W2;
  unsigned char* const t1l = smem + BD_OFF_T1;
  unsigned char* const t2l = smem + BD_OFF_T1;      // (after conv2)
  unsigned char* const xpl = smem + BD_OFF_XP;
  unsigned char* const wrl = smem + BD_OFF_WR;
  unsigned char* const sbl = smem + BD_OFF_T1;      // (after the last MFMA)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  {
    const uint4* wg = reinterpret_cast<const uint4*>(p.w2);
    for (int e = tid; e < 64 * 72; e += BK_NT) {
      const int n = e / 72, c = e - n * 72;
      *reinterpret_cast<uint4*>(w2l + n * BK_W2ROW + c * 16) = wg[e];
    }
  }
  const unsigned xbytes = (unsigned)((size_t)p.N * p.H * p.W * 128), obytes = (unsigned)((size_t)p.N * p.H * p.W * 512);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)obytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    const int per_img = p.tiles_y * p.tiles_x;
    n = t / per_img;
    const int r = t - n * per_img;
    const int ty = r / p.tiles_x;
    y0 = ty * BK_TY;
    x0 = (r - ty * p.tiles_x) * BK_TX;
  };
  // ---- the next tile's x patch: 192 rows x 8 pieces of 16 B = 3 per thread
  u32x4_t px[3];
  auto prefetch = [&](int t) {
    int n, y0, x0;
    const bool live = t < p.ntiles;
    tile_origin(live ? t : 0, n, y0, x0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int v = tid + BK_NT * i;
      const int row = v >> 3, j = v & 7;
      const int py = row / BK_PX, pxx = row - py * BK_PX;
      const int yy = y0 - 1 + py, xx = x0 - 1 + pxx;
      const bool ok = live && row < BK_NP && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      px[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(((n * p.H + yy) * p.W + xx) * 128 + j * 16) : OOB, 0, 0);
    }
  };
  // ---- per-lane constants (igemm layout: a lane owns one output channel and 16 pixels of its 32-pixel block)
  const int r1 = 32 * wave + l31;
  const unsigned a1_base = (unsigned)(r1 * 128), a1_sw = (unsigned)((r1 >> 1) & 7);
  const unsigned bw_base = (unsigned)(l31 * BK_WROW + h * 16);           // weight-region row l31 (+ 32 nb / + 32 wave rows)
  const int mb2 = wave >> 1, nb2 = wave & 1;
  const int ay = 2 * mb2 + (l31 >> 4), ax = l31 & 15;
  const unsigned a2_base = (unsigned)((ay * BK_PX + ax) * 128);
  const int b2_off = (32 * nb2 + l31) * BK_W2ROW + h * 16;
  const unsigned a3_base = (unsigned)(l31 * 128), a3_sw = (unsigned)((l31 >> 1) & 7);
  const float s1a = p.s1[l31], s1b = p.s1[32 + l31], b1a = p.b1[l31], b1b = p.b1[32 + l31];
  const float s2v = p.s2[32 * nb2 + l31], b2v = p.b2[32 * nb2 + l31];
  const float s3v = p.s3[32 * wave + l31], b3v = p.b3[32 * wave + l31];
  const float sdv = p.sd[32 * wave + l31], bdv = p.bd[32 * wave + l31];

  int t = blockIdx.x;
  prefetch(t);
  __syncthreads();                              // w2 is in LDS
  for (; t < p.ntiles; t += gridDim.x) {
    int n, y0, x0;
    tile_origin(t, n, y0, x0);
    int ho = h, lo = l31;                       // (opaque copies: keeps the epilogues' index arithmetic out of the loop-invariant set)
    asm volatile("" : "+v"(ho), "+v"(lo));
    // ================= the patch and w1 into LDS
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int v = tid + BK_NT * i;
      const int row = v >> 3, j = v & 7;
      *reinterpret_cast<u32x4_t*>(xpl + row * 128 + ((j ^ ((row >> 1) & 7)) * 16)) = px[i];
    }
    *reinterpret_cast<u32x4_t*>(wrl + (tid >> 3) * BK_WROW + (tid & 7) * 16) =
        *reinterpret_cast<const u32x4_t*>(p.w1 + (tid >> 3) * 64 + (tid & 7) * 8);
    u32x4_t pwd[4];                             // wd for this tile: requested now, stored after conv1 is done with w1
#pragma unroll
    for (int i = 0; i < 4; ++i) pwd[i] = reinterpret_cast<const u32x4_t*>(p.wd)[tid + BK_NT * i];
    __syncthreads();
    // ================= conv1 (K = 64: one chunk)
    f32x16_t c1a, c1b;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1a[r] = 0.f; c1b[r] = 0.f; }
    if (wave < 6) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(xpl + a1_base + (((unsigned)(2 * ks + h) ^ a1_sw) << 4));
        const uint4 b0 = *reinterpret_cast<const uint4*>(wrl + bw_base + ks * 32);
        const uint4 b1 = *reinterpret_cast<const uint4*>(wrl + 32 * BK_WROW + bw_base + ks * 32);
        c1a = bk_mma<HT>(c1a, a, b0);
        c1b = bk_mma<HT>(c1b, a, b1);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int R = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * ho;
        if (R < BK_NP) {
          const int py = R / BK_PX, pxx = R - py * BK_PX;
          const int yy = y0 - 1 + py, xx = x0 - 1 + pxx;
          const bool in = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
          // (converted unconditionally, then selected: the f16 conversion's register barrier -- common.h f16_src -- would
          //  otherwise keep it under a divergent branch per element instead of a v_cndmask)
          const bf16_t ra = bk_relu1<HT>(c1a[r] * s1a + b1a), rb = bk_relu1<HT>(c1b[r] * s1b + b1b);
          const bf16_t va = in ? ra : (bf16_t)0;
          const bf16_t vb = in ? rb : (bf16_t)0;
          unsigned char* base = t1l + R * 128;
          const int sw = (pxx >> 1) & 7;
          *reinterpret_cast<bf16_t*>(base + (((lo >> 3) ^ sw) * 16) + (lo & 7) * 2) = va;
          *reinterpret_cast<bf16_t*>(base + ((((32 + lo) >> 3) ^ sw) * 16) + (lo & 7) * 2) = vb;
        }
      }
    }
    __syncthreads();                            // t1 is visible; every wave is done with w1
    // ---- wd into the weight region; w3 and the next tile's patch start travelling
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + BK_NT * i;
      *reinterpret_cast<u32x4_t*>(wrl + (v >> 3) * BK_WROW + (v & 7) * 16) = pwd[i];
    }
    u32x4_t pw3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pw3[i] = reinterpret_cast<const u32x4_t*>(p.w3)[tid + BK_NT * i];
    prefetch(t + gridDim.x);
    // ================= conv2
    f32x16_t c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) c2[r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const unsigned sw2 = (unsigned)(((ax + kw) >> 1) & 7);
          const uint4 a = *reinterpret_cast<const uint4*>(t1l + a2_base + (kh * BK_PX + kw) * 128 +
                                                          (((unsigned)(2 * ks + h) ^ sw2) << 4));
          const uint4 b = *reinterpret_cast<const uint4*>(w2l + b2_off + (kh * 3 + kw) * 128 + ks * 32);
          c2 = bk_mma<HT>(c2, a, b);
        }
    __syncthreads();                            // every wave is done with t1 (t2 takes its place); wd is visible
    {
      const int c = 32 * nb2 + lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int P = 32 * mb2 + (r & 3) + 8 * (r >> 2) + 4 * ho;
        *reinterpret_cast<bf16_t*>(t2l + P * 128 + (((c >> 3) ^ ((P >> 1) & 7)) * 16) + (c & 7) * 2) = bk_relu1<HT>(c2[r] * s2v + b2v);
      }
    }
    // ================= the identity branch: bnd(convd(x)) on the patch's centre pixels, rounded to bf16 (two per register)
    unsigned pk[4][8];
    {
      uint4 bf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const uint4*>(wrl + 32 * wave * BK_WROW + bw_base + ks * 32);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int R = (2 * mb + (l31 >> 4) + 1) * BK_PX + (l31 & 15) + 1;      // patch row of tile pixel 32 mb + l31
        const unsigned xa = (unsigned)(R * 128), xs = (unsigned)((R >> 1) & 7);
        f32x16_t cd;
#pragma unroll
        for (int r = 0; r < 16; ++r) cd[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 a = *reinterpret_cast<const uint4*>(xpl + xa + (((unsigned)(2 * ks + h) ^ xs) << 4));
          cd = bk_mma<HT>(cd, a, bf[ks]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) pk[mb][q] = Half16<HT>::pack2(cd[2 * q] * sdv + bdv, cd[2 * q + 1] * sdv + bdv);
      }
    }
    __syncthreads();                            // t2 is visible; every wave is done with wd
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + BK_NT * i;
      *reinterpret_cast<u32x4_t*>(wrl + (v >> 3) * BK_WROW + (v & 7) * 16) = pw3[i];
    }
    __syncthreads();                            // w3 is visible
    // ================= conv3 + bn3 + identity + ReLU (igemm8's epilogue arithmetic), in place over the packed identity
    {
      uint4 bf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[ks] = *reinterpret_cast<const uint4*>(wrl + 32 * wave * BK_WROW + bw_base + ks * 32);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f32x16_t c3;
#pragma unroll
        for (int r = 0; r < 16; ++r) c3[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 a = *reinterpret_cast<const uint4*>(t2l + mb * 4096 + a3_base + (((unsigned)(2 * ks + h) ^ a3_sw) << 4));
          c3 = bk_mma<HT>(c3, a, bf[ks]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v0 = c3[2 * q] * s3v + b3v, v1 = c3[2 * q + 1] * s3v + b3v;
          v0 += Half16<HT>::lo(pk[mb][q]);
          v1 += Half16<HT>::hi(pk[mb][q]);
          pk[mb][q] = bk_relu_pack<HT>(v0, v1);
        }
      }
    }
    __syncthreads();                            // every wave is done with t2 / w3 / the patch: the staging area may overwrite them
    {
      const int c3ch = 32 * wave + lo;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int P = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * ho;
          const unsigned v = pk[mb][r >> 1];
          *reinterpret_cast<bf16_t*>(sbl + P * BK_SBROW + c3ch * 2) = (bf16_t)((r & 1) ? (v >> 16) : (v & 0xffffu));
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + BK_NT * i;
      const int P = id >> 5, cv = id & 31;
      const int yy = y0 + (P >> 4), xx = x0 + (P & 15);
      const uint4 v = *reinterpret_cast<const uint4*>(sbl + P * BK_SBROW + cv * 16);
      const u32x4_t q = {v.x, v.y, v.z, v.w};
      const unsigned o = (yy < p.H && xx < p.W) ? (unsigned)(((n * p.H + yy) * p.W + xx) * 512 + cv * 16) : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(q, rs_o, o, 0, 0);
    }
    __syncthreads();                            // the staging area is free again
  }
}

}  // namespace

// y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + x) for a 256 -> 64 -> 64 (3x3, pad 1) -> 256 identity bottleneck,
// NHWC bf16, FrozenBN as f32 scale / bias vectors.  Bit-identical to the three mega_conv2d_nhwc launches it replaces.
extern "C" int mega_bottleneck64_fwd_dt(const void* x, const void* w1, const float* s1, const float* b1, const void* w2,
                                        const float* s2, const float* b2, const void* w3, const float* s3, const float* b3,
                                        void* out, int N, int H, int W, int dtype, void* stream) {
  mega_clear_error();
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (!x || !w1 || !s1 || !b1 || !w2 || !s2 || !b2 || !w3 || !s3 || !b3 || !out || N <= 0 || H <= 0 || W <= 0) return MEGA_ERR_ARG;
  if ((size_t)N * H * W * 256 * 2 >= 0x7FF00000ull) return MEGA_ERR_ARG;          // 32-bit buffer offsets
  Bneck64Params p;
  p.x = (const bf16_t*)x; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.w3 = (const bf16_t*)w3;
  p.s1 = s1; p.b1 = b1; p.s2 = s2; p.b2 = b2; p.s3 = s3; p.b3 = b3;
  p.out = (bf16_t*)out;
  p.N = N; p.H = H; p.W = W;
  p.tiles_y = cdiv(H, BK_TY);
  p.tiles_x = cdiv(W, BK_TX);
  const long tiles = (long)N * p.tiles_y * p.tiles_x;
  if (tiles > 0x7FFFFFFF) return MEGA_ERR_ARG;
  p.ntiles = (int)tiles;
  int cus = 256;
  {
    static int cached[64] = {0};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached[dev] = n;
      else cached[dev] = 256;
    }
    cus = cached[dev];
  }
  const int grid = (int)(tiles < cus ? tiles : cus);
  if (dtype == MEGA_F16) {
    (void)hipFuncSetAttribute((const void*)bneck64_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS);
    hipLaunchKernelGGL(bneck64_kernel<f16_t>, dim3(grid), dim3(BK_NT), BK_LDS, (hipStream_t)stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)bneck64_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS);
    hipLaunchKernelGGL(bneck64_kernel<bf16_t>, dim3(grid), dim3(BK_NT), BK_LDS, (hipStream_t)stream, p);
  }
  return mega_check_launch();
}

extern "C" int mega_bottleneck64_fwd(const void* x, const void* w1, const float* s1, const float* b1, const void* w2,
                                     const float* s2, const float* b2, const void* w3, const float* s3, const float* b3,
                                     void* out, int N, int H, int W, void* stream) {
  return mega_bottleneck64_fwd_dt(x, w1, s1, b1, w2, s2, b2, w3, s3, b3, out, N, H, W, MEGA_BF16, stream);
}

// The stage's first block with the 1x1 downsample branch: x NHWC bf16 [N][H][W][64] -> out [N][H][W][256],
//   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + bnd(convd(x))).  Bit-identical to the four launches it replaces.
extern "C" int mega_bottleneck64_ds_fwd_dt(const void* x, const void* w1, const float* s1, const float* b1, const void* w2,
                                           const float* s2, const float* b2, const void* w3, const float* s3, const float* b3,
                                           const void* wd, const float* sd, const float* bd, void* out, int N, int H, int W,
                                           int dtype, void* stream) {
  mega_clear_error();
  if (dtype != MEGA_BF16 && dtype != MEGA_F16) return MEGA_ERR_ARG;
  if (!x || !w1 || !s1 || !b1 || !w2 || !s2 || !b2 || !w3 || !s3 || !b3 || !wd || !sd || !bd || !out || N <= 0 || H <= 0 || W <= 0)
    return MEGA_ERR_ARG;
  if ((size_t)N * H * W * 256 * 2 >= 0x7FF00000ull) return MEGA_ERR_ARG;          // 32-bit buffer offsets
  Bneck64DsParams p;
  p.x = (const bf16_t*)x; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.w3 = (const bf16_t*)w3; p.wd = (const bf16_t*)wd;
  p.s1 = s1; p.b1 = b1; p.s2 = s2; p.b2 = b2; p.s3 = s3; p.b3 = b3; p.sd = sd; p.bd = bd;
  p.out = (bf16_t*)out;
  p.N = N; p.H = H; p.W = W;
  p.tiles_y = cdiv(H, BK_TY);
  p.tiles_x = cdiv(W, BK_TX);
  const long tiles = (long)N * p.tiles_y * p.tiles_x;
  if (tiles > 0x7FFFFFFF) return MEGA_ERR_ARG;
  p.ntiles = (int)tiles;
  int cus = 256;
  {
    static int cached[64] = {0};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached[dev] = n;
      else cached[dev] = 256;
    }
    cus = cached[dev];
  }
  const int grid = (int)(tiles < cus ? tiles : cus);
  if (dtype == MEGA_F16) {
    (void)hipFuncSetAttribute((const void*)bneck64_ds_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, BD_LDS);
    hipLaunchKernelGGL(bneck64_ds_kernel<f16_t>, dim3(grid), dim3(BK_NT), BD_LDS, (hipStream_t)stream, p);
  } else {
    (void)hipFuncSetAttribute((const void*)bneck64_ds_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, BD_LDS);
    hipLaunchKernelGGL(bneck64_ds_kernel<bf16_t>, dim3(grid), dim3(BK_NT), BD_LDS, (hipStream_t)stream, p);
  }
  return mega_check_launch();
}

extern "C" int mega_bottleneck64_ds_fwd(const void* x, const void* w1, const float* s1, const float* b1, const void* w2,
                                        const float* s2, const float* b2, const void* w3, const float* s3, const float* b3,
                                        const void* wd, const float* sd, const float* bd, void* out, int N, int H, int W,
                                        void* stream) {
  return mega_bottleneck64_ds_fwd_dt(x, w1, s1, b1, w2, s2, b2, w3, s3, b3, wd, sd, bd, out, N, H, W, MEGA_BF16, stream);
}

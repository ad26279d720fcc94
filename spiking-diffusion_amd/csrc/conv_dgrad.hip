// Data gradient of a 3x3 / stride 1 / pad 1 convolution on 7x7 or 8x8 maps, on the bf16 / fp16 matrix cores with fp32 accuracy
// (the training step of the denoiser, R/snn_model/vq_diffusion.py:166-187 through autograd; the
// reference runs cuDNN's data-gradient kernels here):
//
//   gi[n][y][x][ci] = sum over (co, ky, kx) of  gy[n][y + 1 - ky][x + 1 - kx][co] * w[co][ky][kx][ci]
//
// a GEMM per image with M = 49 (or 64) positions, N = Cin, K = 9 Cout, both operands dense fp32.  Two forms of one kernel template:
//   * THREE bf16 terms per operand by truncation (x = x0 + x1 + x2 exactly: 8 + 8 + 8 significant bits), SIX of the nine cross
//     products on v_mfma_f32_32x32x16_bf16 -- (0,0) (0,1) (1,0) (0,2) (2,0) (1,1): every product is exact, what is dropped is
//     below 2^-24 of |g w| (spk_conv3x3_dgrad_bf16);
//   * TWO fp16 terms per operand after a power-of-two scale per image (gy) / per input channel (w) that puts the largest
//     magnitude into [2^14, 2^15): x 2^s = h + m to 2^-23, THREE products (h,h) (h,m) (m,h) on v_mfma_f32_32x32x16_f16, exact
//     descaling of the result (spk_conv3x3_dgrad_f16x2: half the matrix work, two thirds of the LDS footprint; the form the
//     training backward uses).
// Either way the fp32 accumulation rounds like an fp32 GEMM's -- measured 2-4e-7 relative L2 against fp64 for both, the
// framework's fp32 operator 2-6e-7 -- against ONE product per element pair on the fp32 matrix instruction at 1/16 of the rate.
// The leading product and the small ones go to SEPARATE accumulators (see the kernel).
//
// Mapping.  One workgroup = eight images (one per wave) x 32 NT input channels (NT = 1, or 2 in the two-term form where that
// still fills the CUs); a wave owns its image's two 32-row position tiles (four image rows each; 8x8: two full tiles; 7x7: 7 + 1 padding column per row, 49
// positions + 15 rows computed and dropped) x NT column tiles.  K runs in chunks of 16 output channels (one MFMA k step) x nine
// taps:
//   * the weights of a chunk are pre-packed ONCE per call (dgrad_pack_kernel / dgrad_pack_f16_kernel) as the term planes in the
//     B-fragment order [term][tap][k half][ci][8 co], one contiguous blob per (channel tile, chunk): staging is a flat copy;
//   * the chunk of each image's gy is split while it is deposited into a zero-bordered 9x9 grid per (term, k half): the A
//     fragment of tap (ky, kx) is ONE aligned 16-byte read at a constant offset from a per-lane base (no im2col);
//   * while a chunk is multiplied the next chunk's weight blob is copied global -> LDS (LDS-DMA, two weight buffers) and its gy
//     travels global -> registers (it is split on deposit); the next tap's fragments are read while the current tap's products run.
// No split K: every output is written once, by one wave, in a fixed order (deterministic).
// Measured (B = 32 x T = 16 = 512 images, us, two-term | three-term | framework): 256 -> 128 channels 84 | 96 | 172, 512 -> 256:
// 235-250 | 355-380 | 570-600, 256 -> 512: 237-253 | 350-374 | 570-610, 128 -> 320: 114 | 131-139 | 170-183.  Phase stamps
// (profiles/r3_ab_kernel_variants.txt (16)): the two waves of a SIMD run their tap loops one after the other at ~88 % of the
// MFMA rate; the deposit phase between two barriers is 16 % of a chunk.
#include "spk_common.h"
#include "den_common.h"
#include "../../include/spkdiff.h"

extern "C" long long spk_conv3x3_dgrad_ws_bytes(int Cout, int Cin);

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NIMG = 8, NTHR = 512, KC = 16;               // images per workgroup, threads, output channels per chunk
// maps of HH x HH positions, HH = 7 (the MNIST-shaped latents) or 8 (the CIFAR-shaped ones: 64 positions = two full tiles)
__host__ __device__ constexpr int n_cell(int hh) { return (hh + 2) * (hh + 2) + 1; }     // (HH + 2)^2 grid cells per (term, k half) + one spare
__host__ __device__ constexpr int g_img(int nterm, int hh) { return nterm * 2 * n_cell(hh) * 16; }   // bytes of one image's chunk in LDS: [term][k half][cell][8 co]
__host__ __device__ constexpr int w_blob(int nterm, int nt) { return nterm * 9 * 2 * 32 * nt * 16; }     // bytes of one packed weight chunk

struct DgArgs {
  const float* gy; const uint8_t* wp; float* gi;
  const unsigned* wmax;                                     // F16 form: bits of max |w| per input channel
  const unsigned* tag;                                      // prepacked weights: the channel-tile width they were packed for (or null)
  int N, Cout, Cin;
};

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// F16 form.  x 2^s = h + m + e: h = the fp16 nearest to x 2^s, m = the fp16 nearest to the (exact) remainder.  s puts the largest
// magnitude of the scaled set (an image's gy, an input channel's weights) into [2^14, 2^15).  While the remainder is a NORMAL
// fp16 -- scaled magnitudes down to about 2^-3, i.e. ~17 binades below the set's maximum -- |e| < 2^-22 |x 2^s|: fp32-like
// relative precision per element.  Below that the remainder falls into fp16's subnormals (lsb 2^-24) and the error becomes
// ABSOLUTE: |e| <= 2^-25 scaled = 2^-40 of the set's maximum (11 significant bits at 2^-29 of the maximum).  Accuracy is thus
// relative to the image's / channel's largest element, not per element as in an fp32 operator: an output whose whole 3x3
// neighbourhood lies more than ~17 binades below its image's maximum loses relative precision accordingly
// (tests/test_gpu_parity.py::test_data_gradient_forms_wide_dynamic_range_per_element: the three-term bf16 form, whose terms are
// exact truncations at any magnitude, is the per-element-accurate one).
__device__ __forceinline__ float scale_of(unsigned max_bits) {          // 2^s for a set whose largest magnitude has these bits
  int be = (int)((max_bits >> 23) & 0xFFu);
  if (be == 0) return 1.0f;                                              // all zero (or denormal): nothing to scale
  be = be < 32 ? 32 : be;
  return __uint_as_float((unsigned)(268 - be) << 23);                    // 2^(15 - (be - 126))
}
__device__ __forceinline__ float inv_scale_of(unsigned max_bits) {
  int be = (int)((max_bits >> 23) & 0xFFu);
  if (be == 0) return 1.0f;
  be = be < 32 ? 32 : be;
  return __uint_as_float((unsigned)(be - 14) << 23);                     // 2^-(141 - be)
}
__device__ __forceinline__ void split2h(float xs, _Float16& h, _Float16& m) {
  h = (_Float16)xs;
  m = (_Float16)(xs - (float)h);
}

// F16 form, first pre-pass: bits of max |w| per input channel (wmax zeroed by the caller; positive floats order like their bits)
__global__ __launch_bounds__(256) void dgrad_wmax_kernel(const float* __restrict__ w, unsigned* __restrict__ wmax, int rows, int Cin) {
  __shared__ unsigned part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ci = blockIdx.x * 64 + tx;
  unsigned m = 0;
  if (ci < Cin)
    for (int r = blockIdx.y * 4 + ty; r < rows; r += gridDim.y * 4) {
      const unsigned b = __float_as_uint(w[(long long)r * Cin + ci]) & 0x7FFFFFFFu;
      m = b > m ? b : m;
    }
  part[ty][tx] = m;
  __syncthreads();
  if (ty == 0 && ci < Cin) {
    const unsigned a = part[0][tx] > part[1][tx] ? part[0][tx] : part[1][tx], c = part[2][tx] > part[3][tx] ? part[2][tx] : part[3][tx];
    atomicMax(wmax + ci, a > c ? a : c);
  }
}

// F16 form, second pre-pass: w [Cout][9][Cin] fp32 -> blobs [ci tile][chunk][term 2][tap][k half][ci in tile][8 co] fp16 of w 2^s(ci)
__global__ void dgrad_pack_f16_kernel(const float* __restrict__ w, const unsigned* __restrict__ wmax, _Float16* __restrict__ wp,
                                      int Cout, int Cin, int CI) {
  const long long n = (long long)Cout * 9 * Cin;
  const int n_chunks = Cout / KC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 9), co = (int)(i / (9LL * Cin));
    _Float16 h, m;
    split2h(w[i] * scale_of(wmax[ci]), h, m);
    const int tile = ci / CI, cil = ci % CI, chunk = co / KC, col = co % KC, kh = col >> 3, e = col & 7;
    const long long blob = ((long long)tile * n_chunks + chunk) * (2LL * 9 * 2 * CI * 8);
    const long long o = blob + ((long long)(tap * 2 + kh) * CI + cil) * 8 + e;
    wp[o] = h;
    wp[o + 9LL * 2 * CI * 8] = m;
  }
}

// F16 form, BOTH pre-passes for the weights of up to eight layers in ONE launch (spk_conv3x3_dgrad_f16x2_pack_multi): a training
// iteration packed each layer's weights inside that layer's backward -- a fill, a maximum and a pack launch per layer, 15 small
// launches and ~75 us per iteration at the reference's batch.  Here a workgroup owns EIGHT input channels of one layer: it reads
// their [Cout * 9] x 8 slice once for the maxima (32-byte row segments; LDS atomics, no zeroed global buffer, no ordering
// between workgroups -> deterministic), then again (from L2) for the two fp16 terms.  Same arithmetic as the two kernels above:
// the packed bytes are identical (tests/test_gpu_parity.py::test_dgrad_prepacked_multi_identical).
constexpr int DG_MULTI_MAX = 8, DG_CB = 8;
struct DgPackMulti {
  const float* w[DG_MULTI_MAX]; _Float16* wp[DG_MULTI_MAX]; unsigned* wmax[DG_MULTI_MAX];
  int Cout[DG_MULTI_MAX], Cin[DG_MULTI_MAX], CI[DG_MULTI_MAX], first[DG_MULTI_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void dgrad_pack_f16_multi_kernel(DgPackMulti m) {
  __shared__ unsigned smax[DG_CB];
  int L = 0;
  while (L + 1 < m.n && (int)blockIdx.x >= m.first[L + 1]) ++L;
  const int Cout = m.Cout[L], Cin = m.Cin[L], CI = m.CI[L], ci0 = ((int)blockIdx.x - m.first[L]) * DG_CB;
  const float* __restrict__ w = m.w[L];
  const int tid = threadIdx.x, rows = Cout * 9;
  if (tid < DG_CB) smax[tid] = 0u;
  __syncthreads();
  {
    const int hf = tid & 1;
    unsigned mx[4] = {0u, 0u, 0u, 0u};
    constexpr int UN = 4;
    for (int r0 = tid >> 1; r0 < rows; r0 += 128 * UN) {
      v4f v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int r = r0 + 128 * u;
        v[u] = *reinterpret_cast<const v4f*>(w + (long long)(r < rows ? r : rows - 1) * Cin + ci0 + 4 * hf);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned b = __float_as_uint(v[u][e]) & 0x7FFFFFFFu;
          mx[e] = b > mx[e] ? b : mx[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicMax(&smax[4 * hf + e], mx[e]);
  }
  __syncthreads();
  if (tid < DG_CB) m.wmax[L][ci0 + tid] = smax[tid];
  if (ci0 == 0 && tid == 0) m.wmax[L][Cin] = (unsigned)CI;    // (what the data call checks its own tile width against)
  const int cil = tid & 7, ci = ci0 + cil;
  const float sc = scale_of(smax[cil]);
  const int n_chunks = Cout / KC, tile = ci / CI, cit = ci % CI;
  const long long term = 9LL * 2 * CI * 8;
  _Float16* __restrict__ wp = m.wp[L];
  // item = (8 output channels kk, tap, channel): eight strided reads (32-byte segments across the eight lanes of a row), one 16-byte
  // store per term
  for (int i = tid; i < (Cout / 8) * 72; i += 256) {
    const int tap = (i >> 3) % 9, kk = i / 72, chunk = kk >> 1, kh = kk & 1;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = w[((long long)(kk * 8 + e) * 9 + tap) * Cin + ci];
    v8h h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 a, b;
      split2h(x[e] * sc, a, b);
      h[e] = a; l[e] = b;
    }
    const long long o = ((long long)tile * n_chunks + chunk) * (2 * term) + ((long long)(tap * 2 + kh) * CI + cit) * 8;
    *reinterpret_cast<v8h*>(wp + o) = h;
    *reinterpret_cast<v8h*>(wp + o + term) = l;
  }
}

// x = x0 + x1 + x2 exactly, each a bf16 (the top 16 bits of an fp32): truncate, subtract (exact), truncate, subtract
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));             // at most 8 significant bits are left: its truncation is exact
}

// w [Cout][9][Cin] fp32 -> blobs [ci tile][chunk][term][tap][k half][ci in tile][8 co] bf16
__global__ void dgrad_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int Cin, int CI) {
  const long long n = (long long)Cout * 9 * Cin;
  const int n_chunks = Cout / KC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 9), co = (int)(i / (9LL * Cin));
    unsigned h, m, l;
    split3(w[i], h, m, l);
    const int tile = ci / CI, cil = ci % CI, chunk = co / KC, col = co % KC, kh = col >> 3, e = col & 7;
    const long long blob = ((long long)tile * n_chunks + chunk) * (3LL * 9 * 2 * CI * 8);
    const long long o = blob + ((long long)(tap * 2 + kh) * CI + cil) * 8 + e;
    const long long term = 9LL * 2 * CI * 8;
    wp[o] = (unsigned short)(h >> 16);
    wp[o + term] = (unsigned short)(m >> 16);
    wp[o + 2 * term] = (unsigned short)(l >> 16);
  }
}

template <int NT, bool F16, int HH>
__global__ __launch_bounds__(NTHR, 1) void dgrad3x3_kernel(DgArgs a) {
  constexpr int NTERM = F16 ? 2 : 3, NPROD = F16 ? 3 : 6;
  constexpr int HW7 = HH * HH, GW = HH + 2, NCELL = n_cell(HH);         // (HW7: positions per map, whatever HH)
  constexpr int CI = 32 * NT, WB = w_blob(NTERM, NT), G_IMG = g_img(NTERM, HH);
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* const sW = lds;                                  // [2 buffers][term][tap][k half][ci][8 co]
  uint8_t* const sG = lds + 2 * WB;                         // [image][term][k half][cell][8 co]
  const unsigned sW_addr = spk_lds_addr(sW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int n_ct = a.Cin / CI, n_chunks = a.Cout / KC;
  const int ct = blockIdx.x % n_ct, ig = blockIdx.x / n_ct;
  const int n0 = ig * NIMG, ci0 = ct * CI;

  // zero the gy grids once: the border cells stay zero for the whole launch, interiors are rewritten by every chunk
  for (int i = tid; i < NIMG * G_IMG / 16; i += NTHR) reinterpret_cast<uint4*>(sG)[i] = make_uint4(0, 0, 0, 0);
  float* const s_scale = reinterpret_cast<float*>(lds + 2 * WB + NIMG * G_IMG);            // F16 form: 2^s of the workgroup's images
  float my_inv = 1.0f;
  if constexpr (F16) {
    // this wave's image: largest magnitude of its whole output gradient (49 x Cout values, read once more here)
    const int nn = n0 + wave < a.N ? n0 + wave : a.N - 1;
    const v4f* gp = reinterpret_cast<const v4f*>(a.gy + (long long)nn * HW7 * a.Cout);
    const int n4 = HW7 * a.Cout / 4;
    unsigned mb = 0;
    // (eight loads in flight: one at a time this loop cost a memory round trip per 1 KB -- a quarter of the launch)
    constexpr int UN = 8;
    for (int i0 = lane; i0 < n4; i0 += 64 * UN) {
      v4f v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 64 * u;
        v[u] = gp[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned b = __float_as_uint(v[u][e]) & 0x7FFFFFFFu;
          mb = b > mb ? b : mb;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)mb, off);
      mb = o > mb ? o : mb;
    }
    if (lane == 0) s_scale[wave] = scale_of(mb);
    my_inv = inv_scale_of(mb);
  }
  __syncthreads();

  // staging shares.  gy chunk: 8 images x 49 positions x 4 quarters of 16 channels (item tid + 512 j: image, position, quarter);
  // weights: a flat copy of WB bytes in 16-byte vectors
  constexpr int NG = (NIMG * HW7 * 4 + NTHR - 1) / NTHR;    // 4
  constexpr int NPW = (WB / 1024 + 7) / 8;                  // 1 KiB copy pieces of the weight blob per wave
  static_assert(WB % 1024 == 0, "whole copy pieces");
  v4f rg[NG];                                               // (ext vectors: HIP's uint4 / float4 structs kept these arrays in scratch)
  int g_src[NG], g_dst[NG];                                 // element offset in gy (without the chunk's channel offset) / LDS byte offset, -1: none
  float g_scl[NG];                                          // F16 form: 2^s of the item's image
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int idx = tid + NTHR * j;
    const int img = idx / (HW7 * 4), rem = idx % (HW7 * 4), pos = rem >> 2, q = rem & 3;
    const bool ok = idx < NIMG * HW7 * 4 && n0 + img < a.N;
    const int nn = n0 + img < a.N ? n0 + img : a.N - 1;
    g_src[j] = (nn * HW7 + pos) * a.Cout + 4 * q;
    const int cell = (pos / HH + 1) * GW + (pos % HH + 1);
    g_dst[j] = ok ? img * G_IMG + ((q >> 1) * NCELL + cell) * 16 + (q & 1) * 8 : -1;
    g_scl[j] = F16 ? s_scale[idx < NIMG * HW7 * 4 ? img : 0] : 1.0f;
  }
  const uint8_t* const wsrc = a.wp + ((long long)ct * n_chunks) * WB;
  auto fetch = [&](int chunk) {
    const float* g = a.gy + chunk * KC;
#pragma unroll
    for (int j = 0; j < NG; ++j) rg[j] = *reinterpret_cast<const v4f*>(g + g_src[j]);
    // the weight blob goes straight to the OTHER weight buffer (LDS-DMA, no registers): it was last read two chunks ago
    const uint8_t* wb = wsrc + (long long)chunk * WB;
    const unsigned dst = sW_addr + (unsigned)(chunk & 1) * WB;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      int pc = wave_s + 8 * j;
      pc = pc < WB / 1024 ? pc : WB / 1024 - 1;             // (the last round repeats a piece: same bytes to the same place)
      spk_dma16s(wb + pc * 1024, lane16, dst + pc * 1024);
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (F16) {
        if (g_dst[j] >= 0) {
          _Float16 h[4], m[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split2h(rg[j][e] * g_scl[j], h[e], m[e]);
          typedef _Float16 v4h __attribute__((ext_vector_type(4)));
          uint8_t* d = sG + g_dst[j];
          *reinterpret_cast<v4h*>(d) = (v4h){h[0], h[1], h[2], h[3]};
          *reinterpret_cast<v4h*>(d + 2 * NCELL * 16) = (v4h){m[0], m[1], m[2], m[3]};
        }
      } else if (g_dst[j] >= 0) {
        unsigned h[4], m[4], l[4];
        split3(rg[j][0], h[0], m[0], l[0]); split3(rg[j][1], h[1], m[1], l[1]);
        split3(rg[j][2], h[2], m[2], l[2]); split3(rg[j][3], h[3], m[3], l[3]);
        uint8_t* d = sG + g_dst[j];
        *reinterpret_cast<uint2*>(d) = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
        *reinterpret_cast<uint2*>(d + 2 * NCELL * 16) = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
        *reinterpret_cast<uint2*>(d + 4 * NCELL * 16) = make_uint2((l[0] >> 16) | (l[1] & 0xFFFF0000u), (l[2] >> 16) | (l[3] & 0xFFFF0000u));
      }
    }
  };

  // fragment bases: A row r of tile mt = image row 4 mt + r / 8, column r % 8 (column 7 and image row 7 are padding: computed,
  // never stored), k half = lane / 32; B column = input channel 32 nt + lane % 32.  A 16-lane group = two image rows = grid
  // cells c .. c + 6 and c + 9 .. c + 15: their 16-byte reads fall into distinct banks, and the two padding lanes take the
  // two residues left (c + 7, c + 8) -- consecutive positions (32 mt + r, seven per image row) put two lanes of every group
  // on occupied banks (SQ_LDS_BANK_CONFLICT 32 % of the LDS-active cycles)
  const int row = lane & 31, half = lane >> 5;
  const uint8_t* a_base[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int y = 4 * mt + (row >> 3);
    const int x = row & 7;
    y = y < HH ? y : HH - 1;
    int cell = y * GW + x;                                   // (7x7, x = 7: the padding lane reads the row's border cell ...)
    if (HH == 7 && x == 7 && (row & 8)) cell -= 8;           // (... that of the group's FIRST row + 1 for the second row: residue c + 8)
    a_base[mt] = sG + wave * G_IMG + (half * NCELL + cell) * 16;
  }
  const uint8_t* const b_base0 = sW + (half * CI + row) * 16;

  // TWO accumulators per output tile: the leading product (0,0) and the five small ones.  In one accumulator every small
  // product re-rounds the whole sum (six roundings of size eps |sum| per k step instead of one: 6e-7 relative L2 against fp64
  // where the framework's fp32 operator has 2e-7); the small products are below 2^-7 of the leading one, so the rounding of
  // their own sum is negligible and the leading chain is as long as an fp32 GEMM's.
  v16f acc[2][NT], acs[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[mt][nt][r] = 0.f; acs[mt][nt][r] = 0.f; }

  fetch(0);
  spk_dma_wait_all();
  deposit();
  __syncthreads();
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    if (chunk + 1 < n_chunks) fetch(chunk + 1);
    const uint8_t* const b_base = b_base0 + (chunk & 1) * WB;
    // nine taps: the fragments of tap t + 1 are read while the products of tap t run
    v4i af[2][2][NTERM], bf[2][NT][NTERM];
    auto load_tap = [&](int tap, int slot) {
      // gy position of output (y, x) under tap (ky, kx): (y + 1 - ky, x + 1 - kx) -> grid cell (y + 2 - ky) * 9 + (x + 2 - kx)
      const int toff = ((2 - tap / 3) * GW + (2 - tap % 3)) * 16;
#pragma unroll
      for (int t = 0; t < NTERM; ++t) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[slot][mt][t] = *reinterpret_cast<const v4i*>(a_base[mt] + t * 2 * NCELL * 16 + toff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bf[slot][nt][t] = *reinterpret_cast<const v4i*>(b_base + ((t * 9 + tap) * 2 * CI + nt * 32) * 16);
      }
    };
    load_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = tap & 1;
      if (tap + 1 < 9) load_tap(tap + 1, s ^ 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);                    // all of the next tap's reads are in flight before this tap's MFMAs
      // the products (term of gy, term of w): (0,0) (0,1) (1,0) [(0,2) (2,0) (1,1): three-term form]
      constexpr int TG[6] = {0, 0, 1, 0, 2, 1}, TW[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
      for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            v16f& d = pr == 0 ? acc[mt][nt] : acs[mt][nt];
            if constexpr (F16)
              d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, af[s][mt][TG[pr]]),
                                                         __builtin_bit_cast(v8h, bf[s][nt][TW[pr]]), d, 0, 0, 0);
            else
              d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, af[s][mt][TG[pr]]),
                                                          __builtin_bit_cast(v8bf, bf[s][nt][TW[pr]]), d, 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                        // everyone is done reading this chunk's gy
    spk_dma_wait_all();                                     // this wave's loads and weight pieces of the next chunk have landed
    if (chunk + 1 < n_chunks) deposit();
    __syncthreads();
  }

  const int n = n0 + wave;
  // weights packed by the multi-layer call for ANOTHER tile width (another N): the result would be a silent permutation --
  // NaN instead (the host cannot see the tag without a synchronisation)
  const bool bad_pack = a.tag && a.tag[0] != (unsigned)CI;
  if (n < a.N) {
    float* out = a.gi + (long long)n * HW7 * a.Cin + ci0 + row;
    float cinv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) cinv[nt] = F16 ? inv_scale_of(a.wmax[ci0 + nt * 32 + row]) : 1.0f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * half, y = 4 * mt + (rr >> 3), x = rr & 7;
        const int p = y * HH + x;
        if (x < HH && y < HH) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float v = acc[mt][nt][r] + acs[mt][nt][r];
            out[(long long)p * a.Cin + nt * 32] = bad_pack ? __uint_as_float(0x7FC00000u) : F16 ? (v * my_inv) * cinv[nt] : v;   // (powers of two: exact)
          }
        }
      }
  }
}

// column tiles per wave.  Three-term form: one -- with two accumulators per output tile a second column tile needs 128
// accumulator registers and leaves no room for double-buffered fragments (two tiles in ONE accumulator each: 434 us for the
// 256 -> 512 layer at B = 32 with the reads of a tap exposed).  Two-term form: the count whose (rounds of workgroups over the
// CUs) x (work per workgroup) is smaller, two on a tie -- measured at B = 32 (one | two tiles, us): 128 input channels 94 | 118,
// 256: 305 | 253, 512: 292 | 246, 320: 122 | 125.
int dgrad_nt(bool f16, int N, int Cin) {
  if (!f16 || (Cin % 64)) return 1;
  const long long cus = spk_cu_count(), groups = (N + NIMG - 1) / NIMG;
  const long long r2 = (groups * (Cin / 64) + cus - 1) / cus * 2, r1 = (groups * (Cin / 32) + cus - 1) / cus;
  return r2 <= r1 ? 2 : 1;
}

template <bool F16>
int dgrad_launch(const float* gy_cl, const float* w_cl, uint8_t* ws, long long ws_bytes, float* gi_out, int N, int H, int W,
                 int Cout, int Cin, hipStream_t stream, bool prepacked = false) {
  if (!gy_cl || (!w_cl && !prepacked) || !ws || !gi_out || N <= 0) return SPK_ERR_ARG;
  if (H != W || (H != 7 && H != 8) || Cout <= 0 || Cin <= 0 || (Cout % KC) || (Cin % 32)) return SPK_ERR_UNSUPPORTED;
  if ((long long)N * H * W * Cout >= (1LL << 31)) return SPK_ERR_UNSUPPORTED;    // (32-bit element offsets in the staging table)
  if (ws_bytes < spk_conv3x3_dgrad_ws_bytes(Cout, Cin)) return SPK_ERR_ARG;
  const int nt = dgrad_nt(F16, N, Cin), CI = 32 * nt;
  const long long n = (long long)Cout * 9 * Cin;
  const unsigned pack_grid = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  DgArgs a;
  a.gy = gy_cl; a.wp = ws; a.gi = gi_out; a.N = N; a.Cout = Cout; a.Cin = Cin;
  a.wmax = reinterpret_cast<const unsigned*>(ws + n * 6);
  a.tag = prepacked ? a.wmax + Cin : nullptr;
  if (prepacked) {
    // (the workspace already holds this layer's packed terms and maxima: spk_conv3x3_dgrad_f16x2_pack_multi, same N)
  } else if (F16) {
    unsigned* wmax = reinterpret_cast<unsigned*>(ws + n * 6);
    { const hipError_t e = hipMemsetAsync(wmax, 0, (size_t)Cin * 4, stream); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(dgrad_wmax_kernel, dim3((Cin + 63) / 64, 128), dim3(256), 0, stream, w_cl, wmax, Cout * 9, Cin);
    SPK_LAUNCH_CHECK();
    hipLaunchKernelGGL(dgrad_pack_f16_kernel, dim3(pack_grid), dim3(256), 0, stream, w_cl, wmax, reinterpret_cast<_Float16*>(ws),
                       Cout, Cin, CI);
  } else {
    hipLaunchKernelGGL(dgrad_pack_kernel, dim3(pack_grid), dim3(256), 0, stream, w_cl, reinterpret_cast<unsigned short*>(ws), Cout,
                       Cin, CI);
  }
  if (!prepacked) SPK_LAUNCH_CHECK();
  const int grid = ((N + NIMG - 1) / NIMG) * (Cin / CI);
  const size_t lds = 2 * (size_t)w_blob(F16 ? 2 : 3, nt) + (size_t)NIMG * g_img(F16 ? 2 : 3, H) + 64;
  if ((long long)lds > spk_lds_limit()) return SPK_ERR_UNSUPPORTED;     // (only the workspace has been written so far)
  if (H == 7) {
    if (F16 && nt == 2) hipLaunchKernelGGL((dgrad3x3_kernel<2, true, 7>), dim3(grid), dim3(NTHR), lds, stream, a);
    else if (F16) hipLaunchKernelGGL((dgrad3x3_kernel<1, true, 7>), dim3(grid), dim3(NTHR), lds, stream, a);
    else hipLaunchKernelGGL((dgrad3x3_kernel<1, false, 7>), dim3(grid), dim3(NTHR), lds, stream, a);
  } else {
    if (F16 && nt == 2) hipLaunchKernelGGL((dgrad3x3_kernel<2, true, 8>), dim3(grid), dim3(NTHR), lds, stream, a);
    else if (F16) hipLaunchKernelGGL((dgrad3x3_kernel<1, true, 8>), dim3(grid), dim3(NTHR), lds, stream, a);
    else hipLaunchKernelGGL((dgrad3x3_kernel<1, false, 8>), dim3(grid), dim3(NTHR), lds, stream, a);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

}  // namespace

extern "C" long long spk_conv3x3_dgrad_ws_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || (Cout % KC) || (Cin % 32)) return -1;
  return (long long)Cout * 9 * Cin * 6 + (long long)Cin * 4 + 16;              // packed weight terms + per-channel maxima
}

extern "C" int spk_conv3x3_dgrad_bf16(const float* gy_cl, const float* w_cl, uint8_t* ws, long long ws_bytes, float* gi_out,
                                      int N, int H, int W, int Cout, int Cin, hipStream_t stream) {
  return dgrad_launch<false>(gy_cl, w_cl, ws, ws_bytes, gi_out, N, H, W, Cout, Cin, stream);
}

extern "C" int spk_conv3x3_dgrad_f16x2(const float* gy_cl, const float* w_cl, uint8_t* ws, long long ws_bytes, float* gi_out,
                                       int N, int H, int W, int Cout, int Cin, hipStream_t stream) {
  return dgrad_launch<true>(gy_cl, w_cl, ws, ws_bytes, gi_out, N, H, W, Cout, Cin, stream);
}

// The weight half of spk_conv3x3_dgrad_f16x2 for up to eight layers in ONE launch, and the data half on its own.  The packed
// layout depends on the column tiles per wave, which depend on N: pack with the N the data call will be made with (the workspace
// records the tile width it was packed for; a data call that needs another one writes NaN, not a permuted result).  Host arrays of n entries; ws[i] of
// spk_conv3x3_dgrad_ws_bytes(Cout[i], Cin[i]) bytes.  Cin % 32 == 0 and Cout % 16 == 0 as for the one-layer call.
extern "C" int spk_conv3x3_dgrad_f16x2_pack_multi(const float* const* w_cl, uint8_t* const* ws, const long long* ws_bytes,
                                                  const int* N, const int* Cout, const int* Cin, int n, hipStream_t stream) {
  if (!w_cl || !ws || !ws_bytes || !N || !Cout || !Cin || n <= 0) return SPK_ERR_ARG;
  if (n > DG_MULTI_MAX) return SPK_ERR_UNSUPPORTED;
  DgPackMulti m;
  m.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!w_cl[i] || !ws[i] || N[i] <= 0) return SPK_ERR_ARG;
    if (Cout[i] <= 0 || Cin[i] <= 0 || (Cout[i] % KC) || (Cin[i] % 32)) return SPK_ERR_UNSUPPORTED;
    if (ws_bytes[i] < spk_conv3x3_dgrad_ws_bytes(Cout[i], Cin[i])) return SPK_ERR_ARG;
    const long long ne = (long long)Cout[i] * 9 * Cin[i];
    m.w[i] = w_cl[i]; m.wp[i] = reinterpret_cast<_Float16*>(ws[i]); m.wmax[i] = reinterpret_cast<unsigned*>(ws[i] + ne * 6);
    m.Cout[i] = Cout[i]; m.Cin[i] = Cin[i]; m.CI[i] = 32 * dgrad_nt(true, N[i], Cin[i]);
    m.first[i] = blocks;
    blocks += Cin[i] / DG_CB;
  }
  for (int i = n; i <= DG_MULTI_MAX; ++i) m.first[i] = blocks;
  hipLaunchKernelGGL(dgrad_pack_f16_multi_kernel, dim3(blocks), dim3(256), 0, stream, m);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_conv3x3_dgrad_f16x2_prepacked(const float* gy_cl, const uint8_t* ws, long long ws_bytes, float* gi_out, int N,
                                                 int H, int W, int Cout, int Cin, hipStream_t stream) {
  return dgrad_launch<true>(gy_cl, nullptr, const_cast<uint8_t*>(ws), ws_bytes, gi_out, N, H, W, Cout, Cin, stream, true);
}

// Weight gradient of a 3x3 / stride 1 / pad 1 convolution whose INPUT is a spike tensor, on the bf16 matrix cores
// (the training step of the denoiser, R/snn_model/vq_diffusion.py:166-187 through autograd; the
// reference runs cuDNN's fp32 weight-gradient kernels here):
//
//   gw[co][ky][kx][ci] = sum over (t, b, y, x) of  gy[t,b,co,y,x] * s[t,b,ci,y+ky-1,x+kx-1]          s in {0, 1}
//
// a GEMM with M = Cout, N = 9 Cin, K = T*B*49.  The spike operand is exact in bf16; the fp32 output gradient is split into
// THREE bf16 terms by truncation (x = hi + mid + lo exactly: 8 + 8 + 8 significant bits), so every product is exact and only
// the fp32 accumulation of v_mfma_f32_32x32x16_bf16 rounds -- an fp32 GEMM's accuracy at the bf16 rate (3 MFMAs per product
// tile: 178 GFLOP executed for the 256 -> 512 layer at B = 32).
//
// Mapping.  One workgroup (8 waves, two per SIMD) = 128 output channels x 64 input channels x all 9 taps: wave (w, c) owns
// output channels 32 w .. 32 w + 31 and input channels 32 c .. 32 c + 31 (144 accumulator registers), over a slice of the T*B
// images (split K; the partial sums of the slices are added in a fixed order by a second launch: deterministic).  K runs image
// by image: the MFMA's two 8-wide k groups are two IMAGE ROWS (7 positions + one zero), four k steps cover the 7 rows + one
// zero row (23 % of the products are padding), so that
//   * the A fragment of one term (gy, one output channel, one image row) is ONE aligned 16-byte read of an LDS image kept
//     [term][co][row][8] in bf16: the three-term split is done ONCE per element while the tile is deposited (not once per wave
//     that multiplies it: with the split in the multiply loop the launch was bound by vector-instruction issue, 228 us for the
//     256 -> 512 layer at B = 32);
//   * the B fragment of tap (ky, kx) (spikes, one input channel, the row shifted by the tap) is ONE aligned 16-byte read of
//     an LDS image kept [kx][ci][row + 1][8] in bf16: three copies pre-shifted by kx, zero rows above and below.
// Images are double buffered: the next image's tiles travel global -> registers while the current one is multiplied, and are
// split / deposited between the k steps.  Measured on the 256 -> 512 layer at B = 32 (512 images, 32 per workgroup; MI355X,
// rocprofv3): 161 us + 16.5 us for the reduction, the framework's operator 487 us; the MFMA pipe is busy 62 % of the launch
// (SQ_VALU_MFMA_BUSY_CYCLES).  With fetch and deposit switched off the image loop runs at ~90 % of the MFMA rate; ~30 us of
// the launch are outside the loop: the first image's latency and the 75 MB of partial sums (2048 waves x 144 accumulator
// registers: the price of filling 256 CUs with 16 output tiles) written here and read by the reduction.
#include "spk_common.h"
#include "den_common.h"
#include "../../include/spkdiff.h"

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int WG_CO = 128, WG_CI = 64, NTHR = 512;
// Maps of HH x HH positions, HH = 7 (MNIST-shaped latents) or 8 (CIFAR-shaped: the 64 positions fill the four k steps exactly,
// no zero column / zero row).  LDS images:
//   gy     [term][co][row][8] bf16.  7x7: 7 rows, pitch 112 B (the 16-lane phases of a 16-byte read hit 16 x 4 distinct banks).
//          8x8: 8 rows = 128 B, which alone would put every second channel on the same banks -- row r of channel co sits in
//          slot r ^ ((co >> 1) & 7) (16 consecutive channels x one row: 16 distinct (parity, slot) pairs); a 144-byte pitch
//          does not fit (2 x 3 x 128 x 144 + the spike images > 160 KB)
//   spikes [kx][ci][row + 1][8] bf16, zero rows above and below.  7x7: 9 rows, pitch 144 B.  8x8: 10 rows, pitch 160 B, which
//          repeats its bank pattern every 8 channels: the rows of channels with bit 3 set are rotated by one slot
__host__ __device__ constexpr int g_pitch(int hh) { return hh == 7 ? 56 : 64; }
__host__ __device__ constexpr int s_pitch(int hh) { return hh == 7 ? 72 : 80; }
__host__ __device__ constexpr int g_halfs(int hh) { return 3 * WG_CO * g_pitch(hh); }      // 7x7: 43 008 B, 8x8: 49 152 B
__host__ __device__ constexpr int s_halfs(int hh) { return 3 * WG_CI * s_pitch(hh); }      // 7x7: 27 648 B, 8x8: 30 720 B
__host__ __device__ constexpr int lds_bytes(int hh) { return 2 * g_halfs(hh) * 2 + 2 * s_halfs(hh) * 2 + 16; }   // + one 16-byte zero vector

struct WgArgs {
  const float* gy; const float* s; float* part;
  float* part_gb;                                   // [ksplit][Cout] column sums of gy (the bias gradient), or null
  int TB, Cout, Cin, ksplit;
};

template <int HH>
__global__ __launch_bounds__(NTHR, 1) void wgrad3x3_bf16_kernel(WgArgs a) {
  constexpr int HW7 = HH * HH, G_PITCH = g_pitch(HH), S_PITCH = s_pitch(HH), G_HALFS = g_halfs(HH), S_HALFS = s_halfs(HH),
                LDS_BYTES = lds_bytes(HH);                     // (HW7: positions per map, whatever HH)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  unsigned short* const sG = reinterpret_cast<unsigned short*>(lds);                       // [2][G_HALFS]
  unsigned short* const sS = sG + 2 * G_HALFS;                                            // [2][S_HALFS]
  const unsigned short* const sZ = sS + 2 * S_HALFS;                                      // 8 zeros
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ct = tid >> 8;
  const int n_ci = a.Cin / WG_CI;
  int bid = blockIdx.x;
  const int ks = bid % a.ksplit; bid /= a.ksplit;
  const int tn = bid % n_ci, tm = bid / n_ci;
  const int co0 = tm * WG_CO, ci0 = tn * WG_CI;
  const int per = (a.TB + a.ksplit - 1) / a.ksplit;
  const int i0 = ks * per, i1 = (i0 + per < a.TB) ? i0 + per : a.TB;

  // zero everything once: the border rows / columns of the spike image and the zero vector stay zero
  for (int i = tid; i < LDS_BYTES / 16; i += NTHR) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // this thread's share of an image: whole image ROWS, so that every LDS deposit is one aligned 16-byte write (element-wise
  // 2-byte deposits kept the LDS pipe busier than the matrix cores).  gy tile: item (channel e % 128, row e / 128) for
  // e = tid and tid + 512 (896 items); spike tile: item (channel tid % 64, row tid / 64) for tid < 448
  // (8x8: 1024 gy items = two per thread, 512 spike items = one per thread)
  float rg[2][HH], rs[HH];
  float gsum[2] = {0.f, 0.f};                       // bias gradient: this thread's rows of gy, summed over its images
  const int g_co = tid & 127, g_y0 = tid >> 7, s_ci = tid & 63, s_y = tid >> 6;
  const bool g_two = tid < HH * WG_CO - NTHR, s_on = tid < HH * WG_CI;                   // (both wave-uniform)
  // addresses = a wave-uniform pointer (image, column x, tile origin: scalar registers) + ONE 32-bit per-thread offset
  const unsigned g_off = (unsigned)(g_y0 * HH * a.Cout + g_co), s_off = (unsigned)(s_y * HH * a.Cin + s_ci);
  const int g_swz = HH == 8 ? (g_co >> 1) & 7 : 0;            // 8x8: slot of row r = r ^ g_swz (see above)
  const int s_rot = HH == 8 ? (s_ci >> 3) & 1 : 0;            // 8x8: slot of row r = (r + s_rot) mod 10
  auto fetch = [&](int img) {
    const float* g = a.gy + ((long long)img * HW7) * a.Cout + co0;
    const float* sp = a.s + ((long long)img * HW7) * a.Cin + ci0;
#pragma unroll
    for (int x = 0; x < HH; ++x) rg[0][x] = (g + x * a.Cout)[g_off];
    if (g_two) {
#pragma unroll
      for (int x = 0; x < HH; ++x) rg[1][x] = (g + (4 * HH + x) * a.Cout)[g_off];
    }
    if (s_on) {
#pragma unroll
      for (int x = 0; x < HH; ++x) rs[x] = (sp + x * a.Cin)[s_off];
    }
  };
  auto deposit = [&](int buf, int part) {        // part 0 / 1: the gy items, 2: the spike item, -1: all
    unsigned short* G = sG + buf * G_HALFS + g_co * G_PITCH;
    const int s_slot = HH == 8 ? (s_y + 1 + s_rot) % 10 : s_y + 1;
    unsigned short* S = sS + buf * S_HALFS + s_ci * S_PITCH + s_slot * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if ((part < 0 || part == j) && (j == 0 || g_two)) {
        // x = hi + mid + lo exactly, each term a bf16 (top 16 bits of an fp32): truncate, subtract (exact), truncate, subtract;
        // at most 8 significant bits are left for lo: its truncation is exact
        unsigned h[8], m[8], l[8];
        if constexpr (HH == 7) gsum[j] += ((rg[j][0] + rg[j][1]) + (rg[j][2] + rg[j][3])) + ((rg[j][4] + rg[j][5]) + rg[j][6]);
        else gsum[j] += ((rg[j][0] + rg[j][1]) + (rg[j][2] + rg[j][3])) + ((rg[j][4] + rg[j][5]) + (rg[j][6] + rg[j][HH - 1]));
#pragma unroll
        for (int x = 0; x < HH; ++x) {
          h[x] = __float_as_uint(rg[j][x]) & 0xFFFF0000u;
          const float p = rg[j][x] - __uint_as_float(h[x]);
          m[x] = __float_as_uint(p) & 0xFFFF0000u;
          l[x] = __float_as_uint(p - __uint_as_float(m[x]));
        }
        if constexpr (HH == 7) h[7] = m[7] = l[7] = 0u;                                    // column 7: the zero pad
        v4i vh, vm, vl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vh[q] = (int)((h[2 * q] >> 16) | h[2 * q + 1]);
          vm[q] = (int)((m[2 * q] >> 16) | m[2 * q + 1]);
          vl[q] = (int)((l[2 * q] >> 16) | (l[2 * q + 1] & 0xFFFF0000u));
        }
        unsigned short* d = G + ((g_y0 + 4 * j) ^ g_swz) * 8;
        *reinterpret_cast<v4i*>(d) = vh;
        *reinterpret_cast<v4i*>(d + WG_CO * G_PITCH) = vm;
        *reinterpret_cast<v4i*>(d + 2 * WG_CO * G_PITCH) = vl;
      }
    }
    if ((part < 0 || part == 2) && s_on) {
      // (the bf16 of the value by truncation: exact for spikes and for spike COUNTS up to 256 -- the time-collapsed last layer)
      unsigned v[8];
#pragma unroll
      for (int x = 0; x < HH; ++x) v[x] = __float_as_uint(rs[x]) >> 16;
      if constexpr (HH == 7) v[7] = 0u;
      // copy kx: slot xx of row y + 1 holds s(y, xx + kx - 1)
      const unsigned p12 = v[1] | (v[2] << 16), p34 = v[3] | (v[4] << 16), p56 = v[5] | (v[6] << 16);
      v4i c0, c1, c2;
      c0[0] = (int)(v[0] << 16); c0[1] = (int)p12; c0[2] = (int)p34; c0[3] = (int)p56;                 // 0 s0 | s1 s2 | s3 s4 | s5 s6
      c1[0] = (int)(v[0] | (v[1] << 16)); c1[1] = (int)(v[2] | (v[3] << 16)); c1[2] = (int)(v[4] | (v[5] << 16));
      c1[3] = (int)(v[6] | (v[7] << 16));                                                               // (7x7: s6 0)
      c2[0] = (int)p12; c2[1] = (int)p34; c2[2] = (int)p56; c2[3] = (int)v[7];                          // s1 s2 | s3 s4 | s5 s6 | s7 0 (7x7: 0 0)
      *reinterpret_cast<v4i*>(S) = c0;
      *reinterpret_cast<v4i*>(S + WG_CI * S_PITCH) = c1;
      *reinterpret_cast<v4i*>(S + 2 * WG_CI * S_PITCH) = c2;
    }
  };

  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (i0 < i1) { fetch(i0); deposit(0, -1); }
  __syncthreads();
  const int row = lane & 31, half = lane >> 5;
  for (int img = i0; img < i1; ++img) {
    const int buf = (img - i0) & 1;
    if (img + 1 < i1) fetch(img + 1);
    const unsigned short* G = sG + buf * G_HALFS + (wave * 32 + row) * G_PITCH;
    const unsigned short* S = sS + buf * S_HALFS + (ct * 32 + row) * S_PITCH;
    // 36 (k step, tap) products, the B fragment read three products ahead of its use and the A fragments of the next k step
    // half a k step ahead (read-then-use in source order left the LDS latency exposed before every tap)
    const int a_swz = HH == 8 ? ((wave * 32 + row) >> 1) & 7 : 0, b_rot = HH == 8 ? ((ct * 32 + row) >> 3) & 1 : 0;
    auto a_ptr = [&](int kstep, int term) -> const unsigned short* {
      if constexpr (HH == 8) return G + term * WG_CO * G_PITCH + ((2 * kstep + half) ^ a_swz) * 8;
      return (kstep == 3 && half) ? sZ : G + term * WG_CO * G_PITCH + (2 * kstep + half) * 8;     // row 7 = the zero row
    };
    auto b_read = [&](int i) -> v4i {
      const int kstep = i / 9, tap = i % 9, ky = tap / 3, kx = tap % 3;
      int r = 2 * kstep + half + ky;                        // LDS row of image row y + ky - 1
      if constexpr (HH == 8) {
        r += b_rot;                                         // rows 0 .. 9, rotated by one slot for channels with bit 3 set
        r = r >= 10 ? r - 10 : r;
      } else {
        r = r > 8 ? 8 : r;                                  // (only the zero row y = 7 can reach past the image: its A is zero)
      }
      return *reinterpret_cast<const v4i*>(S + kx * WG_CI * S_PITCH + r * 8);
    };
    constexpr int AHEAD = 3;
    v4i av[2][3], bq[AHEAD + 1];
#pragma unroll
    for (int t = 0; t < 3; ++t) av[0][t] = *reinterpret_cast<const v4i*>(a_ptr(0, t));
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) bq[i] = b_read(i);
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      const int kstep = i / 9, tap = i % 9;
      if (i + AHEAD < 36) bq[(i + AHEAD) % (AHEAD + 1)] = b_read(i + AHEAD);
      asm volatile("" ::: "memory");                        // (keeps every read where it is written: no merging with a later one)
      if (tap == 4 && kstep < 3) {
#pragma unroll
        for (int t = 0; t < 3; ++t) av[(kstep + 1) & 1][t] = *reinterpret_cast<const v4i*>(a_ptr(kstep + 1, t));
      }
      const v8bf b = __builtin_bit_cast(v8bf, bq[i % (AHEAD + 1)]);
#pragma unroll
      for (int t = 0; t < 3; ++t)
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, av[kstep & 1][t]), b, acc[tap], 0, 0, 0);
      // the next image's deposit rides between the k steps (its vector work next to the matrix cores' rather than in one
      // block after the loop, where both waves of a SIMD did it at the same time with the MFMA pipe idle)
      if (tap == 8 && kstep < 3 && img + 1 < i1) deposit(buf ^ 1, kstep);
      __builtin_amdgcn_sched_barrier(0);                    // source order = issue order, product by product
    }
    __syncthreads();
  }
  // the bias gradient of this slice (workgroups of the first input-channel tile): the seven row sums of a channel meet in LDS
  // in a fixed order
  if (a.part_gb && tn == 0) {
    float* sB = reinterpret_cast<float*>(lds);      // [HH rows][128 channels] (the operand buffers are done)
    sB[g_y0 * WG_CO + g_co] = gsum[0];
    if (g_two) sB[(g_y0 + 4) * WG_CO + g_co] = gsum[1];
    __syncthreads();
    if (tid < WG_CO) {
      float t = 0.f;
#pragma unroll
      for (int y = 0; y < HH; ++y) t += sB[y * WG_CO + tid];
      a.part_gb[(long long)ks * a.Cout + co0 + tid] = t;
    }
  }
  // partial sums of this slice: part[ks][co][tap][ci]
  float* out = a.part + (long long)ks * a.Cout * 9 * a.Cin;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[((long long)co * 9 + tap) * a.Cin + ci0 + ct * 32 + row] = acc[tap][r];
    }
}

// 64 consecutive outputs x 4 groups of slices per workgroup: every thread adds its quarter of the slices in four independent
// chains (a thread that walked all 128 slices of the small layer alone took 75 us: one memory latency per slice), the groups meet
// in LDS; every sum has ONE fixed order: deterministic
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, long long n,
                                                           int ksplit, const float* __restrict__ part_gb, float* __restrict__ gb,
                                                           int Cout) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int per = (ksplit + 3) / 4, k0 = grp * per, k1 = k0 + per < ksplit ? k0 + per : ksplit;
  for (long long base = (long long)blockIdx.x * 64; base < n + Cout; base += (long long)gridDim.x * 64) {
    // outputs [0, n): the weight gradient; [n, n + Cout): the bias gradient (its partial sums live behind the weight's)
    const long long i = base + lane;
    const bool is_w = i < n, live = i < n + (gb ? Cout : 0);
    const float* src = is_w ? part + i : part_gb + (i - n);
    const long long stride = is_w ? n : Cout;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (live) {
      int k = k0;
      for (; k + 3 < k1; k += 4) {
        a0 += src[(long long)k * stride]; a1 += src[(long long)(k + 1) * stride];
        a2 += src[(long long)(k + 2) * stride]; a3 += src[(long long)(k + 3) * stride];
      }
      for (; k < k1; ++k) a0 += src[(long long)k * stride];
    }
    sh[grp][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && live) {
      const float t = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
      if (is_w) gw[i] = t; else gb[i - n] = t;
    }
    __syncthreads();
  }
}

// slices of the image range: one workgroup per CU in ONE round (never a second, partly filled round)
int wgrad_ksplit(int TB, int Cout, int Cin) {
  const int tiles = (Cout / WG_CO) * (Cin / WG_CI);
  int ks = spk_cu_count() / tiles;
  // a single output tile (the 64 -> 128 layer): 256 slices would write and re-read 256 x 295 KB of partial sums for 11 GFLOP of
  // matrix work (measured 90 us against the library's 52); 128 slices halve that traffic at twice the (short) multiply time
  if (tiles == 1 && ks > 128) ks = 128;
  if (ks > TB) ks = TB;
  if (ks < 1) ks = 1;
  return ks;
}

}  // namespace

extern "C" long long spk_conv3x3_wgrad_ws_bytes(int TB, int Cout, int Cin) {
  if (TB <= 0 || Cout <= 0 || Cin <= 0 || (Cout % WG_CO) || (Cin % WG_CI)) return -1;
  return (long long)wgrad_ksplit(TB, Cout, Cin) * ((long long)Cout * 9 * Cin + Cout) * 4;      // weight + bias partial sums
}

extern "C" int spk_conv3x3_wgrad_bf16(const float* gy_cl, const float* spikes_cl, float* ws, long long ws_bytes,
                                      float* gw_out, float* gb_out_or_null, int TB, int H, int W, int Cout, int Cin,
                                      hipStream_t stream) {
  if (!gy_cl || !spikes_cl || !ws || !gw_out || TB <= 0) return SPK_ERR_ARG;
  if (H != W || (H != 7 && H != 8) || (Cout % WG_CO) || (Cin % WG_CI)) return SPK_ERR_UNSUPPORTED;
  const int tiles = (Cout / WG_CO) * (Cin / WG_CI);
  const int ks = wgrad_ksplit(TB, Cout, Cin);
  const long long n = (long long)Cout * 9 * Cin;
  if (ws_bytes < (long long)ks * (n + Cout) * 4) return SPK_ERR_ARG;
  WgArgs a;
  a.gy = gy_cl; a.s = spikes_cl; a.part = ws;
  a.part_gb = gb_out_or_null ? ws + (long long)ks * n : nullptr; a.TB = TB; a.Cout = Cout; a.Cin = Cin; a.ksplit = ks;
  if ((long long)lds_bytes(H) > spk_lds_limit()) return SPK_ERR_UNSUPPORTED;
  if (H == 7) hipLaunchKernelGGL(wgrad3x3_bf16_kernel<7>, dim3(tiles * ks), dim3(NTHR), (size_t)lds_bytes(7), stream, a);
  else hipLaunchKernelGGL(wgrad3x3_bf16_kernel<8>, dim3(tiles * ks), dim3(NTHR), (size_t)lds_bytes(8), stream, a);
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + Cout + 63) / 64 > 8192 ? 8192 : (n + Cout + 63) / 64)), dim3(256), 0, stream,
                     ws, gw_out, n, ks, a.part_gb, gb_out_or_null, Cout);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

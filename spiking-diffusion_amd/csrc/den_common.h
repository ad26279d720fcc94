// Helpers shared by the two matrix-core denoiser kernels (den_mfma.hip: int8 digit planes, den_mfma_fp6.hip: fp6).
#pragma once
#include "spk_common.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

#define SPK_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define SPK_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

// 16x16 bit-matrix transpose inside every 16-lane row (lane = row, bit = column) with DPP lane exchanges:
// lane^8 = row_mirror o row_half_mirror, lane^4 = row_half_mirror o quad-reverse, lane^2 / lane^1 = quad_perm.
// Round s (8, 4, 2, 1) swaps the off-diagonal s-bit blocks with lane ^ s.  Both lane parities run the SAME two instructions on
// per-lane constants -- the partner's word rotated by s towards the block it lands in (v_alignbit: right by s for the lanes with
// bit s set, right by 32 - s = left by s for the others) and merged under the keep mask (v_bfi) -- instead of a select between
// two shift-and-mask expressions, which hipcc compiled as two exec-masked branches per round (45 vector + 24 scalar
// instructions per transpose; now 14).  Bits 16..31 of the result are garbage (rotated-out blocks): callers use the low half.
#ifndef SPK_TR16_OLD
#define SPK_TR16_OLD 0          // 1: the rounds 1-3 form (a select between two shift-and-mask expressions per round), for A/B runs
#endif
__device__ __forceinline__ unsigned spk_transpose16_rows(unsigned x, int lane) {
  unsigned y;
#if SPK_TR16_OLD
  y = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true), 0x141, 0xF, 0xF, true);
  x = (lane & 8) ? (((y >> 8) & 0x00FFu) | (x & 0xFF00u)) : ((x & 0x00FFu) | ((y & 0x00FFu) << 8));
  y = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  x = (lane & 4) ? (((y >> 4) & 0x0F0Fu) | (x & 0xF0F0u)) : ((x & 0x0F0Fu) | ((y & 0x0F0Fu) << 4));
  y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
  x = (lane & 2) ? (((y >> 2) & 0x3333u) | (x & 0xCCCCu)) : ((x & 0x3333u) | ((y & 0x3333u) << 2));
  y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
  x = (lane & 1) ? (((y >> 1) & 0x5555u) | (x & 0xAAAAu)) : ((x & 0x5555u) | ((y & 0x5555u) << 1));
  return x;
#else
  // (opaque copy of the lane id: the three per-lane constants of a round are recomputed here -- three vector instructions --
  //  instead of being hoisted out of the caller's loops, where eight of them stayed live across the K loop: 256 registers + spills)
  int ln = lane;
  asm volatile("" : "+v"(ln));
#define SPK_TR16_ROUND(S, LOW)                                                                       \
  do {                                                                                               \
    const unsigned sh = (unsigned)ln & (unsigned)(S);            /* 0 or S */                         \
    const unsigned keep = (unsigned)(LOW) << sh;                                                     \
    const unsigned amt = (32u - (unsigned)(S)) + 2u * sh;        /* 32 - S, or 32 + S = S (mod 32) */ \
    const unsigned yr = __builtin_amdgcn_alignbit(y, y, amt);                                        \
    x = (x & keep) | (yr & ~keep);                                                                   \
  } while (0)
  y = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true), 0x141, 0xF, 0xF, true);
  SPK_TR16_ROUND(8, 0x00FFu);
  y = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  SPK_TR16_ROUND(4, 0x0F0Fu);
  y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
  SPK_TR16_ROUND(2, 0x3333u);
  y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
  SPK_TR16_ROUND(1, 0x5555u);
#undef SPK_TR16_ROUND
  return x;
#endif
}

// CU count of the current device, queried once (a constant of the machine; keeps device queries out of hipGraph capture)
static inline int spk_cu_count() {
  static const int cus = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      return v;
    return 256;
  }();
  return cus;
}

// LDS a workgroup of the current device may allocate (queried once, like the CU count): launches that ask for more than the
// 64 KB every device grants check it first and return SPK_ERR_UNSUPPORTED instead of failing inside the launch
static inline long long spk_lds_limit() {
  static const long long lim = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
      return (long long)v;
    return 65536ll;
  }();
  return lim;
}

// LDS-DMA of 16 bytes per lane: LDS[lds_base + lane * 16] = *(gsrc of this lane); lds_base is wave-uniform.
// Issued as inline assembly on purpose: hipcc treats the builtin (__builtin_amdgcn_global_load_lds) as a FLAT access
// that may touch LDS, and while one is in flight every LDS wait it inserts degrades to s_waitcnt lgkmcnt(0) -- which
// exposes the latency of the most recent fragment prefetch at every fourth MFMA group.  Hidden from the compiler the
// copy only moves the VM counter, so the caller MUST drain it itself (spk_dma_wait_all) before the barrier that
// publishes the data.
__device__ __forceinline__ void spk_dma16(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               : : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory", "m0");
}
// The same copy with the source given as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no vector
// address arithmetic per piece.
__device__ __forceinline__ void spk_dma16s(const void* sbase, unsigned voff, unsigned lds_base) {
  const unsigned long long b = (unsigned long long)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(voff), "s"(bs), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory", "m0");
}
// ... and with an explicit EXEC mask for the copy (all lanes must be active around the call: EXEC is restored to -1).
// A piece that covers only lanes 0..31 costs two scalar moves instead of a saveexec / branch / restore sequence.
__device__ __forceinline__ void spk_dma16s_masked(const void* sbase, unsigned voff, unsigned lds_base, unsigned long long mask) {
  const unsigned long long b = (unsigned long long)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
  const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mask), mhi = __builtin_amdgcn_readfirstlane((unsigned)(mask >> 32));
  const unsigned long long ms = ((unsigned long long)mhi << 32) | mlo;
  asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
               : : "v"(voff), "s"(bs), "s"(__builtin_amdgcn_readfirstlane(lds_base)), "s"(ms) : "memory", "m0");
}
__device__ __forceinline__ void spk_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned spk_lds_addr(const void* p) { return (unsigned)(size_t)SPK_LDS(p); }

// Layout of the position-list buffer written by spk_select_needed (psample.hip) and read by the listed form of the fp6v2
// kernel, for B image slots and R radii.  Bytes: [0, 64) header (u32 ticket of the list-building pass); per radius 16 int32
// (slots per tile-count class 1..6); per radius 6 x B int32 (the slots of each class); per radius B 64-byte records.
__host__ __device__ inline long long spk_need_off_cnt(int r) { return 64 + (long long)r * 64; }
__host__ __device__ inline long long spk_need_off_list(int B, int R, int r) {
  return 64 + (long long)R * 64 + (long long)r * 6 * B * 4;
}
__host__ __device__ inline long long spk_need_off_rec(int B, int R, int r) {
  const long long lists = 64 + (long long)R * 64 + (long long)R * 6 * B * 4;
  return (lists + 63) / 64 * 64 + (long long)r * B * 64;
}

// reflect-pad(1) + conv3x3 + bias + ReLU (+ fused nearest-x2 input / 2x2 max-pool output) on SP16 activations:
// the interior layers of the encoders / decoders in f16x3 mode (conv3x3_f16.hip explains the split arithmetic;
// conv_f16_dev.h the SP16 format).  Same results, bit for bit, as conv3x3_f16_kernel on the fp32 form of the input.
//
// What SP16 buys: the consumer no longer converts anything.  Activation halo tiles AND weight slabs go
// global -> LDS by DMA (global_load_lds_dwordx4: per-lane global address, wave-uniform LDS base + lane * 16), so
// staging costs no VGPRs, no VALU and no ds_write, and a 16-channel chunk can be in flight into the second LDS
// stage while the matrix cores work on the first:
//
//   persistent workgroup (one per CU, 8 waves = 32 x 16 output pixels, wave w owns rows 2w, 2w+1); work units are
//   (tile, cout group) pairs dealt out per XCD, a job is one 16-channel chunk of a unit; per job
//       s_waitcnt vmcnt(0); s_barrier      -- this job's stage has landed for every wave, the other stage is free
//       9 taps x CT x 2 x 3 MFMAs (32x32x16 f16) on this stage, and INSIDE the tap loop, a slice per tap,
//         - the DMA of the NEXT job (also across tile boundaries: no exposed prologue)
//         - the epilogue of the PREVIOUS tile (its accumulators are parked in registers when its last chunk ends)
//   ONE barrier per chunk (the register-staged kernel needs two and converts between them).
//   Why the epilogue is deferred: persistent workgroups with equal job lists run in lockstep, so an epilogue at the
//   tile boundary makes all 256 CUs store at once (measured: 20-45 k cycles per tile with the matrix cores idle, HBM
//   write-bound) and then all compute with HBM idle.  Spread over the next tile's MFMAs the stores are free.
//
// LDS (16-B units): stage = act[640 pixel slots][4 pieces, XOR-swizzled] (612 halo pixels of the 34 x 18 tile)
//                          + wgt[tap][hl][kh][COW];  2 stages + bias = 154 KB at COW = 64.
// Every MFMA operand is one conflict-free ds_read_b128 (see sp_slot below for the activation layout).
// 128 and more couts run as cout groups of 64 (a 128-wide weight slab would not leave room for the second stage); the
// groups of a tile are separate work units on the same XCD, so the activation tile is shared through that XCD's L2.
#include "wct_common.h"
#include "conv_f16_dev.h"
#include <cstdlib>

namespace {

constexpr int SPH = 16;
constexpr int SP_NPH = FHW * (SPH + 2);      // 612
constexpr int SP_NBLK = (SP_NPH + 63) / 64;  // 10
constexpr int SP_NPP = SP_NBLK * 64;         // 640 (10240 B == 0 mod 256)
constexpr int SP_ACT_DMA = (SP_NPH * 4 + 63) / 64;   // 39 wave-instructions per chunk for the activations
// Activation stage in LDS: pixel-major, 64 B per halo pixel = the four 16-byte pieces q = 2 kh + hl of the chunk, stored at
// slot 4 pix + (q ^ ((pix >> 2) & 3)).  Why: a DMA wave-instruction writes 64 CONSECUTIVE 16-byte LDS slots, so with this
// layout its lanes read 16 pixels x 64 contiguous bytes (16 half-used 128-B lines per instruction); the plane layout
// act[q][pix] made every lane fetch 16 B of a different line (64 lines per instruction, 8x L2->L1 read amplification,
// measured as the difference between this kernel and its skeleton in tools/experiments/mfma_loop.hip).  The XOR keeps
// the MFMA operand reads conflict-free: the 16 lanes of a ds_read_b128 group read 16 pixels whose indices are distinct
// mod 16, i.e. distinct (pix & 3, (pix >> 2) & 3) pairs -> 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int sp_slot(int pix, int q) { return pix * 4 + (q ^ ((pix >> 2) & 3)); }

struct SpArgs {
  const char* in;        // SP16 (conv_f16_dev.h): [cin / 16] planes of 64-byte pixel records
  char* out;             // SP16 or fp32 NHWC
  const u32x4* wpk;      // [chunk][9][hl][kh][cout_pad] x 16 B
  const float* bias;     // [cout_pad]
  const float* inv_scale_ptr;
  float inv_scale;
  int H, W, inW, inH;
  int cin, cout, cin_chunks, cout_pad, groups;
  int tiles_x, tiles_y, up_in, relu;
  unsigned* sat;         // sticky saturation counter of the context (may be null)
};

typedef __attribute__((address_space(3))) void* lds_ptr;

// (A 16-wave variant -- each wave one cout tile, four waves per SIMD -- was measured 25 % SLOWER: spills at the 128
// VGPR cap, 1.5x the LDS operand reads and a 16-wave barrier.  Kept out.)
// GROUPS (more than one cout group, i.e. >= 128 couts) changes no code: it gives those launches their own kernel name
// so that profiler rows (rocprofv3, PMC) can be matched to the library's profile families.
#ifdef WCT_SP_TIMING   // tools/experiments/sp_timing.sh: shader-clock cycles per phase, wave 0 of every workgroup, summed over all launches
__device__ unsigned long long g_sp_t[4];   // wait for the stage + barrier | tap loop (MFMA + riding DMA / epilogue slices) | rest | jobs
#define SP_STAMP(i) do { if (threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         spt[i] += t_ - sptl; sptl = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define SP_STAMP(i)
#endif

template <int CT, bool POOL, bool OUTF32, bool GROUPS>
__global__ __launch_bounds__(512) void conv3x3_sp_kernel(SpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int COW = CT * 32;
  constexpr int NWV = 8, CPW = CT;                     // waves per workgroup, cout tiles per wave
  constexpr int STAGE16 = 4 * SP_NPP + 36 * COW;
  constexpr int W_DMA = 36 * COW / 64;                 // weight wave-instructions per chunk (36 / 18)
  constexpr int W_PER_WAVE = (W_DMA + NWV - 1) / NWV;
  constexpr int SP_ACT_PER_WAVE = (SP_ACT_DMA + NWV - 1) / NWV;
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  float* biasL = reinterpret_cast<float*>(lds + 2 * STAGE16);   // [cout_pad]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int rw = wave, c0 = 0;                               // row pair / first cout tile of this wave
  const int ntiles = a.tiles_x * a.tiles_y;
  for (int e = tid; e < a.cout_pad; e += NWV * 64) biasL[e] = a.bias[e];
  const float inv = a.inv_scale_ptr ? *a.inv_scale_ptr : a.inv_scale;
  const size_t in_plane = sp16_plane_bytes(a.inH, a.inW);
  // saturation record (conv_f16_dev.h SatTrack): this kernel sits at 256 VGPRs, and any per-lane running maximum spilled
  // 7..18 of them (80 -> 92 us per launch of the 64-cout layers) -- so the "|x| > 65504" lane masks are OR-ed in SCALAR
  // registers instead (v_cmp + s_or_b64 per value; the epilogue runs with all 64 lanes active)
  unsigned long long satmask = 0ull;
  const float lob = a.relu ? 0.f : -65504.f;   // lower clamp bound of the SP16 epilogue: ReLU rides on the range clamp

  // ---- DMA of one job into a stage.  Activations: wave-instruction idx = wave + 8 i covers pixel block idx % 10 of
  // plane idx / 10; each lane's pixel offset for the wave's five blocks is recomputed when the tile changes.
  size_t poff[SP_ACT_PER_WAVE];
  auto tile_offsets = [&](int tile) {
    const int ty0 = (tile / a.tiles_x) * SPH, tx0 = (tile % a.tiles_x) * FTW;
#pragma unroll
    for (int i = 0; i < SP_ACT_PER_WAVE; ++i) {
      int idx = wave + NWV * i;
      idx = idx < SP_ACT_DMA ? idx : SP_ACT_DMA - 1;   // surplus waves re-send the last block (same bytes, same place)
      const int slot_pix = idx * 16 + (lane >> 2);      // LDS slot idx * 64 + lane = 4 slot_pix + (lane & 3)
      const int q = (lane & 3) ^ ((slot_pix >> 2) & 3);  // the piece that belongs there
      const int pix = slot_pix < SP_NPH ? slot_pix : SP_NPH - 1;
      const int py = pix / FHW, px = pix - py * FHW;
      int gy = reflect_clamp(ty0 - 1 + py, a.H), gx = reflect_clamp(tx0 - 1 + px, a.W);
      if (a.up_in) { gy >>= 1; gx >>= 1; }
      poff[i] = ((size_t)gy * a.inW + gx) * 64 + q * 16;   // within a chunk plane
    }
  };
  // slice i (0 .. SP_ACT_PER_WAVE - 1) of a job's DMA: one activation and (if any is left) one weight wave-instruction
  auto issue_slice = [&](int i, int ch, int grp, int stage) {
    u32x4* act = lds + stage * STAGE16;
    u32x4* wgt = act + 4 * SP_NPP;
    {
      int idx = wave + NWV * i;
      idx = idx < SP_ACT_DMA ? idx : SP_ACT_DMA - 1;
      __builtin_amdgcn_global_load_lds(a.in + (size_t)ch * in_plane + poff[i], (lds_ptr)(act + idx * 64), 16, 0, 0);
    }
    if (i < W_PER_WAVE) {
      int idx = wave + NWV * i;
      idx = idx < W_DMA ? idx : W_DMA - 1;
      const u32x4* g;
      if constexpr (COW == 64) g = a.wpk + ((size_t)(ch * 36 + idx) * a.cout_pad + grp * 64 + lane);
      else g = a.wpk + ((size_t)(ch * 36 + idx * 2 + (lane >> 5)) * a.cout_pad + grp * 32 + (lane & 31));
      __builtin_amdgcn_global_load_lds(g, (lds_ptr)(wgt + idx * 64), 16, 0, 0);
    }
  };
  static_assert(W_PER_WAVE <= SP_ACT_PER_WAVE, "weight slices ride on the activation slices");

  // ---- one piece (cout tile c, 8-channel group q, tile row p) of a finished tile's epilogue.
  // D: col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cout)
  constexpr int NPIECE = CPW * 4 * (POOL ? 1 : 2);
  // The four pieces q = 0..3 of one (cout tile, row) complete the same 128-byte lines of the output, so they are issued
  // in the SAME tap (4 pieces per tap on the first NPIECE / 4 taps): spread over four taps the lines were written back
  // half-filled in between (PMC: +18 % HBM write traffic).
  constexpr int PPT = 4;
  // The pieces ride on the LAST taps of a job, after the job's DMA slices (taps 0 .. SP_ACT_PER_WAVE - 1).  (Tried on top of
  // that order: waiting for vmcnt(NPIECE) instead of vmcnt(0) at the top of a job -- gfx9 counts stores in vmcnt and retires
  // it in order, so the stage has landed while the stores may still be in flight.  No gain: the 2.7 k of 11.7 k cycles per
  // job that wave 0 spends at the wait + barrier (tools/experiments/sp_timing.sh) are the other wave of its SIMD still on
  // the matrix core, not write acknowledgements.)
  constexpr int EP_TAP0 = 9 - (NPIECE + PPT - 1) / PPT;
  static_assert(EP_TAP0 >= SP_ACT_PER_WAVE, "epilogue slices after the DMA slices");
  auto epilogue_piece = [&](const f32x16 (&r)[CPW][2], int piece, int ty0, int tx0, int pgrp) {
    const int gx = tx0 + li;
    const int oH = POOL ? a.H >> 1 : a.H, oW = POOL ? a.W >> 1 : a.W;
    const int q = piece & 3, cp = piece >> 2, p = POOL ? 0 : cp & 1, c = POOL ? cp : cp >> 1;
    const int co = pgrp * COW + (c0 + c) * 32 + 8 * q + 4 * kh;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(biasL + co);
    f32x4 x;
    int oy, ox;
    bool ok;
    // OUTF32: x = the activation.  SP16 out: x = the PRE-activation; ReLU rides on the split's range clamp (lower bound lob)
    if constexpr (POOL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t = fmaxf(r[c][0][4 * q + k], r[c][1][4 * q + k]);
        x[k] = fmaxf(t, lane_xor1(t));
      }
      x = fma4(x, inv, bias);
      if (OUTF32 && a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
      }
      oy = (ty0 + rw * 2) >> 1; ox = gx >> 1;
      ok = !(li & 1) && oy < oH && ox < oW && co < a.cout;
    } else {
      x = fma4(f32x4{r[c][p][4 * q], r[c][p][4 * q + 1], r[c][p][4 * q + 2], r[c][p][4 * q + 3]}, inv, bias);
      if (OUTF32 && a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
      }
      oy = ty0 + rw * 2 + p; ox = gx;
      ok = oy < oH && ox < oW && co < a.cout;
    }
    if constexpr (OUTF32) {
      if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + ((size_t)oy * oW + ox) * a.cout + co) = x;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) satmask |= __ballot(x[k] > 65504.f);
      if (!a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) satmask |= __ballot(x[k] < -65504.f);
      }
      const u32x4 w = sp16_pair_exchange_lob(x, lob);
      if (ok) *reinterpret_cast<u32x4*>(a.out + sp16_piece(sp16_plane_bytes(oH, oW), (size_t)oy * oW + ox, co >> 3, kh)) = w;
    }
  };

  f32x16 acc[CPW][2], pend[CPW][2];
  // job state (uniform).  Work units are (tile, cout group) pairs, dealt out per XCD: workgroup b runs on XCD b & 7, which
  // owns a contiguous range of tiles (as xcd_swizzle does) and all their groups; its workgroups walk that unit list with
  // stride grid/8, group fastest -- the groups of one tile run side by side on ONE XCD and share the activation tile in
  // its L2.  (Units used to be whole tiles with the groups in sequence: a 135x240x512 layer -- 72 tiles x 8 groups -- kept
  // 72 of 256 CUs busy, 135 TF instead of 400+.)
  const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
  const int tbase = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int ngroups = GROUPS ? a.groups : 1;
  const int nunits = (xq + (xcd < xr ? 1 : 0)) * ngroups, ustep = gridDim.x >> 3;
  int v = blockIdx.x >> 3, ch = 0, stage = 0;
  if (v >= nunits) return;
  int grp = GROUPS ? v % ngroups : 0;
  int tile = tbase + (GROUPS ? v / ngroups : v);
  int dma_tile = tile;          // tile the offsets in poff[] belong to
  int pgrp = 0, pty0 = 0, ptx0 = 0;   // group / origin of the tile whose finished accumulators wait in pend[]
  bool have_pend = false;
  const unsigned txm = tile_div_magic(a.tiles_x);
  tile_offsets(tile);
#pragma unroll
  for (int i = 0; i < SP_ACT_PER_WAVE; ++i) issue_slice(i, 0, grp, 0);
#ifdef WCT_SP_TIMING
  unsigned long long spt[4] = {0, 0, 0, 0}, sptl = __builtin_amdgcn_s_memtime();
#endif
  while (true) {
    // next job
    int nv = v, ngrp = grp, nch = ch + 1;
    if (nch == a.cin_chunks) { nch = 0; nv = v + ustep; ngrp = GROUPS ? nv % ngroups : 0; }
    const bool more = nv < nunits;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    SP_STAMP(0);
    if (more) {
      const int ntile = nv == v ? tile : tbase + (GROUPS ? nv / ngroups : nv);
      if (ntile != dma_tile) { tile_offsets(ntile); dma_tile = ntile; }
    }
    if (ch == 0) {
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.f;
    }
    const u32x4* act = lds + stage * STAGE16;
    const u32x4* wgt = act + 4 * SP_NPP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pix = (rw * 2 + p + dy) * FHW + li + dx;
        bh[p] = __builtin_bit_cast(f16x8, act[sp_slot(pix, 2 * kh)]);
        bl[p] = __builtin_bit_cast(f16x8, act[sp_slot(pix, 2 * kh + 1)]);
      }
      f16x8 ah[CPW], al[CPW];
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        ah[c] = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 0) * 2 + kh) * COW + (c0 + c) * 32 + li]);
        al[c] = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 1) * 2 + kh) * COW + (c0 + c) * 32 + li]);
      }
      // riding on this tap: a slice of the next job's DMA and a slice of the previous tile's epilogue
      if (tap < SP_ACT_PER_WAVE && more) issue_slice(tap, nch, ngrp, stage ^ 1);
      if (tap >= EP_TAP0 && have_pend) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
          if ((tap - EP_TAP0) * PPT + k < NPIECE) epilogue_piece(pend, (tap - EP_TAP0) * PPT + k, pty0, ptx0, pgrp);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[c] : ah[c], term == 1 ? bl[p] : bh[p], acc[c][p], 0, 0, 0);
    }
    SP_STAMP(1);
    have_pend = false;
    if (ch + 1 == a.cin_chunks) {   // park the finished tile; its stores ride on the next job
#pragma unroll
      for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) pend[c][p] = acc[c][p];
      int trow_, tcol_;
      tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
      pty0 = trow_ * SPH; ptx0 = tcol_ * FTW; pgrp = grp; have_pend = true;
    }
#ifdef WCT_SP_TIMING
    spt[3] += 1;
#endif
    if (!more) break;
    if (nv != v) { v = nv; tile = dma_tile; }
    grp = ngrp; ch = nch; stage ^= 1;
    SP_STAMP(2);
  }
  if (have_pend) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) epilogue_piece(pend, k, pty0, ptx0, pgrp);
  }
  if (satmask != 0ull && lane == 0 && a.sat) sat_raise(a.sat);
#ifdef WCT_SP_TIMING
  SP_STAMP(2);
  if (threadIdx.x == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_sp_t[i], spt[i]);
#endif
}

// =====================================================================================================================
// The HBM-side layers (32 couts, 16 or 32 input channels: conv21 / conv22 of the encoders, conv22 of the decoders) with THREE
// activation stages and resident weights (round 4).  In the kernel above a job's DMA is issued during the previous job, i.e. one tile
// (16 cin) or half a tile (32 cin) ahead: at 1728 matrix-core cycles per wave and job a CU then spends most of a job waiting at
// vmcnt(0) -- for the next stage, and for the previous tile's stores, which gfx9 counts in the same in-order counter -- with one
// stage of loads in flight: 3.6-3.8 TB/s algorithmic on the 1080 x 1920 maps, 0.6 of what a copy reaches.  Here
//   * the weights of ALL chunks (<= 2 x 18 KB) are loaded once per workgroup and stay in LDS,
//   * job j issues the DMA of job j + 2 into the third stage, on its LAST taps, and the parked epilogue of the previous tile on its
//     FIRST taps, so that at the top of job j + 1 `s_waitcnt vmcnt(5)` (the wave's five newest vector-memory operations = the DMA
//     of job j + 2, issued unconditionally) waits for job j + 1's stage and for the stores, while a whole stage stays in flight.
// Same tiles, same MFMA order, same epilogue as the kernel above: results are bit-identical (tests/test_hip_parity.py).
// (Also built and measured, not kept: the opposite order -- DMA on the first taps, the stores LAST and unconditional (lanes without an
// output pixel writing to a sink), `vmcnt(5 + stores)` so that the stores stay in flight across the barrier as well.  4 % slower on the
// same box, 0.705 against 0.678 ms per step for the family: it is the stage of loads in flight that pays, not the stores' latency.
// And the opposite design altogether -- an OCCUPANCY kernel: one 32 x 8 tile per 4-wave workgroup, not persistent, the whole halo and
// the weights DMA'd at once, one barrier, 2-3 workgroups per CU hiding each other's loads and stores (the store probe,
// tools/experiments/write_bw.hip, favours more store-issuing waves per CU) -- bit-identical, and 6 % / 15 % SLOWER for the plain / pooled
// layers (0.838 / 0.69 against 0.79 / 0.60 ms per step, profiles/r04_spq_occupancy_kernel_ab_slow_box.txt): a third more halo, the
// weights reloaded per 256 pixels, 8 100 workgroups to dispatch.)
// LDS (16-B units): 3 x act[640 x 4] + wgt[chunks][36 x 32] + bias = 141 KB (16 cin) / 159 KB (32 cin).
template <bool POOL, bool OUTF32>
__global__ __launch_bounds__(512) void conv3x3_sp3_kernel(SpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int COW = 32, NWV = 8, CPW = 1, NST = 3;
  constexpr int ACT16 = 4 * SP_NPP;                    // one activation stage
  constexpr int WCH16 = 36 * COW;                      // one chunk's weights
  constexpr int W_DMA = WCH16 / 64;                    // 18 wave-instructions per chunk
  constexpr int SP_ACT_PER_WAVE = (SP_ACT_DMA + NWV - 1) / NWV;   // 5
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  u32x4* wgt_all = lds + NST * ACT16;
  float* biasL = reinterpret_cast<float*>(wgt_all + a.cin_chunks * WCH16);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int rw = wave;
  const int ntiles = a.tiles_x * a.tiles_y;
  for (int e = tid; e < a.cout_pad; e += NWV * 64) biasL[e] = a.bias[e];
  const float inv = a.inv_scale_ptr ? *a.inv_scale_ptr : a.inv_scale;
  const size_t in_plane = sp16_plane_bytes(a.inH, a.inW);
  unsigned long long satmask = 0ull;
  const float lob = a.relu ? 0.f : -65504.f;

  size_t poff[SP_ACT_PER_WAVE];
  auto tile_offsets = [&](int tile) {
    const int ty0 = (tile / a.tiles_x) * SPH, tx0 = (tile % a.tiles_x) * FTW;
#pragma unroll
    for (int i = 0; i < SP_ACT_PER_WAVE; ++i) {
      int idx = wave + NWV * i;
      idx = idx < SP_ACT_DMA ? idx : SP_ACT_DMA - 1;
      const int slot_pix = idx * 16 + (lane >> 2);
      const int q = (lane & 3) ^ ((slot_pix >> 2) & 3);
      const int pix = slot_pix < SP_NPH ? slot_pix : SP_NPH - 1;
      const int py = pix / FHW, px = pix - py * FHW;
      int gy = reflect_clamp(ty0 - 1 + py, a.H), gx = reflect_clamp(tx0 - 1 + px, a.W);
      if (a.up_in) { gy >>= 1; gx >>= 1; }
      poff[i] = ((size_t)gy * a.inW + gx) * 64 + q * 16;
    }
  };
  auto issue_act = [&](int i, int ch, int stage) {
    int idx = wave + NWV * i;
    idx = idx < SP_ACT_DMA ? idx : SP_ACT_DMA - 1;
    __builtin_amdgcn_global_load_lds(a.in + (size_t)ch * in_plane + poff[i], (lds_ptr)(lds + stage * ACT16 + idx * 64), 16, 0, 0);
  };

  constexpr int NPIECE = CPW * 4 * (POOL ? 1 : 2), PPT = 4;
  auto epilogue_piece = [&](const f32x16 (&r)[CPW][2], int piece, int ty0, int tx0) {
    const int gx = tx0 + li;
    const int oH = POOL ? a.H >> 1 : a.H, oW = POOL ? a.W >> 1 : a.W;
    const int q = piece & 3, cp = piece >> 2, p = POOL ? 0 : cp & 1;
    const int co = 8 * q + 4 * kh;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(biasL + co);
    f32x4 x;
    int oy, ox;
    bool ok;
    if constexpr (POOL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t = fmaxf(r[0][0][4 * q + k], r[0][1][4 * q + k]);
        x[k] = fmaxf(t, lane_xor1(t));
      }
      x = fma4(x, inv, bias);
      if (OUTF32 && a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
      }
      oy = (ty0 + rw * 2) >> 1; ox = gx >> 1;
      ok = !(li & 1) && oy < oH && ox < oW && co < a.cout;
    } else {
      x = fma4(f32x4{r[0][p][4 * q], r[0][p][4 * q + 1], r[0][p][4 * q + 2], r[0][p][4 * q + 3]}, inv, bias);
      if (OUTF32 && a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaxf(x[k], 0.f);
      }
      oy = ty0 + rw * 2 + p; ox = gx;
      ok = oy < oH && ox < oW && co < a.cout;
    }
    if constexpr (OUTF32) {
      if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + ((size_t)oy * oW + ox) * a.cout + co) = x;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) satmask |= __ballot(x[k] > 65504.f);
      if (!a.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) satmask |= __ballot(x[k] < -65504.f);
      }
      const u32x4 w = sp16_pair_exchange_lob(x, lob);
      if (ok) *reinterpret_cast<u32x4*>(a.out + sp16_piece(sp16_plane_bytes(oH, oW), (size_t)oy * oW + ox, co >> 3, kh)) = w;
    }
  };

  f32x16 acc[CPW][2], pend[CPW][2];
  const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
  const int tbase = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int nunits = xq + (xcd < xr ? 1 : 0), ustep = gridDim.x >> 3;
  const int nch = a.cin_chunks;
  int v = blockIdx.x >> 3, ch = 0, stage = 0;
  if (v >= nunits) return;
  // job (v, ch) -> its successor; a job exists while v < nunits
  auto succ = [&](int& jv, int& jc) { if (++jc == nch) { jc = 0; jv += ustep; } };
  // ---- prologue: weights of every chunk (resident), then the stages of jobs 0 and 1
  for (int idx = wave; idx < nch * W_DMA; idx += NWV) {
    const int c = idx / W_DMA, k = idx - c * W_DMA;
    const u32x4* g = a.wpk + ((size_t)(c * 36 + k * 2 + (lane >> 5)) * a.cout_pad + (lane & 31));
    __builtin_amdgcn_global_load_lds(g, (lds_ptr)(wgt_all + c * WCH16 + k * 64), 16, 0, 0);
  }
  int dma_tile = tbase + v;
  tile_offsets(dma_tile);
#pragma unroll
  for (int i = 0; i < SP_ACT_PER_WAVE; ++i) issue_act(i, 0, 0);
  int v1 = v, c1 = 0;                       // job + 1
  succ(v1, c1);
  bool have1 = v1 < nunits;
  if (have1) {
    if (tbase + v1 != dma_tile) { dma_tile = tbase + v1; tile_offsets(dma_tile); }
#pragma unroll
    for (int i = 0; i < SP_ACT_PER_WAVE; ++i) issue_act(i, c1, 1);
  }
  int pty0 = 0, ptx0 = 0;
  bool have_pend = false;
  const unsigned txm = tile_div_magic(a.tiles_x);
  while (true) {
    // the stage of THIS job has landed once at most the DMA of the job after it (the wave's newest five operations) is outstanding
    if (have1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int v2 = v1, c2 = c1;                   // job + 2: its DMA goes into the stage the previous job has just released
    bool have2 = false;
    if (have1) { succ(v2, c2); have2 = v2 < nunits; }
    if (have2 && tbase + v2 != dma_tile) { dma_tile = tbase + v2; tile_offsets(dma_tile); }
    const int stage2 = stage == 0 ? 2 : stage - 1;        // (stage + 2) % 3
    if (ch == 0) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][p][r] = 0.f;
    }
    const u32x4* act = lds + stage * ACT16;
    const u32x4* wgt = wgt_all + ch * WCH16;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pix = (rw * 2 + p + dy) * FHW + li + dx;
        bh[p] = __builtin_bit_cast(f16x8, act[sp_slot(pix, 2 * kh)]);
        bl[p] = __builtin_bit_cast(f16x8, act[sp_slot(pix, 2 * kh + 1)]);
      }
      const f16x8 ah = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 0) * 2 + kh) * COW + li]);
      const f16x8 al = __builtin_bit_cast(f16x8, wgt[((tap * 2 + 1) * 2 + kh) * COW + li]);
      // first taps: the parked tile's stores; last taps: the DMA of job + 2 (so that it is the newest thing in the wave's vmcnt queue)
      if (tap * PPT < NPIECE && have_pend) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
          if (tap * PPT + k < NPIECE) epilogue_piece(pend, tap * PPT + k, pty0, ptx0);
      }
      if (tap >= 9 - SP_ACT_PER_WAVE && have2) issue_act(tap - (9 - SP_ACT_PER_WAVE), c2, stage2);
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          acc[0][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al : ah, term == 1 ? bl[p] : bh[p], acc[0][p], 0, 0, 0);
    }
    have_pend = false;
    if (ch + 1 == nch) {
#pragma unroll
      for (int p = 0; p < 2; ++p) pend[0][p] = acc[0][p];
      int trow_, tcol_;
      tile_rc(tbase + v, a.tiles_x, txm, trow_, tcol_);
      pty0 = trow_ * SPH; ptx0 = tcol_ * FTW; have_pend = true;
    }
    if (!have1) break;
    v = v1; ch = c1; v1 = v2; c1 = c2; have1 = have2;
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (have_pend) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) epilogue_piece(pend, k, pty0, ptx0);
  }
  if (satmask != 0ull && lane == 0 && a.sat) sat_raise(a.sat);
}

// =====================================================================================================================
// Layers BEHIND a nearest-x2 upsample (CONV_UP_IN), on the low-resolution grid.
// The 3x3 convolution of the upsampled map is, per output parity (a, b), a 2x2 convolution of the low-resolution map with
// summed taps (wct_api.hip pack_up_phase_f16): 4 instead of 9 products per output and channel pair.  The kernel above gathered
// the upsampled 34 x 18 halo (each low-resolution pixel up to four times) and ran nine taps; this one stages the LOW-RESOLUTION
// 34 x 18 halo of a 32 x 16 low-resolution tile (= 64 x 32 outputs) and a work unit is (tile, 32-cout group, row parity a): its
// two accumulator sets are the column parities b = 0, 1 (the register budget of the 64-cout kernel above), its job walks the six
// patch offsets (dy in {a, a + 1}) x (dx in {0, 1, 2}), offset dx serving (b, j) = (0, dx) and (1, dx - 1): 48 MFMAs, 24
// activation and 16 weight reads per wave and chunk for 2 x 64 x 32 outputs, against 4 x 54 in the nine-tap form.
// Reflect padding of the upsampled map = CLAMPING the low-resolution coordinates.  Everything else -- DMA staging, one barrier
// per chunk, parked accumulators whose stores ride on the next job -- is the kernel above.
// Weights: [chunk][a][(b, i, j, hl, kh) = 32][cout_pad] x 16 B with their own power-of-two scale (ConvDesc::wup16 / inv_scale_up).
template <bool OUTF32>
__global__ __launch_bounds__(512) void conv3x3_sp_up_kernel(SpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int COW = 32, NWV = 8;
  constexpr int WSL = 32 * COW;                         // weight slots per stage (1024)
  constexpr int STAGE16 = 4 * SP_NPP + WSL;
  constexpr int W_DMA = WSL / 64, W_PER_WAVE = W_DMA / NWV;   // 16 pieces, 2 per wave
  constexpr int SP_ACT_PER_WAVE = (SP_ACT_DMA + NWV - 1) / NWV;
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  float* biasL = reinterpret_cast<float*>(lds + 2 * STAGE16);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5, rw = wave;
  const int ntiles = a.tiles_x * a.tiles_y;            // low-resolution tiles
  for (int e = tid; e < a.cout_pad; e += NWV * 64) biasL[e] = a.bias[e];
  const float inv = a.inv_scale;
  const size_t in_plane = sp16_plane_bytes(a.inH, a.inW);
  unsigned long long satmask = 0ull;
  const float lob = a.relu ? 0.f : -65504.f;

  size_t poff[SP_ACT_PER_WAVE];
  auto tile_offsets = [&](int tile) {
    const int ty0 = (tile / a.tiles_x) * SPH, tx0 = (tile % a.tiles_x) * FTW;
#pragma unroll
    for (int i = 0; i < SP_ACT_PER_WAVE; ++i) {
      int idx = wave + NWV * i;
      idx = idx < SP_ACT_DMA ? idx : SP_ACT_DMA - 1;
      const int slot_pix = idx * 16 + (lane >> 2);
      const int q = (lane & 3) ^ ((slot_pix >> 2) & 3);
      const int pix = slot_pix < SP_NPH ? slot_pix : SP_NPH - 1;
      const int py = pix / FHW, px = pix - py * FHW;
      int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
      gy = gy < 0 ? 0 : (gy >= a.inH ? a.inH - 1 : gy);
      gx = gx < 0 ? 0 : (gx >= a.inW ? a.inW - 1 : gx);
      poff[i] = ((size_t)gy * a.inW + gx) * 64 + q * 16;
    }
  };
  // unit group code: g32 = grp >> 1 (32-cout group), pa = grp & 1 (row parity)
  auto issue_slice = [&](int i, int ch, int grp, int stage) {
    u32x4* act = lds + stage * STAGE16;
    u32x4* wgt = act + 4 * SP_NPP;
    {
      int idx = wave + NWV * i;
      idx = idx < SP_ACT_DMA ? idx : SP_ACT_DMA - 1;
      __builtin_amdgcn_global_load_lds(a.in + (size_t)ch * in_plane + poff[i], (lds_ptr)(act + idx * 64), 16, 0, 0);
    }
    if (i < W_PER_WAVE) {
      const int idx = wave + NWV * i;                   // 64 slots = two of the 32 (b, i, j, hl, kh) rows
      const u32x4* g = a.wpk + ((size_t)((ch * 2 + (grp & 1)) * 32 + idx * 2 + (lane >> 5)) * a.cout_pad + (grp >> 1) * 32 + (lane & 31));
      __builtin_amdgcn_global_load_lds(g, (lds_ptr)(wgt + idx * 64), 16, 0, 0);
    }
  };

  // one piece of a finished unit's epilogue: k -> b = k & 1, q = (k >> 1) & 3, p = k >> 3; the four pieces of one iteration
  // (b, q & 1) x fixed (q >> 1, p) complete the same 128-byte lines of the output (two neighbouring pixel records)
  constexpr int NPIECE = 16, PPT = 4;
  auto epilogue_piece = [&](const f32x16 (&r)[2][2], int k, int ty0, int tx0, int pgrp) {
    const int b = k & 1, q = ((k >> 1) & 1) | (((k >> 2) & 1) << 1), p = k >> 3, pa = pgrp & 1;
    const int co = (pgrp >> 1) * COW + 8 * q + 4 * kh;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(biasL + co);
    f32x4 x = fma4(f32x4{r[b][p][4 * q], r[b][p][4 * q + 1], r[b][p][4 * q + 2], r[b][p][4 * q + 3]}, inv, bias);
    if (OUTF32 && a.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
    }
    const int oy = 2 * (ty0 + rw * 2 + p) + pa, ox = 2 * (tx0 + li) + b;
    const bool ok = oy < a.H && ox < a.W && co < a.cout;
    if constexpr (OUTF32) {
      if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + ((size_t)oy * a.W + ox) * a.cout + co) = x;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) satmask |= __ballot(x[e] > 65504.f);
      if (!a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) satmask |= __ballot(x[e] < -65504.f);
      }
      const u32x4 w = sp16_pair_exchange_lob(x, lob);
      if (ok) *reinterpret_cast<u32x4*>(a.out + sp16_piece(sp16_plane_bytes(a.H, a.W), (size_t)oy * a.W + ox, co >> 3, kh)) = w;
    }
  };

  f32x16 acc[2][2], pend[2][2];     // [column parity b][tile row p]
  const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
  const int tbase = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int ngroups = a.groups;     // 2 x (cout_pad / 32)
  const int nunits = (xq + (xcd < xr ? 1 : 0)) * ngroups, ustep = gridDim.x >> 3;
  int v = blockIdx.x >> 3, ch = 0, stage = 0;
  if (v >= nunits) return;
  int grp = v % ngroups;
  int tile = tbase + v / ngroups;
  int dma_tile = tile;
  int pgrp = 0, pty0 = 0, ptx0 = 0;
  bool have_pend = false;
  const unsigned txm = tile_div_magic(a.tiles_x);
  tile_offsets(tile);
#pragma unroll
  for (int i = 0; i < SP_ACT_PER_WAVE; ++i) issue_slice(i, 0, grp, 0);
  while (true) {
    int nv = v, ngrp = grp, nch = ch + 1;
    if (nch == a.cin_chunks) { nch = 0; nv = v + ustep; ngrp = nv % ngroups; }
    const bool more = nv < nunits;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (more) {
      const int ntile = nv == v ? tile : tbase + nv / ngroups;
      if (ntile != dma_tile) { tile_offsets(ntile); dma_tile = ntile; }
    }
    if (ch == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[b][p][r] = 0.f;
    }
    const u32x4* act = lds + stage * STAGE16;
    const u32x4* wgt = act + 4 * SP_NPP;
    const int pa = grp & 1;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int i = it / 3, dx = it - i * 3;          // patch row i (image row offset pa + i), column offset dx
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pix = (rw * 2 + p + pa + i) * FHW + li + dx;
        bh[p] = __builtin_bit_cast(f16x8, act[sp_slot(pix, 2 * kh)]);
        bl[p] = __builtin_bit_cast(f16x8, act[sp_slot(pix, 2 * kh + 1)]);
      }
      if (it < SP_ACT_PER_WAVE && more) issue_slice(it, nch, ngrp, stage ^ 1);
      if (it >= 2 && have_pend) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) epilogue_piece(pend, (it - 2) * PPT + k, pty0, ptx0, pgrp);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = dx - b;
        if (j < 0 || j > 1) continue;
        const f16x8 ah = __builtin_bit_cast(f16x8, wgt[((((b * 2 + i) * 2 + j) * 2 + 0) * 2 + kh) * COW + li]);
        const f16x8 al = __builtin_bit_cast(f16x8, wgt[((((b * 2 + i) * 2 + j) * 2 + 1) * 2 + kh) * COW + li]);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            acc[b][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al : ah, term == 1 ? bl[p] : bh[p], acc[b][p], 0, 0, 0);
      }
    }
    have_pend = false;
    if (ch + 1 == a.cin_chunks) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 2; ++p) pend[b][p] = acc[b][p];
      int trow_, tcol_;
      tile_rc(tile, a.tiles_x, txm, trow_, tcol_);
      pty0 = trow_ * SPH; ptx0 = tcol_ * FTW; pgrp = grp; have_pend = true;
    }
    if (!more) break;
    if (nv != v) { v = nv; tile = tbase + nv / ngroups; }
    grp = ngrp; ch = nch; stage ^= 1;
  }
  if (have_pend) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) epilogue_piece(pend, k, pty0, ptx0, pgrp);
  }
  if (satmask != 0ull && lane == 0 && a.sat) sat_raise(a.sat);
}

template <typename K>
hipError_t launch_sp(K k, const SpArgs& a, size_t lds, hipStream_t s, int threads) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // (tile, group) units per XCD queue x 8 queues, capped at one workgroup per CU; a multiple of 8 (workgroup b -> XCD b & 7)
  const int ntiles = a.tiles_x * a.tiles_y, want = 8 * ((ntiles + 7) / 8) * a.groups, cus = (num_cus() & ~7) > 8 ? (num_cus() & ~7) : 8;
  const int grid = want < cus ? want : cus;
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a);
  return hipGetLastError();
}

}  // namespace

bool conv_sp_supported(const ConvDesc& d) {
  // WCT_SP_DMA_MASK (experiments): which layer families take the DMA kernel -- 1: 32 couts, 2: 32 couts + pool,
  // 4: 64 couts, 8: 64 couts + pool, 16: >= 128 couts.  Measured (4K bench, ms per step, DMA vs register-staged on the
  // same SP16 input): 32: 1.29 / 1.30, 32+pool: 0.72 / 0.59, 64: 1.90 / 1.95, 64+pool: 0.33 / 0.33, >=128: 1.31 / 1.70
  // -> default 29: everything but the pooled 32-cout layers (re-measured after the chunk-planar SP16 layout and the
  // per-XCD work units: 0.565 / 0.565 ms there now -- a tie, left as it was).  Round 4: with the three-stage kernel
  // (conv3x3_sp3_kernel) the pooled 32-cout layers run 0.551 against 0.578 ms there -> default 31, everything.
  static const int mask = [] { const char* e = wct_debug_env("WCT_SP_DMA_MASK"); return e ? atoi(e) : 31; }();
  const bool pool = d.flags & CONV_POOL_OUT;
  const int fam = d.cout_pad >= 128 ? 16 : d.cout_pad == 64 ? (pool ? 8 : 4) : (pool ? 2 : 1);
  if (!(mask & fam)) return false;
  return d.wpk16 && !(d.flags & (CONV_IN_NCHW3 | CONV_OUT_NCHW3)) && (d.cin % 16) == 0 && d.cout_pad >= 32 && (d.cout % 8) == 0 &&
         d.cout_pad <= 512;
}

// does launch_conv3x3_sp run this layer in the upsample form (per-parity 2x2 convolutions on the low-resolution grid)?
bool conv_sp_up_form(const ConvDesc& d, int H, int W) {
  static const int up_env = [] { const char* e = wct_debug_env("WCT_SP_UP"); return e ? atoi(e) : 1; }();
  return up_env && (d.flags & CONV_UP_IN) && d.wup16 && !(d.flags & CONV_POOL_OUT) && !d.inv_scale_ptr && !(H & 1) && !(W & 1) &&
         (d.cout_pad % 32) == 0;
}

// in: SP16 (CONV_IN_SP16 must be set); out: SP16 (CONV_OUT_SP16) or fp32 NHWC
hipError_t launch_conv3x3_sp(const ConvDesc& d, const void* in, void* out, int H, int W, hipStream_t s) {
  if (H < 2 || W < 2 || !conv_sp_supported(d) || !(d.flags & CONV_IN_SP16)) return hipErrorInvalidValue;
  SpArgs a;
  a.in = reinterpret_cast<const char*>(in); a.out = reinterpret_cast<char*>(out);
  a.wpk = reinterpret_cast<const u32x4*>(d.wpk16); a.bias = d.bias;
  a.inv_scale_ptr = d.inv_scale_ptr; a.inv_scale = d.inv_scale;
  a.H = H; a.W = W;
  a.up_in = (d.flags & CONV_UP_IN) ? 1 : 0;
  a.inW = a.up_in ? W / 2 : W; a.inH = a.up_in ? H / 2 : H;
  a.cin = d.cin; a.cout = d.cout; a.cin_chunks = d.cin_chunks; a.cout_pad = d.cout_pad;
  a.tiles_x = (W + FTW - 1) / FTW; a.tiles_y = (H + SPH - 1) / SPH;
  a.relu = (d.flags & CONV_NO_RELU) ? 0 : 1;
  a.sat = d.sat;
  const bool pool = d.flags & CONV_POOL_OUT, f32 = !(d.flags & CONV_OUT_SP16);
  // behind an upsample: per-parity 2x2 convolutions on the low-resolution grid (conv3x3_sp_up_kernel)
  if (conv_sp_up_form(d, H, W)) {
    a.wpk = reinterpret_cast<const u32x4*>(d.wup16);
    a.inv_scale = d.inv_scale_up;
    a.tiles_x = (a.inW + FTW - 1) / FTW; a.tiles_y = (a.inH + SPH - 1) / SPH;
    a.groups = 2 * (d.cout_pad / 32);
    const size_t ldsu = (size_t)2 * (4 * SP_NPP + 32 * 32) * 16 + (size_t)d.cout_pad * sizeof(float);   // 114.8 KB
    return f32 ? launch_sp(conv3x3_sp_up_kernel<true>, a, ldsu, s, 512) : launch_sp(conv3x3_sp_up_kernel<false>, a, ldsu, s, 512);
  }
  const int ct = (d.cout_pad % 64 == 0) ? 2 : 1;
  a.groups = d.cout_pad / (ct * 32);
  // 32 couts from 16 / 32 input channels: three activation stages, resident weights (conv3x3_sp3_kernel); WCT_SP3=0 keeps the two-stage kernel
  static const int sp3_env = [] { const char* e = wct_debug_env("WCT_SP3"); return e ? atoi(e) : 1; }();
  if (sp3_env && d.cout_pad == 32 && d.cin_chunks <= 2) {
    const size_t lds3 = ((size_t)3 * 4 * SP_NPP + (size_t)d.cin_chunks * 36 * 32) * 16 + (size_t)d.cout_pad * sizeof(float);
    if (pool) return f32 ? launch_sp(conv3x3_sp3_kernel<true, true>, a, lds3, s, 512) : launch_sp(conv3x3_sp3_kernel<true, false>, a, lds3, s, 512);
    return f32 ? launch_sp(conv3x3_sp3_kernel<false, true>, a, lds3, s, 512) : launch_sp(conv3x3_sp3_kernel<false, false>, a, lds3, s, 512);
  }
  const size_t lds = (size_t)2 * (4 * SP_NPP + 36 * ct * 32) * 16 + (size_t)d.cout_pad * sizeof(float);
#define WCT_SP_CASE(CTV, GV)                                                                                          \
  if (ct == CTV && (a.groups > 1) == GV) {                                                                            \
    if (pool) return f32 ? launch_sp(conv3x3_sp_kernel<CTV, true, true, GV>, a, lds, s, 512) : launch_sp(conv3x3_sp_kernel<CTV, true, false, GV>, a, lds, s, 512); \
    return f32 ? launch_sp(conv3x3_sp_kernel<CTV, false, true, GV>, a, lds, s, 512) : launch_sp(conv3x3_sp_kernel<CTV, false, false, GV>, a, lds, s, 512);         \
  }
  WCT_SP_CASE(1, false)
  WCT_SP_CASE(2, false)
  WCT_SP_CASE(2, true)
#undef WCT_SP_CASE
  return hipErrorInvalidValue;
}

#ifdef WCT_SP_TIMING
extern "C" int wct_debug_sp_timing(unsigned long long* out4) {   // read and reset
  unsigned long long z[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_sp_t), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sp_t), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

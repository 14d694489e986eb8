// bf16 MFMA kernels for the relative-position attention (RelPos1dMultiHeadAttention.forwardQKV, nnet/attentions.py:280-323;
// rel_to_abs :234-278).  One workgroup = 2 waves = 64 queries of one (batch, head); every wave owns 32 queries and sees all keys.
//
// Everything is computed TRANSPOSED so that a query is a lane (column of the 32x32 MFMA result) and keys run over registers:
//     S^T[j][i]   = K_j . Q_i                          mfma(A = K rows, B = Q rows)
//     R^T[e][i]   = E_e . Q_i   (e over the window)    mfma(A = E rows, B = Q rows);   S^T[j][i] += R^T[j + 31 - il][i]   ("skew")
//     softmax over j = over registers (+ one cross-half shuffle), P^T stays in registers as the B operand of
//     O^T[c][i]   = sum_j V^T[c][j] P^T[j][i]          mfma(A = V^T via ds_read_b64_tr_b16, B = P^T registers)
// The reference materialises Q.E^T as (B,H,T,2T-1) and re-indexes it with pad/reshape; here the re-indexing is the skewed read of a
// 2-tile fp32 ring in LDS (row e = j + 31 - il of the wave's window, conflict-free: bank = il).
// LDS images of K, V, E, Q are [row][pitch] bf16 with pitch 128 B (d <= 64) or 256 B (d <= 128) and 16-byte chunks XOR-swizzled by a
// bijection of the row bits chosen so that BOTH the ds_read_b128 operand reads (lanes = rows) and the transposed reads (4 rows x 32 B
// blocks) are bank-conflict free:  pitch 128: chunk ^= ((row>>1)&1)<<2 | (row>>2)&3;   pitch 256: chunk ^= (row&3)<<2 | (row>>2)&3.
#include "attention.h"
#ifndef AVEC_ATTN_ABL
#define AVEC_ATTN_ABL 0
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef short v4s_t __attribute__((ext_vector_type(4)));

static constexpr int NTMAX = 12;     // key tiles of 32 held in registers: kernels are instantiated for NTM = 7 (T <= 224, two workgroups per CU) and 12 (T <= 384)
static constexpr int CTMAX = 3;      // channel tiles of 32 (d <= 96)

template <int PITCH> __device__ __forceinline__ int aswz(int row) {
  return PITCH == 128 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3));
}
__device__ __forceinline__ f32x16 mma(const chunk16& a, const chunk16& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ chunk16 tr8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) v4s_t* lp_t;
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)p1);
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  chunk16 f; f.w[0] = ua.x; f.w[1] = ua.y; f.w[2] = ub.x; f.w[3] = ub.y; return f;
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return f32x2_to_bf16x2(lo, hi); }

// stage `nrows` rows x d channels of a bf16 matrix into the swizzled LDS image (zero outside [0, limit) and beyond d): register path, kept for the ALIAS
// turns of long sequences (the image being replaced is still read by the other wave until the barrier in front of it)
template <int PITCH, int NTHR>
__device__ __forceinline__ void stage_rows(char* dst, const bf16* src, long long ld, int row0, int nrows, int limit, int d, int dpad) {
  const int CH = dpad >> 3;
  for (int idx = threadIdx.x; idx < nrows * CH; idx += NTHR) {
    const int r = idx / CH, c = idx - r * CH; const int gr = row0 + r;
    chunk16 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u;
    if (gr >= 0 && gr < limit) {
      const bf16* p = src + (long long)gr * ld + c * 8;
      if (c * 8 + 8 <= d) v = ldg16(p);      // (gfx950 serves dword loads at any 2-byte address: rows of odd heads -- d = 45, audio stage 0 -- take the 16-byte path too)
      else {
        uint32_t h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (c * 8 + e < d) ? (uint32_t)p[e].v : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) v.w[e] = h[2 * e] | (h[2 * e + 1] << 16);
      }
    }
    *(chunk16*)(dst + r * PITCH + ((c ^ aswz<PITCH>(r)) << 4)) = v;
  }
}

// The same image by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, no VGPR round trip, every instruction of a wave in flight at once).  Round 4: the register
// path above was 8-15 us of a 18-22 us forward launch -- two waves per workgroup, one per SIMD, a load -> wait -> store round trip per 16 bytes.  An instruction
// writes 64 consecutive 16-byte slots = 8 (pitch 128) or 4 (pitch 256) image rows; lane l owns slot l, i.e. (row l / SLOTS, position l % SLOTS), and fetches the
// logical chunk that the swizzle puts there.  The DMA has no fill value, so:
//   * rows outside [0, limit) are CLAMPED to the nearest valid row (finite values): every product that touches them is masked (keys beyond T get -inf / p = 0,
//     E rows outside the window are only read for such pairs, query rows beyond T are never stored);
//   * channels d .. dpad-1 of a row hold whatever follows the head in memory; the caller zeroes them in the Q / dO tiles only (zero_pad_channels) -- K, V and E
//     meet them only in products contracted over the channels with Q or dO, or in output channels that are never stored;
//   * the one 16-byte read that would run past the END of the tensor (last row, last head, d % 8 != 0) is replaced by a zero chunk + element loads.
// Source addresses need 2-byte alignment only (tools/glds_align_probe.hip: any even offset copies right).
__device__ __forceinline__ void attn_glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __attribute__((aligned(64))) unsigned char attn_zero16[64];

// rows [row0, row0 + nrows) of `src` (row stride ld elements, `last_row` = last row of the whole tensor, rowbytes = bytes of a tensor row counted from src's column)
// -> image at LDS byte address lds_img; nrows is a multiple of 8; wave w of NW issues instructions w, w + NW, ...
template <int PITCH, int NW>
__device__ __forceinline__ void stage_rows_dma(unsigned lds_img, char* img, const bf16* src, long long ld, int row0, int nrows, int limit, int d, int dpad,
                                               long long last_row, int rowbytes, int wave, int lane) {
  constexpr int SLOTS = PITCH / 16, RPI = 64 / SLOTS;            // slots per image row, rows per instruction
  constexpr int NV = PITCH == 128 ? 2 : 4;                       // the swizzle of row RPI k + rg depends on k only through k % NV (aswz: bits 1-3 / 0-3 of the row)
  const int CH = dpad >> 3;
  const int rg = lane / SLOTS, pos = lane % SLOTS;
  const int ninstr = nrows / RPI;
  // a row of the image never leaves 2^31 bytes of its tensor (T <= 384 rows of one batch element): 32-bit byte offsets, one multiply-add per instruction; the
  // chunk a lane fetches and whether it is the chunk that would run past the end of the tensor are fixed per k % NV.  The loop is branch-free: it was ~50 VALU
  // instructions + a divergent branch per DMA instruction, 64 DMA instructions per workgroup -- most of the 4 us the staging took.
  const unsigned ldb = (unsigned)ld * 2u;
  const bool can_past = dpad * 2 > rowbytes && last_row <= (long long)limit - 1;      // (uniform) the last row of the TENSOR is among the rows, and its padded width overhangs
  const int lastr = can_past ? (int)last_row : -1;
  // the overhanging chunk of the tensor's LAST row (clamped copies of that row are never used: they get the zero chunk and keep it): its elements are requested
  // here, in front of the DMAs, and stored behind them -- no memory round trip of their own (they cost the last head's workgroups 2.5 us when d % 8 != 0)
  unsigned short pe[8]; int pk = -1, pn = 0; unsigned pc = 0;
  if (can_past) {
    const int rt = lastr - row0;
    if (rt >= 0 && rt < nrows && (rt / RPI) % NW == wave && rt % RPI == rg) {
      int c = pos ^ aswz<PITCH>(rt);
      if (c < CH && c * 16 + 16 > rowbytes) {
        pk = rt / RPI; pc = (unsigned)c * 16u; pn = (rowbytes - c * 16) >> 1; pn = pn < 0 ? 0 : pn;
        const unsigned short* q = (const unsigned short*)((const char*)src + ((unsigned)lastr * ldb + pc));
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = e < pn ? q[e] : (unsigned short)0;
      }
    }
  }
  // this wave's instructions are k = wave + j NW; the v-th of every NV consecutive ones has the swizzle of row RPI (wave + v NW) + rg
  unsigned coff[NV]; bool cpast[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int c = pos ^ aswz<PITCH>(RPI * (wave + v * NW) + rg); c = c < CH ? c : 0;            // unused slots re-read chunk 0
    coff[v] = (unsigned)c * 16u; cpast[v] = c * 16 + 16 > rowbytes;
  }
  for (int k0 = wave; k0 < ninstr; k0 += NW * NV) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int k = k0 + v * NW;
      if (k < ninstr) {
        int gr = row0 + k * RPI + rg; gr = gr < 0 ? 0 : (gr >= limit ? limit - 1 : gr);
        const char* p = (const char*)src + ((unsigned)gr * ldb + coff[v]);
        const bool past = gr == lastr && cpast[v];
        attn_glds16(past ? (const void*)attn_zero16 : (const void*)p, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_img + k * 1024)));
      }
    }
  }
  if (can_past && __builtin_amdgcn_readfirstlane(__any(pk >= 0) ? 1 : 0)) {      // (one wave of the last head's workgroups)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pk >= 0) {
      unsigned short* dst = (unsigned short*)(img + pk * 1024 + lane * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < pn) dst[e] = pe[e];
    }
  }
}
// zero channels d .. dpad-1 of image rows [0, nrows): run by the wave that issued the row's DMA, after its s_waitcnt vmcnt(0)
template <int PITCH, int NW>
__device__ __forceinline__ void zero_pad_channels(char* img, int nrows, int d, int dpad, int wave, int lane) {
  if (d == dpad) return;
  constexpr int SLOTS = PITCH / 16, RPI = 64 / SLOTS;
  const int rg = lane / SLOTS, pos = lane % SLOTS;
  const int ninstr = nrows / RPI;
  for (int k = wave; k < ninstr; k += NW) {
    const int r = k * RPI + rg;
    const int c = pos ^ aswz<PITCH>(r);
    unsigned short* dst = (unsigned short*)(img + k * 1024 + lane * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) if (c * 8 + e >= d && c * 8 + e < dpad) dst[e] = 0;
  }
}

struct MfmaGeom { int Tp, NT, NE, dpad, DKS, CT, alias; size_t offK, offV, offE, offQ, offS, total; };
static MfmaGeom mfma_geom(int T, int d, int pitch, bool bwd) {
  MfmaGeom g; g.Tp = (T + 31) / 32 * 32; g.NT = g.Tp / 32; g.NE = g.Tp + 64; g.dpad = (d + 15) / 16 * 16; g.DKS = g.dpad / 16; g.CT = (d + 31) / 32;
  // long sequences: K and V take turns in ONE image (K for the scores, V for P.V / dP, K again for dQ) -- costs two barriers per turn, fits T = 384 in 160 KB
  for (g.alias = 0; g.alias < 2; ++g.alias) {
    g.offK = 0; g.offV = g.alias ? g.offK : g.offK + (size_t)g.Tp * pitch; g.offE = g.offV + (size_t)g.Tp * pitch; g.offQ = g.offE + (size_t)g.NE * pitch;
    g.offS = g.offQ + (size_t)64 * pitch * (bwd ? 2 : 1);            // bwd: Q and dO tiles
    g.total = g.offS + (size_t)2 * 8192;                             // per wave: 2 x [32][32] fp32 skew ring (also the output staging)
    if (g.total <= 160 * 1024) break;
  }
  if (g.alias > 1) g.alias = 1;
  return g;
}

// S^T tiles of one wave: S[jt][r] = scale * (K_j.Q_i + E_{T-1-i+j}.Q_i) (+ -1e9 on masked keys, -inf beyond T) for key
// j = 32 jt + (r&3) + 8 (r>>2) + 4 hf and query il = lane & 31; returns this lane-half's running maximum.
// DKS: K-steps of 16 channels as a compile-time constant (3 / 4 / 6 = head widths 45 / 64 / 90; 0: G.DKS at run time).  With a run-time trip count the loops
// below stay rolled: two LDS reads -> wait -> one MFMA per iteration, with ONE wave per SIMD and nothing to overlap the round trips with.
template <int PITCH, int NTM, int DKS>
__device__ __forceinline__ float scores(f32x16 (&S)[NTM], const char* Ks, const char* Es, const char* qrow, float* ring, const MfmaGeom& G, float scale,
                                        int w, int il, int hf, int Tn, int klen, bool row_masked) {
  const int dks = DKS ? DKS : G.DKS;
  const int swr = aswz<PITCH>(il);                       // swizzle of operand row (tile bases are multiples of 32: same bits)
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) S[jt][r] = 0.f;
  // R^T tile et -> ring slot et & 1 (rows e_local = 32 et + ..., this wave's window starts 32 (1 - w) rows into the staged E window)
  auto rel_tile = [&](int et) {
    f32x16 R;
#pragma unroll
    for (int r = 0; r < 16; ++r) R[r] = 0.f;
    const char* erow = Es + (32 * et + 32 * (1 - w) + il) * PITCH;
#pragma unroll
    for (int kk = 0; kk < dks; ++kk) {
      const chunk16 fa = *(const chunk16*)(erow + (((2 * kk + hf) ^ swr) << 4));
      const chunk16 fb = *(const chunk16*)(qrow + (((2 * kk + hf) ^ swr) << 4));
      R = mma(fa, fb, R);
    }
    float* slot = ring + (et & 1) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) slot[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + il] = R[r];
  };
  rel_tile(0);
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt) {
    if (jt < G.NT) {
      const char* krow = Ks + (32 * jt + il) * PITCH;
#pragma unroll
      for (int kk = 0; kk < dks; ++kk) {
        const chunk16 fa = *(const chunk16*)(krow + (((2 * kk + hf) ^ swr) << 4));
        const chunk16 fb = *(const chunk16*)(qrow + (((2 * kk + hf) ^ swr) << 4));
        S[jt] = mma(fa, fb, S[jt]);
      }
      rel_tile(jt + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS is in-order per wave: the ring writes above are visible to the reads below
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = (r & 3) + 8 * (r >> 2) + 4 * hf;            // key within the tile
        const int el = 32 * jt + jl + 31 - il;                     // window row; the ring holds rows [32 jt, 32 jt + 64)
        S[jt][r] += ring[(el & 63) * 32 + il];
      }
      asm volatile("" ::: "memory");
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * hf;
        float v = S[jt][r] * scale;
        if (row_masked || j >= klen) v += -1e9f;
        v = j < Tn ? v : -INFINITY;
        S[jt][r] = v; mx = fmaxf(mx, v);
      }
    }
  }
  return mx;
}

// transposed-read operand offsets of a [rows][PITCH] image for channel tile ct: lane (16-lane group g4, t) fetches rows 8 hh + 4 (g4>>1) + (t>>2)
template <int PITCH>
__device__ __forceinline__ void tr_offsets(int (&off)[CTMAX][2], int lane) {
  const int g4 = lane >> 4, t = lane & 15;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int key = 8 * hh + 4 * (g4 >> 1) + (t >> 2);   // + a multiple of 16: the swizzle bits of the row do not change
#pragma unroll
    for (int ct = 0; ct < CTMAX; ++ct) {
      const int cb = 32 * ct + 16 * (g4 & 1);
      off[ct][hh] = key * PITCH + ((((cb >> 3) + ((t & 3) >> 1)) ^ aswz<PITCH>(key)) << 4) + (t & 1) * 8;
    }
  }
}

// stage a [32 queries][d] fp32 register tile held as O^T (channel tiles x 16 registers) through the wave's ring and store its rows coalesced
__device__ __forceinline__ void store_rows_T(const f32x16 (&O)[CTMAX], float mul, float* ring, const MfmaGeom& G, bf16* op, long long ldo, int nrow, int d, int lane) {
  const int hf = lane >> 5, il = lane & 31;
  bf16* ost = (bf16*)ring; const int OP = G.dpad + 4;          // 8 KB ring >= 32 * (96 + 4) * 2 B
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int ct = 0; ct < CTMAX; ++ct) {
    if (ct < G.CT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * ct + 8 * q + 4 * hf;
        uint2 u; u.x = pack2(O[ct][4 * q] * mul, O[ct][4 * q + 1] * mul); u.y = pack2(O[ct][4 * q + 2] * mul, O[ct][4 * q + 3] * mul);
        if (c0 < G.dpad) *(uint2*)(ost + il * OP + c0) = u;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // element pairs as (possibly unaligned) dword stores, the last element of an odd row on its own
  typedef uint32_t __attribute__((aligned(2))) u32_u;
  const int dh = d >> 1;
  for (int idx = lane; idx < nrow * dh; idx += 64) { const int r = idx / dh, c = (idx - r * dh) * 2; *(u32_u*)(op + (long long)r * ldo + c) = *(const uint32_t*)(ost + r * OP + c); }
  if (d & 1) for (int r = lane; r < nrow; r += 64) op[(long long)r * ldo + d - 1] = ost[r * OP + d - 1];
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup = 2 compute waves (32 queries each) + 2 waves that only help with the staging DMAs and leave at the first barrier (not in the ALIAS variants, whose later
// staging turns are written for 128 threads): the staging loop is ~50 VALU instructions of address arithmetic per DMA instruction, 64 of them per workgroup.
template <int PITCH, int NTM, bool ALIAS, int DKS>
__global__ __launch_bounds__(ALIAS ? 128 : 256) void attn_mfma_fwd_kernel(AttnArgs a, MfmaGeom G) {
  constexpr int NWS = ALIAS ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* Ks = sm + G.offK; char* Vs = sm + G.offV; char* Es = sm + G.offE; char* Qs = sm + G.offQ;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5, il = lane & 31;
  float* ring = (float*)(sm + G.offS + w * 8192);
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int Tn = a.T, d = a.d, i0 = blockIdx.x * 64;
  const bf16* qp = (const bf16*)a.q + (long long)b * Tn * a.ld + h * d;
  const bf16* kp = (const bf16*)a.k + (long long)b * Tn * a.ld + h * d;
  const bf16* vp = (const bf16*)a.v + (long long)b * Tn * a.ld + h * d;
  const bf16* ep = (const bf16*)a.e + h * d;
  {
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)sm;
    const int wv = __builtin_amdgcn_readfirstlane(w);
    const long long lastq = (long long)(a.B - 1 - b) * Tn + Tn - 1;      // last row of the q / k / v tensors, counted from this batch element's first row
    const int rb = (a.H - h) * d * 2;                                    // bytes from this head's first column to the end of a tensor row
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offK, Ks, kp, a.ld, 0, G.Tp, Tn, d, G.dpad, lastq, rb, wv, lane);
    if (!ALIAS) stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offV, Vs, vp, a.ld, 0, G.Tp, Tn, d, G.dpad, lastq, rb, wv, lane);
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offQ, Qs, qp, a.ld, i0, 64, Tn, d, G.dpad, lastq, rb, wv, lane);
    // (E last: for the last head of a width that is not a multiple of 8 its staging ends with a wait + patch of the tensor's last row, which would serialise what follows)
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offE, Es, ep, a.lde, (Tn - 1) - (i0 + 63), G.NE, 2 * Tn - 1, d, G.dpad, 2 * Tn - 2, rb, wv, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zero_pad_channels<PITCH, NWS>(Qs, 64, d, G.dpad, wv, lane);
  }
  __syncthreads();
  if (NWS > 2 && w >= 2) return;                          // the staging helpers are done
#if AVEC_ATTN_ABL == 1
  if (sm[tid * 16] == 123) a.lse[0] = 1.f;
  return;
#endif

  const int i = i0 + 32 * w + il;                        // this lane's query
  const char* qrow = Qs + (32 * w + il) * PITCH;
  const int klen = a.lens ? (int)(a.lens[b] / a.len_div) : Tn;
  const bool row_masked = i >= a.q_full;

  f32x16 S[NTM];
  float mx = scores<PITCH, NTM, DKS>(S, Ks, Es, qrow, ring, G, a.scale, w, il, hf, Tn, klen, row_masked);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float lsum = 0.f;
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float p = __expf(S[jt][r] - mx); S[jt][r] = p; lsum += p; }
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  if (hf == 0 && i < Tn) { a.lse[((long long)bh * Tn + i) * 2] = mx; a.lse[((long long)bh * Tn + i) * 2 + 1] = lsum; }
#if AVEC_ATTN_ABL == 2
  return;
#endif

  if (ALIAS) { __syncthreads(); stage_rows<PITCH, 128>(Vs, vp, a.ld, 0, G.Tp, Tn, d, G.dpad); __syncthreads(); }      // V takes K's place
  // O^T = V^T P^T
  f32x16 O[CTMAX];
#pragma unroll
  for (int ct = 0; ct < CTMAX; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
  int offv[CTMAX][2]; tr_offsets<PITCH>(offv, lane);
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        chunk16 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) pb.w[e] = pack2(S[jt][8 * s + 2 * e], S[jt][8 * s + 2 * e + 1]);
        const char* vb = Vs + (32 * jt + 16 * s) * PITCH;
#pragma unroll
        for (int ct = 0; ct < CTMAX; ++ct) if (ct < G.CT) O[ct] = mma(tr8(vb + offv[ct][0], vb + offv[ct][1]), pb, O[ct]);
      }
    }
  }
  bf16* op = (bf16*)a.o + ((long long)b * Tn + i0 + 32 * w) * a.ldo + h * d;
  store_rows_T(O, 1.f / lsum, ring, G, op, a.ldo, min(32, Tn - (i0 + 32 * w)), d, lane);
}

// ------------------------------------------------------------------------------------------------
// backward, row pass: P and dS (stored for the batched dK / dV / dE GEMMs) and dQ
//     dP^T[j][i] = V_j . dO_i ;  dS^T = P^T o (dP^T - delta_i) * scale ;  dQ^T = K^T dS^T + E_win^T unskew(dS^T)
// ------------------------------------------------------------------------------------------------
template <int PITCH, int NTM, bool ALIAS, int DKS>
__global__ __launch_bounds__(ALIAS ? 128 : 256) void attn_mfma_bwd_kernel(AttnArgs a, MfmaGeom G) {
  constexpr int NWS = ALIAS ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* Ks = sm + G.offK; char* Vs = sm + G.offV; char* Es = sm + G.offE; char* Qs = sm + G.offQ; char* Gs = Qs + 64 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hf = lane >> 5, il = lane & 31;
  float* ring = (float*)(sm + G.offS + w * 8192);
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int Tn = a.T, d = a.d, i0 = blockIdx.x * 64;
  const bf16* qp = (const bf16*)a.q + (long long)b * Tn * a.ld + h * d;
  const bf16* kp = (const bf16*)a.k + (long long)b * Tn * a.ld + h * d;
  const bf16* vp = (const bf16*)a.v + (long long)b * Tn * a.ld + h * d;
  const bf16* ep = (const bf16*)a.e + h * d;
  const bf16* gp = (const bf16*)a.dout + (long long)b * Tn * a.ldo + h * d;
  {
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)sm;
    const int wv = __builtin_amdgcn_readfirstlane(w);
    const long long lastq = (long long)(a.B - 1 - b) * Tn + Tn - 1;
    const int rb = (a.H - h) * d * 2;
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offK, Ks, kp, a.ld, 0, G.Tp, Tn, d, G.dpad, lastq, rb, wv, lane);
    if (!ALIAS) stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offV, Vs, vp, a.ld, 0, G.Tp, Tn, d, G.dpad, lastq, rb, wv, lane);
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offQ, Qs, qp, a.ld, i0, 64, Tn, d, G.dpad, lastq, rb, wv, lane);
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offQ + 64 * PITCH, Gs, gp, a.ldo, i0, 64, Tn, d, G.dpad, lastq, rb, wv, lane);
    stage_rows_dma<PITCH, NWS>(lds0 + (unsigned)G.offE, Es, ep, a.lde, (Tn - 1) - (i0 + 63), G.NE, 2 * Tn - 1, d, G.dpad, 2 * Tn - 2, rb, wv, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zero_pad_channels<PITCH, NWS>(Qs, 64, d, G.dpad, wv, lane);
    zero_pad_channels<PITCH, NWS>(Gs, 64, d, G.dpad, wv, lane);
  }
  __syncthreads();
  if (NWS > 2 && w >= 2) return;                          // the staging helpers are done
#if AVEC_ATTN_ABL == 1
  if (sm[tid * 16] == 123) a.lse[0] = 1.f;
  return;
#endif

  const int i = i0 + 32 * w + il;
  const int swr = aswz<PITCH>(il);
  const char* qrow = Qs + (32 * w + il) * PITCH; const char* grow = Gs + (32 * w + il) * PITCH;
  const int klen = a.lens ? (int)(a.lens[b] / a.len_div) : Tn;
  const bool row_masked = i >= a.q_full, iv = i < Tn;
  // delta_i = dO_i . O_i  (each lane-half takes half of the channels)
  float delta = 0.f, m_i = 0.f, il_i = 0.f;
  if (iv) {
    const bf16* orow = (const bf16*)a.o + ((long long)b * Tn + i) * a.ldo + h * d;
    const int c0 = hf * ((d + 1) >> 1), c1 = hf ? d : ((d + 1) >> 1);
    for (int c = c0; c < c1; ++c) {
      const int ch = c >> 3; const bf16* gq = (const bf16*)(grow + ((ch ^ swr) << 4)) + (c & 7);
      delta += bf16_to_f32(gq->v) * bf16_to_f32(orow[c].v);
    }
    m_i = a.lse[((long long)bh * Tn + i) * 2]; il_i = 1.f / a.lse[((long long)bh * Tn + i) * 2 + 1];
  }
  delta += __shfl_xor(delta, 32, 64);

  f32x16 S[NTM];
  scores<PITCH, NTM, DKS>(S, Ks, Es, qrow, ring, G, a.scale, w, il, hf, Tn, klen, row_masked);
#if AVEC_ATTN_ABL == 2
  if (S[0][0] + S[1][3] + delta == 1234.5f) a.lse[0] = 1.f;
  return;
#endif
  if (ALIAS) { __syncthreads(); stage_rows<PITCH, 128>(Vs, vp, a.ld, 0, G.Tp, Tn, d, G.dpad); __syncthreads(); }      // V takes K's place for dP
  bf16* prow = (bf16*)a.pbuf + ((long long)bh * Tn + i) * a.ldt;
  bf16* srow = (bf16*)a.dsbuf + ((long long)bh * Tn + i) * a.ldt;
  bf16* rrow = a.dsrel ? (bf16*)a.dsrel + ((long long)h * a.B * Tn + (long long)b * Tn + i) * a.ldr + (Tn - 1 - i) : nullptr;
  const bool st8 = (a.ldt & 3) == 0 && ((((size_t)a.pbuf) | ((size_t)a.dsbuf)) & 7) == 0;
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt) {
    if (jt < G.NT) {
      f32x16 dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) dP[r] = 0.f;
      const char* vrow = Vs + (32 * jt + il) * PITCH;
#pragma unroll
      for (int kk = 0; kk < (DKS ? DKS : G.DKS); ++kk) {
        const chunk16 fa = *(const chunk16*)(vrow + (((2 * kk + hf) ^ swr) << 4));
        const chunk16 fb = *(const chunk16*)(grow + (((2 * kk + hf) ^ swr) << 4));
        dP = mma(fa, fb, dP);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 32 * jt + 8 * q + 4 * hf;
        float pv[4], dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e; const bool ok = iv && (j0 + e) < Tn;
          const float p = ok ? __expf(S[jt][r] - m_i) * il_i : 0.f;
          pv[e] = p; dv[e] = p * (dP[r] - delta) * a.scale; S[jt][r] = dv[e];
        }
        if (iv && j0 < Tn) {
          if (st8 && j0 + 3 < Tn) {
            uint2 up, ud; up.x = pack2(pv[0], pv[1]); up.y = pack2(pv[2], pv[3]); ud.x = pack2(dv[0], dv[1]); ud.y = pack2(dv[2], dv[3]);
            *(uint2*)(prow + j0) = up; *(uint2*)(srow + j0) = ud;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (j0 + e < Tn) { prow[j0 + e].v = f32_to_bf16(pv[e]); srow[j0 + e].v = f32_to_bf16(dv[e]); }
          }
          if (rrow) {
            // (the skewed row starts at an odd element in every other row: an 8-byte store at a 2-byte-aligned address, like the dword stores of store_rows_T)
            typedef uint2 __attribute__((aligned(2))) u64_u;
            if (j0 + 3 < Tn) { uint2 ud; ud.x = pack2(dv[0], dv[1]); ud.y = pack2(dv[2], dv[3]); *(u64_u*)(rrow + j0) = ud; }
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (j0 + e < Tn) rrow[j0 + e].v = f32_to_bf16(dv[e]);
            }
          }
        }
      }
    }
  }
#if AVEC_ATTN_ABL == 3
  return;
#endif
  if (ALIAS) { __syncthreads(); stage_rows<PITCH, 128>(Ks, kp, a.ld, 0, G.Tp, Tn, d, G.dpad); __syncthreads(); }      // and K comes back for dQ
  // dQ^T = K^T dS^T   (A = K^T via transposed reads, B = dS^T registers)
  f32x16 DQ[CTMAX];
#pragma unroll
  for (int ct = 0; ct < CTMAX; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) DQ[ct][r] = 0.f;
  int offt[CTMAX][2]; tr_offsets<PITCH>(offt, lane);
#pragma unroll
  for (int jt = 0; jt < NTM; ++jt) {
    if (jt < G.NT) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        chunk16 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) pb.w[e] = pack2(S[jt][8 * s + 2 * e], S[jt][8 * s + 2 * e + 1]);
        const char* kb = Ks + (32 * jt + 16 * s) * PITCH;
#pragma unroll
        for (int ct = 0; ct < CTMAX; ++ct) if (ct < G.CT) DQ[ct] = mma(tr8(kb + offt[ct][0], kb + offt[ct][1]), pb, DQ[ct]);
      }
    }
  }
  // relative term: window row e_local of query il pairs with key j = e_local - 31 + il: un-skew dS^T through the ring (tile jt -> slot jt & 1,
  // "tile -1" and tile NT are zero), one window tile at a time, and multiply by E_win^T
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 16; ++r) ring[1024 + ((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + il] = 0.f;     // slot 1 = tile -1
#pragma unroll
  for (int et = 0; et <= NTM; ++et) {
    if (et <= G.NT) {
      float* slot = ring + (et & 1) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = 0.f;
        if (et < NTM) v = (et < G.NT) ? S[et < NTM ? et : 0][r] : 0.f;
        slot[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + il] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float rs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * et + (r & 3) + 8 * (r >> 2) + 4 * hf - 31 + il;      // in [32 et - 31, 32 et + 31]: tiles et - 1 and et
        rs[r] = ring[(j & 63) * 32 + il];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        chunk16 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) pb.w[e] = pack2(rs[8 * s + 2 * e], rs[8 * s + 2 * e + 1]);
        const char* eb = Es + (32 * et + 32 * (1 - w) + 16 * s) * PITCH;
#pragma unroll
        for (int ct = 0; ct < CTMAX; ++ct) if (ct < G.CT) DQ[ct] = mma(tr8(eb + offt[ct][0], eb + offt[ct][1]), pb, DQ[ct]);
      }
    }
  }
  bf16* dqp = (bf16*)a.dq + ((long long)b * Tn + i0 + 32 * w) * a.lddq + h * d;
  store_rows_T(DQ, 1.f, ring, G, dqp, a.lddq, min(32, Tn - (i0 + 32 * w)), d, lane);
}

static const bool g_mfma_off = getenv("AVEC_NO_MFMA_ATTN") != nullptr;

template <typename K> static int mfma_set_lds(K kern, size_t bytes) {
  static const void* done[40]; static size_t done_bytes[40]; static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern && done_bytes[i] >= bytes) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  if (e != hipSuccess) { avec_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
  if (ndone < 40) { done[ndone] = (const void*)kern; done_bytes[ndone] = 160 * 1024; ++ndone; }
  return 0;
}

int attn_mfma_fwd(const AttnArgs& a, hipStream_t st) {
  if (g_mfma_off || a.mask || a.d > 96 || a.T > 32 * NTMAX || a.Tk != a.T) return 1;
  const int pitch = a.d <= 64 ? 128 : 256;
  const MfmaGeom G = mfma_geom(a.T, a.d, pitch, false);
  if (G.total > 160 * 1024) return 1;
  dim3 grid((a.T + 63) / 64, a.B * a.H);
#define AVEC_LAUNCH_ATTN(KERNEL) do { if (int r = mfma_set_lds(KERNEL, G.total)) return r; hipLaunchKernelGGL(KERNEL, grid, dim3(G.alias ? 128 : 256), G.total, st, a, G); } while (0)
#define AVEC_PICK_DKS(NAME, N, A) do { \
    if (pitch == 128) { if (G.DKS == 4) AVEC_LAUNCH_ATTN((NAME<128, N, A, 4>)); else if (G.DKS == 3) AVEC_LAUNCH_ATTN((NAME<128, N, A, 3>)); else AVEC_LAUNCH_ATTN((NAME<128, N, A, 0>)); } \
    else { if (G.DKS == 6) AVEC_LAUNCH_ATTN((NAME<256, N, A, 6>)); else AVEC_LAUNCH_ATTN((NAME<256, N, A, 0>)); } } while (0)
#define AVEC_PICK_ATTN(NAME) do { \
    if (a.T <= 224 && !G.alias) AVEC_PICK_DKS(NAME, 7, false); \
    else if (!G.alias) AVEC_PICK_DKS(NAME, 12, false); \
    else AVEC_PICK_DKS(NAME, 12, true); } while (0)
  AVEC_PICK_ATTN(attn_mfma_fwd_kernel);
  AVEC_LAUNCH_CHECK(); return 0;
}

int attn_mfma_bwd_rows(const AttnArgs& a, hipStream_t st) {
  if (g_mfma_off || a.mask || a.d > 96 || a.T > 32 * NTMAX || a.Tk != a.T || !a.pbuf || !a.dsbuf || !a.dq || !a.dout) return 1;
  const int pitch = a.d <= 64 ? 128 : 256;
  const MfmaGeom G = mfma_geom(a.T, a.d, pitch, true);
  if (G.total > 160 * 1024) return 1;
  dim3 grid((a.T + 63) / 64, a.B * a.H);
  AVEC_PICK_ATTN(attn_mfma_bwd_kernel);
#undef AVEC_PICK_ATTN
#undef AVEC_PICK_DKS
#undef AVEC_LAUNCH_ATTN
  AVEC_LAUNCH_CHECK(); return 0;
}

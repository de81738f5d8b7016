// trust4_amd/csrc/t4_kernels.h -- hand-written gfx950 kernels of the stage-1 hot path.
//
// One 64-lane wavefront owns one read from its 2-bit image in HBM to its scored overlaps:
//   unpack -> roll k-mers (fwd + rc) -> probe the index -> expand postings into 64-bit hit keys in LDS
//   -> wave bitonic sort by (strand, seq, diagonal, seq offset) -> runs of concordant hits
//   -> one lane per run: LIS chaining -> overlap records -> rank sort -> one lane per overlap:
//   anchor walk + banded integer gap DPs -> similarity filters -> (annotate) V/J/C selection.
// Nothing here is GEMM shaped: it is integer scan / gather / sort work, so there is no MFMA.
// Reference semantics followed (file:line in /root/reference):
//   k-mer roll + probe   KmerCode.hpp:94-109, KmerIndex.hpp:29-33,104-116, SeqSet.hpp:1341-1501
//   hit order            SeqSet.hpp:1306-1339 (only the (strand, idx) grouping is observable)
//   runs + chaining      SeqSet.hpp:763-1063, LIS 342-499, VJ rescue 1066-1161
//   overlap scoring      SeqSet.hpp:1508-2124, low complexity 590-617
//   gap DPs              AlignAlgo.hpp:57-216 (posWeight, linear gap), 218-424 (affine)
//   rough annotation     SeqSet.hpp:6016-6066, 6167-6321, contig split 5289-5321
#pragma once
#include <hip/hip_runtime.h>

#include "t4_device.h"

namespace t4k {

// ------------------------------------------------------------------------------------------------
// wave helpers (a workgroup is exactly one wavefront: blockDim.x == 64)
// ------------------------------------------------------------------------------------------------
// A workgroup (NT = blockDim.x threads, a multiple of 64) owns one read; wave-level helpers below are
// combined through a few LDS words into workgroup-level scans / reductions.
// out-of-line device functions keep their registers to themselves (and cost a stack frame in scratch): T4_INLINE_ALL inlines them
#ifdef T4_INLINE_ALL
#define T4_NI
#else
#define T4_NI __attribute__((noinline))
#endif
#ifndef T4_OPT_JOBSORT
#define T4_OPT_JOBSORT 1
#endif
#ifndef T4_OPT_ROWWALK
#define T4_OPT_ROWWALK 1
#endif
#ifndef T4_NI_R3
#define T4_NI_R3 0
#endif
#ifdef __HIPCC__
#define T4_LDS_AS __attribute__((address_space(3)))
#else
#define T4_LDS_AS   /* the emulator build has one address space */
#endif
// (a pointer known to address LDS, cast so: ds_read / ds_write instead of flat instructions, which are slower and tie the LDS counter to
// the vector-memory one)
__device__ __forceinline__ int laneId() { return threadIdx.x & 63; }
__device__ __forceinline__ int tid() { return threadIdx.x; }
__device__ __forceinline__ int nthr() { return blockDim.x; }

__device__ __forceinline__ int waveInclScan(int v) {
  int lane = laneId();
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ int waveSum(int v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ int waveMax(int v) {
  for (int d = 32; d > 0; d >>= 1) { int t = __shfl_xor(v, d); v = t > v ? t : v; }
  return v;
}

// workgroup inclusive scan / sum / any; `red` points at >= 16 ints of LDS. Two barriers each.
__device__ __forceinline__ int blockInclScan(int v, int *red, int &total) {
  int inc = waveInclScan(v);
  int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (laneId() == 63) red[wave] = inc;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < nw; ++w) { int x = red[w]; if (w < wave) off += x; tot += x; }
  __syncthreads();
  total = tot;
  return inc + off;
}
// workgroup inclusive MAX scan; total = the maximum over the whole group. Two barriers.
__device__ __forceinline__ int blockInclMaxScan(int v, int *red, int &total) {
  const int lane = laneId();
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(v, d); if (lane >= d && t > v) v = t; }
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 63) red[wave] = v;
  __syncthreads();
  int pre = -0x7FFFFFFF, tot = -0x7FFFFFFF;
  for (int w = 0; w < nw; ++w) { int x = red[w]; if (w < wave && x > pre) pre = x; if (x > tot) tot = x; }
  __syncthreads();
  total = tot;
  return pre > v ? pre : v;
}
__device__ __forceinline__ int blockSum(int v, int *red) {
  v = waveSum(v);
  int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (laneId() == 0) red[wave] = v;
  __syncthreads();
  int tot = 0;
  for (int w = 0; w < nw; ++w) tot += red[w];
  __syncthreads();
  return tot;
}

__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// KmerIndex::Search (KmerIndex.hpp:104-116): (start, cnt) of the posting list of a valid k-mer.
__device__ __forceinline__ void indexLookup(const T4IndexView &ix, unsigned long long code, int barcode,
                                            unsigned &start, unsigned &cnt) {
  if (ix.direct == 1) {
    uint2 e = ix.table[code];
    start = e.x; cnt = e.y;
    return;
  }
  if (ix.direct >= 2) {   // per-barcode image (2) or live set (3, t4_index_apply_delta): open addressing on the code, an empty
                          // slot holds code ~0; the keys of a live set are never removed (a list may be empty)
    unsigned long long i = mix64(code) & ix.hashMask;
    for (;;) {
      T4HashEntC e = ix.ctab[i];
      if (e.code == code) { start = e.start; cnt = e.cnt; return; }
      if (e.code == ~0ull) { start = 0; cnt = 0; return; }
      i = (i + 1) & ix.hashMask;
    }
  }
  int h = (int)((code + (unsigned long long)(long long)(ix.considerBarcode ? barcode + 1 : 0)) % 1000003ull);
  unsigned long long i = mix64(code * 1000003ull + (unsigned long long)h) & ix.hashMask;
  for (;;) {
    T4HashEnt e = ix.htab[i];
    if (e.h < 0) { start = 0; cnt = 0; return; }
    if (e.code == code && e.h == h) { start = e.start; cnt = e.cnt; return; }
    i = (i + 1) & ix.hashMask;
  }
}

// Neighbour exchange by one lane as a DPP wave shift (a v_mov with a few cycles of latency instead of a ds_bpermute round
// trip through the LDS crossbar): lane i receives lane i-1 (waveUp1; lane 0 keeps its value) or lane i+1 (waveDown1; lane 63
// keeps its value) -- the __shfl_up / __shfl_down(v, 1) semantics. All 64 lanes must be active.
__device__ __forceinline__ int waveUp1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int waveDown1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned waveUp1(unsigned v) { return (unsigned)waveUp1((int)v); }
// the same inside each row of 16 lanes (lane 0 / 15 of a row keeps its value)
__device__ __forceinline__ int rowUp1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int rowDown1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x101 /* row_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned rowUp1(unsigned v) { return (unsigned)rowUp1((int)v); }
__device__ __forceinline__ unsigned rowDown1(unsigned v) { return (unsigned)rowDown1((int)v); }
// row shifts whose edge lane (lane 0 of the row for Up, lane 15 for Down) receives `edge` instead of keeping its own value
__device__ __forceinline__ int rowUp1E(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x111, 0xf, 0xf, false); }
__device__ __forceinline__ int rowDown1E(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x101, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned rowUp1E(unsigned v, unsigned edge) { return (unsigned)rowUp1E((int)v, (int)edge); }
__device__ __forceinline__ unsigned rowDown1E(unsigned v, unsigned edge) { return (unsigned)rowDown1E((int)v, (int)edge); }
__device__ __forceinline__ int rowSum16(int v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; }
__device__ __forceinline__ unsigned waveDown1(unsigned v) { return (unsigned)waveDown1((int)v); }

// isRef of a sequence without touching the sequence table when the set is homogeneous (reference genes only, or novel
// contigs only -- the two kinds of set stage 1 builds)
__device__ __forceinline__ bool seqIsRef(const T4IndexView &ix, int idx) {
  if (ix.hasNovel == 0) return true;    // reference genes only
  if (ix.hasNovel == 2) return false;   // novel contigs only
  return ix.seqs[idx].isRef != 0;
}

__device__ __forceinline__ int nuc2(char c) { // nucToNum[c-'A'] & 3 for the packed alphabet
  return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
}

// ------------------------------------------------------------------------------------------------
// gap DPs. One alignment per lane; score rows live in a lane-interleaved scratch, traceback
// decisions (1 byte / band cell) in a per-lane scratch. Results are the GetAlignStats counts.
// ------------------------------------------------------------------------------------------------
struct DPScratch {
  int *rows;            // [(arr * ROWW + col) * 64 + lane]
  unsigned char *dir;   // this lane's T4_DIR_BYTES
};
#define T4_ROWW (T4_MAXGAP + 2)
#define T4_ROW(arr, col) sc.rows[(((arr) * T4_ROWW) + (col)) * 64 + lane]

// AlignAlgo::GlobalAlignment (AlignAlgo.hpp:218-424) + GetAlignStats. t: consensus chars (global),
// p: read chars (LDS). Returns false when the problem exceeds the scratch (caller flags the read).
__device__ bool dpAffine(const char *t, int lent, const char *p, int lenp, DPScratch sc, int lane,
                         int &nMatch, int &nMis, int &nIndel) {
  nMatch = nMis = nIndel = 0;
  if (lent == 0 || lenp == 0) return true;
  if (lent == 1 && lenp == 1) {
    char a = t[0], b = p[0];
    if (a == b || a == 'N' || b == 'N') nMatch = 1; else nMis = 1;
    return true;
  }
  if (lent > T4_MAXGAP || lenp > T4_MAXGAP) return false;
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  int DW = leftBand + rightBand + 1;
  if (DW > lent) DW = lent;
  if ((long long)(lenp + 1) * DW > T4_DIR_BYTES) return false;
  const int negInf = (lent + 1) * (lenp + 1) * (-4);
  // row 0 (prev = arrays 0..2 : m,e,f)
  int pr = 0, cu = 3;
  int end1 = (1 + rightBand > lent) ? lent : (1 + rightBand);
  T4_ROW(pr + 0, 0) = 0; T4_ROW(pr + 1, 0) = 0; T4_ROW(pr + 2, 0) = 0;
  for (int j = 1; j <= end1; ++j) {
    T4_ROW(pr + 0, j) = -4 - 4 * j;
    T4_ROW(pr + 1, j) = -4 + (lenp + 1) * (-4);  // the reference's stale loop variable (line 271)
    T4_ROW(pr + 2, j) = -4 - j;
  }
  if (end1 < lent) { T4_ROW(pr + 0, end1 + 1) = -4 - 4 * (end1 + 1); T4_ROW(pr + 1, end1 + 1) = -4 + (lenp + 1) * (-4); T4_ROW(pr + 2, end1 + 1) = -4 - (end1 + 1); }
  for (int i = 1; i <= lenp; ++i) {
    int start = (i - leftBand < 1) ? 1 : (i - leftBand);
    int end = (i + rightBand > lent) ? lent : (i + rightBand);
    if (start > 1) { T4_ROW(cu + 0, start - 1) = negInf; T4_ROW(cu + 1, start - 1) = negInf; T4_ROW(cu + 2, start - 1) = negInf; }
    else { T4_ROW(cu + 0, 0) = -4 - 4 * i; T4_ROW(cu + 1, 0) = -4 - i; T4_ROW(cu + 2, 0) = -4 - 4 * i; }
    if (end < lent) { T4_ROW(cu + 0, end + 1) = negInf; T4_ROW(cu + 1, end + 1) = negInf; T4_ROW(cu + 2, end + 1) = negInf; }
    char pc = p[i - 1];
    int mLeft = T4_ROW(cu + 0, start - 1), fLeft = T4_ROW(cu + 2, start - 1);
    int mDiag = T4_ROW(pr + 0, start - 1);
    unsigned char *drow = sc.dir + (size_t)i * DW;
    for (int j = start; j <= end; ++j) {
      int mUp = T4_ROW(pr + 0, j), eUp = T4_ROW(pr + 1, j);
      int e = eUp - 1;
      int eo = mUp - 5;
      if (eo > e) e = eo;
      int f = fLeft - 1;
      int fo = mLeft - 5;
      if (fo > f) f = fo;
      char tc = t[j - 1];
      bool eq = (tc == pc || tc == 'N' || pc == 'N');
      int dsc = mDiag + (eq ? 2 : -2);
      int m = dsc;
      if (e > m) m = e;
      if (f > m) m = f;
      T4_ROW(cu + 0, j) = m; T4_ROW(cu + 1, j) = e; T4_ROW(cu + 2, j) = f;
      unsigned char bits = (unsigned char)((f >= e ? 1 : 0) | (dsc == m ? 2 : 0) | (eq ? 4 : 0) | (eo == e ? 8 : 0) | (fo == f ? 16 : 0));
      drow[j - start] = bits;
      mDiag = mUp; mLeft = m; fLeft = f;
    }
    int tsw = pr; pr = cu; cu = tsw;
  }
  // traceback (border cells are evaluated from their closed forms)
  int tagi = lenp, tagj = lent, mat = 0;
  while (tagi > 0 || tagj > 0) {
    bool interior = tagi > 0 && tagj > 0;
    unsigned char bits = 0;
    if (interior) {
      int start = (tagi - leftBand < 1) ? 1 : (tagi - leftBand);
      bits = sc.dir[(size_t)tagi * DW + (tagj - start)];
    }
    if (mat == 0) {
      int a;  // 2 insert, 3 delete, 0 match, 1 mismatch
      if (interior) {
        a = (bits & 1) ? 3 : 2;
        if (bits & 2) a = (bits & 4) ? 0 : 1;
      } else if (tagi == 0) {
        // e[0][j] = -4-4(lenp+1), f[0][j] = -4-j
        a = (-4 - tagj >= -4 + (lenp + 1) * (-4)) ? 3 : 2;
      } else {
        // e[i][0] = -4-i, f[i][0] = -4-4i  => f >= e only for i <= 0
        a = 2;
      }
      if (a == 0) { ++nMatch; --tagi; --tagj; }
      else if (a == 1) { ++nMis; --tagi; --tagj; }
      else if (a == 2) mat = 1;
      else mat = 2;
    } else if (mat == 1) {
      ++nIndel;
      if (tagi > 0) {
        bool open;
        if (interior) open = (bits & 8) != 0;
        else { // column 0: m[i-1][0] - 5 == e[i][0] = -4 - i
          int mUp = (tagi - 1 == 0) ? 0 : (-4 - 4 * (tagi - 1));
          open = (mUp - 5 == -4 - tagi);
        }
        --tagi;
        mat = open ? 0 : 1;
      } else mat = 2;
    } else {
      ++nIndel;
      if (tagj > 0) {
        bool open;
        if (interior) open = (bits & 16) != 0;
        else { // row 0: m[0][j-1] - 5 == f[0][j] = -4 - j
          int mL = (tagj - 1 == 0) ? 0 : (-4 - 4 * (tagj - 1));
          open = (mL - 5 == -4 - tagj);
        }
        --tagj;
        mat = open ? 0 : 2;
      } else mat = 1;
    }
  }
  return true;
}

// AlignAlgo::IsBaseEqual (AlignAlgo.hpp:49-55)
// c is a character of an unpacked read (A, C, G, T or N): bits 1-2 of the four letters' codes tell them apart (A 00, C 01,
// T 10, G 11), so the predicate bit is picked without a branch -- this test sits in the innermost loop of every alignment
// against a novel contig, one character per lane, and a chain of comparisons there compiles to divergent branches.
__device__ __forceinline__ bool baseEqualW(T4PW w, char c) {
  return (((unsigned)w >> 4) | (unsigned)(c == 'N') | ((unsigned)w >> ((0xB4u >> (c & 6)) & 3u))) & 1u;
}

// AlignAlgo::GlobalAlignment_PosWeight (AlignAlgo.hpp:57-216). When `align` is non-null the edit
// string (terminated by -1) is written there (lane-private global memory, needs lent+lenp+2 bytes).
__device__ bool dpPosWeight(const T4PW *w, int lent, const char *p, int lenp, DPScratch sc, int lane,
                            int &nMatch, int &nMis, int &nIndel, signed char *align) {
  nMatch = nMis = nIndel = 0;
  if (lent == 0 || lenp == 0) { if (align) align[0] = -1; return true; }
  if (lent == 1 && lenp == 1) {
    bool eq = baseEqualW(w[0], p[0]);
    if (eq) nMatch = 1; else nMis = 1;
    if (align) { align[0] = eq ? 0 : 1; align[1] = -1; }
    return true;
  }
  if (lent == lenp) {
    int score = 0, mm = 0;
    for (int i = 0; i < lent; ++i) {
      bool eq = baseEqualW(w[i], p[i]);
      score += eq ? 2 : -2;
      mm += eq ? 0 : 1;
      if (align) align[i] = eq ? 0 : 1;
    }
    if (score >= lent * 2 - 8) {
      if (align) align[lent] = -1;
      nMatch = lent - mm; nMis = mm;
      return true;
    }
  }
  if (lent > T4_MAXGAP || lenp > T4_MAXGAP) return false;
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  int DW = leftBand + rightBand + 1;
  if (DW > lent) DW = lent;
  if ((long long)(lenp + 1) * DW > T4_DIR_BYTES) return false;
  const int negInf = (lent + 1) * (lenp + 1) * (-4);
  int pr = 0, cu = 3;
  int end1 = (1 + rightBand > lent) ? lent : (1 + rightBand);
  T4_ROW(pr, 0) = 0;
  for (int j = 1; j <= end1; ++j) T4_ROW(pr, j) = -4 - 4 * j;
  if (end1 < lent) T4_ROW(pr, end1 + 1) = -4 - 4 * (end1 + 1);
  for (int i = 1; i <= lenp; ++i) {
    int start = (i - leftBand < 1) ? 1 : (i - leftBand);
    int end = (i + rightBand > lent) ? lent : (i + rightBand);
    if (start > 1) T4_ROW(cu, start - 1) = negInf; else T4_ROW(cu, 0) = -4 - 4 * i;
    if (end < lent) T4_ROW(cu, end + 1) = negInf;
    char pc = p[i - 1];
    int mLeft = T4_ROW(cu, start - 1), mDiag = T4_ROW(pr, start - 1);
    unsigned char *drow = sc.dir + (size_t)i * DW;
    for (int j = start; j <= end; ++j) {
      int mUp = T4_ROW(pr, j);
      bool eq = baseEqualW(w[j - 1], pc);
      int dsc = mDiag + (eq ? 2 : -2);
      int m = dsc;
      if (mLeft - 4 > m) m = mLeft - 4;
      if (mUp - 4 > m) m = mUp - 4;
      T4_ROW(cu, j) = m;
      drow[j - start] = (unsigned char)((mLeft - 4 == m ? 1 : 0) | (mUp - 4 == m ? 2 : 0) | (dsc == m ? 4 : 0) | (eq ? 8 : 0));
      mDiag = mUp; mLeft = m;
    }
    int tsw = pr; pr = cu; cu = tsw;
  }
  int tagi = lenp, tagj = lent, tag = 0;
  while (tagi > 0 || tagj > 0) {
    int a = 0;
    if (tagi > 0 && tagj > 0) {
      int start = (tagi - leftBand < 1) ? 1 : (tagi - leftBand);
      unsigned char bits = sc.dir[(size_t)tagi * DW + (tagj - start)];
      if (bits & 1) a = 3;
      if (bits & 2) a = 2;
      if (bits & 4) a = (bits & 8) ? 0 : 1;
    } else if (tagi == 0) { // row 0: m[0][j] = -4-4j, m[0][0] = 0
      int mL = (tagj - 1 == 0) ? 0 : (-4 - 4 * (tagj - 1));
      if (mL - 4 == -4 - 4 * tagj) a = 3;
    } else {                // column 0
      int mU = (tagi - 1 == 0) ? 0 : (-4 - 4 * (tagi - 1));
      if (mU - 4 == -4 - 4 * tagi) a = 2;
    }
    if (align) align[tag] = (signed char)a;
    ++tag;
    if (a == 0) ++nMatch; else if (a == 1) ++nMis; else ++nIndel;
    if (a == 3) --tagj; else if (a == 2) --tagi; else { --tagi; --tagj; }
  }
  if (align) {
    align[tag] = -1;
    for (int i = 0, j = tag - 1; i < j; ++i, --j) { signed char x = align[i]; align[i] = align[j]; align[j] = x; }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// Forward-only gap DPs for overlap scoring. GetOverlapsFromRead only consumes GetAlignStats of the
// alignment (matches, mismatches, indels), and the reference's traceback is a deterministic function
// of the DP cell it stands on, so the statistics of "the traceback path from this cell" obey the same
// recurrences as the scores and can be carried forward: no direction matrix, no traceback, and a
// single band-wide row per alignment (T4_DPW columns, updated in place) that lives in LDS.
// Row storage: slotBase[(arr * T4_DPW + d) * slotStride], arr in {M, E, C0, C1} (affine) / {M, C}.
// Counts are packed match | mismatch << 10 | indel << 20 (each < 1024 because gaps are <= T4_MAXGAP).
// Returns false when the band is wider than T4_DPW (caller falls back to the scratch version).
// ------------------------------------------------------------------------------------------------
#define T4_DPW 32
#define DP_PENDING 0xFFFFFFFFu
#define DP_FAIL 0xFFFFFFFEu
#define CNT_MATCH 1u
#define CNT_MIS (1u << 10)
#define CNT_INDEL (1u << 20)
#define T4_SLOT(arr, d) slot[((arr) * T4_DPW + (d)) * slotStride]

__device__ bool dpAffineFwd(const char *t, int lent, const char *p, int lenp, int *slot, int slotStride,
                            int &nMatch, int &nMis, int &nIndel) {
  nMatch = nMis = nIndel = 0;
  if (lent == 0 || lenp == 0) return true;
  if (lent == 1 && lenp == 1) {
    char a = t[0], b = p[0];
    if (a == b || a == 'N' || b == 'N') nMatch = 1; else nMis = 1;
    return true;
  }
  if (lent == lenp) {
    // Substitution-only shortcut. With equal lengths every alignment that leaves the main diagonal opens at
    // least one insertion and one deletion (2 x -5) and so scores <= 2*len - 12 on every prefix, while the
    // diagonal prefix scores 2*i - 4*mm_i; for mm <= 3 the diagonal is therefore optimal on every prefix and
    // the reference's traceback (which tests the diagonal move first, AlignAlgo.hpp:338-349) walks it.
    int mm = 0;
    for (int i = 0; i < lent; ++i) { char a = t[i], b = p[i]; mm += (a == b || a == 'N' || b == 'N') ? 0 : 1; }
    if (mm <= 3) { nMatch = lent - mm; nMis = mm; return true; }
  }
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  const int W = leftBand + rightBand + 1;
  if (W > T4_DPW || lent > T4_MAXGAP || lenp > T4_MAXGAP) return false;
  const int negInf = (lent + 1) * (lenp + 1) * (-4);
  const int e0 = -4 + (lenp + 1) * (-4);   // e[0][j], j >= 1 (the reference's stale loop variable)
  // row 0 in the band coordinates of row 0: d' <-> j' = d' - leftBand
  for (int d = 0; d < W; ++d) {
    int j = d - leftBand;
    if (j < 0 || j > lent) continue;
    if (j == 0) { T4_SLOT(0, d) = 0; T4_SLOT(1, d) = 0; T4_SLOT(2, d) = 0; T4_SLOT(3, d) = 0; }
    else {
      T4_SLOT(0, d) = -4 - 4 * j; T4_SLOT(1, d) = e0;
      T4_SLOT(2, d) = (int)(CNT_INDEL * (unsigned)(j + (j > 4 * (lenp + 1) ? 1 : 0)));  // state 0 at (0, j)
      T4_SLOT(3, d) = (int)(CNT_INDEL * (unsigned)(1 + j));                               // state 1 at (0, j)
    }
  }
  for (int i = 1; i <= lenp; ++i) {
    const int start = (i - leftBand < 1) ? 1 : (i - leftBand);
    const int end = (i + rightBand > lent) ? lent : (i + rightBand);
    const char pc = p[i - 1];
    int mLeft, fLeft; unsigned c0Left, c2Left;
    if (start > 1) { mLeft = negInf; fLeft = negInf; c0Left = 0; c2Left = 0; }
    else { mLeft = -4 - 4 * i; fLeft = -4 - 4 * i; c0Left = CNT_INDEL * (unsigned)i; c2Left = CNT_INDEL * (unsigned)(1 + i); }
    for (int j = start; j <= end; ++j) {
      const int d = j - i + leftBand;          // band column in row i; row i-1 keeps (i-1, j-1) at d, (i-1, j) at d+1
      int mDiag = T4_SLOT(0, d); unsigned c0Diag = (unsigned)T4_SLOT(2, d);
      int mUp, eUp; unsigned c0Up, c1Up;
      if (d + 1 < W) { mUp = T4_SLOT(0, d + 1); eUp = T4_SLOT(1, d + 1); c0Up = (unsigned)T4_SLOT(2, d + 1); c1Up = (unsigned)T4_SLOT(3, d + 1); }
      else { mUp = negInf; eUp = negInf; c0Up = 0; c1Up = 0; }
      int e = eUp - 1, eo = mUp - 5;
      if (eo > e) e = eo;
      int f = fLeft - 1, fo = mLeft - 5;
      if (fo > f) f = fo;
      const char tc = t[j - 1];
      const bool eq = (tc == pc || tc == 'N' || pc == 'N');
      const int dsc = mDiag + (eq ? 2 : -2);
      int m = dsc;
      if (e > m) m = e;
      if (f > m) m = f;
      const unsigned c1 = CNT_INDEL + ((eo == e) ? c0Up : c1Up);
      const unsigned c2 = CNT_INDEL + ((fo == f) ? c0Left : c2Left);
      unsigned c0;
      if (dsc == m) c0 = c0Diag + (eq ? CNT_MATCH : CNT_MIS);
      else c0 = (f >= e) ? c2 : c1;
      T4_SLOT(0, d) = m; T4_SLOT(1, d) = e; T4_SLOT(2, d) = (int)c0; T4_SLOT(3, d) = (int)c1;
      mLeft = m; fLeft = f; c0Left = c0; c2Left = c2;
    }
    // column 0 of this row, read as (i, 0) by cell (i + 1, 1)
    const int d0 = leftBand - i;
    if (d0 >= 0) { T4_SLOT(0, d0) = -4 - 4 * i; T4_SLOT(1, d0) = -4 - i; T4_SLOT(2, d0) = (int)(CNT_INDEL * (unsigned)i); T4_SLOT(3, d0) = (int)(CNT_INDEL * (unsigned)i); }
    // sentinel right of the band, read as (i, end + 1) by cell (i + 1, end + 1): its band column in row i is W
    // (handled by the d + 1 < W test); a sentinel left of the band is the `start > 1` case above.
  }
  const unsigned c = (unsigned)T4_SLOT(2, lent - lenp + leftBand);
  nMatch = (int)(c & 1023u); nMis = (int)((c >> 10) & 1023u); nIndel = (int)(c >> 20);
  return true;
}

__device__ bool dpPosWeightFwd(const T4PW *w, int lent, const char *p, int lenp, int *slot, int slotStride,
                               int &nMatch, int &nMis, int &nIndel) {
  nMatch = nMis = nIndel = 0;
  if (lent == 0 || lenp == 0) return true;
  if (lent == 1 && lenp == 1) {
    if (baseEqualW(w[0], p[0])) nMatch = 1; else nMis = 1;
    return true;
  }
  if (lent == lenp) {
    int mm = 0;
    for (int i = 0; i < lent; ++i) mm += baseEqualW(w[i], p[i]) ? 0 : 1;
    if ((lent - mm) * 2 - mm * 2 >= lent * 2 - 8) { nMatch = lent - mm; nMis = mm; return true; }
  }
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  const int W = leftBand + rightBand + 1;
  if (W > T4_DPW || lent > T4_MAXGAP || lenp > T4_MAXGAP) return false;
  const int negInf = (lent + 1) * (lenp + 1) * (-4);
  for (int d = 0; d < W; ++d) {
    int j = d - leftBand;
    if (j < 0 || j > lent) continue;
    if (j == 0) { T4_SLOT(0, d) = 0; T4_SLOT(1, d) = 0; }
    else { T4_SLOT(0, d) = -4 - 4 * j; T4_SLOT(1, d) = (int)(CNT_MATCH + CNT_INDEL * (unsigned)(j - 1)); }
  }
  for (int i = 1; i <= lenp; ++i) {
    const int start = (i - leftBand < 1) ? 1 : (i - leftBand);
    const int end = (i + rightBand > lent) ? lent : (i + rightBand);
    const char pc = p[i - 1];
    int mLeft; unsigned cLeft;
    if (start > 1) { mLeft = negInf; cLeft = 0; }
    else { mLeft = -4 - 4 * i; cLeft = CNT_MATCH + CNT_INDEL * (unsigned)(i - 1); }
    for (int j = start; j <= end; ++j) {
      const int d = j - i + leftBand;
      int mDiag = T4_SLOT(0, d); unsigned cDiag = (unsigned)T4_SLOT(1, d);
      int mUp; unsigned cUp;
      if (d + 1 < W) { mUp = T4_SLOT(0, d + 1); cUp = (unsigned)T4_SLOT(1, d + 1); } else { mUp = negInf; cUp = 0; }
      const bool eq = baseEqualW(w[j - 1], pc);
      const int dsc = mDiag + (eq ? 2 : -2);
      int m = dsc;
      if (mLeft - 4 > m) m = mLeft - 4;
      if (mUp - 4 > m) m = mUp - 4;
      unsigned c;
      if (dsc == m) c = cDiag + (eq ? CNT_MATCH : CNT_MIS);
      else if (mUp - 4 == m) c = cUp + CNT_INDEL;
      else c = cLeft + CNT_INDEL;
      T4_SLOT(0, d) = m; T4_SLOT(1, d) = (int)c;
      mLeft = m; cLeft = c;
    }
    const int d0 = leftBand - i;
    if (d0 >= 0) { T4_SLOT(0, d0) = -4 - 4 * i; T4_SLOT(1, d0) = (int)(CNT_MATCH + CNT_INDEL * (unsigned)(i - 1)); }
  }
  const unsigned c = (unsigned)T4_SLOT(1, lent - lenp + leftBand);
  nMatch = (int)(c & 1023u); nMis = (int)((c >> 10) & 1023u); nIndel = (int)(c >> 20);
  return true;
}

// ------------------------------------------------------------------------------------------------
// LIS of one run (SeqSet::LongestIncreasingSubsequence, SeqSet.hpp:342-499), executed by one lane.
// hits: (b << 12 | a) sorted by (b, a). Returns the chain length; chain written to lisOut.
// ------------------------------------------------------------------------------------------------
#define PA(v) ((int)((v) & 0xFFFu))
#define PB(v) ((int)((v) >> 12))
__device__ __forceinline__ double dabs(double x) { return x < 0 ? -x : x; }

__device__ int lisLane(const unsigned *hits, int size, unsigned *LIS, unsigned short *top, unsigned short *link) {
  double avgDiff = 0;
  for (int i = 1; i < size; ++i) avgDiff += (PA(hits[i]) - PB(hits[i]));
  avgDiff /= size;
  int ret = 1;
  top[0] = 0; link[0] = 0xFFFF;
  for (int i = 1; i < size; ++i) {
    int ai = PA(hits[i]);
    int tag;
    if (PA(hits[top[ret - 1]]) <= ai) tag = ret - 1;
    else {
      int l = 0, r = ret - 1; tag = -2;
      while (l <= r) {
        int m = (l + r) / 2;
        int am = PA(hits[top[m]]);
        if (ai == am) { tag = m; break; }
        else if (ai < am) r = m - 1; else l = m + 1;
      }
      if (tag == -2) tag = l - 1;
    }
    if (tag == -1) { top[0] = (unsigned short)i; link[i] = 0xFFFF; }
    else {
      int at = PA(hits[top[tag]]);
      if (ai > at) {
        if (tag == ret - 1) { top[ret] = (unsigned short)i; ++ret; link[i] = top[tag]; }
        else if (ai < PA(hits[top[tag + 1]])) { top[tag + 1] = (unsigned short)i; link[i] = top[tag]; }
      } else if (ai == at) {
        unsigned ht = hits[top[tag]];
        if (dabs(ai - PB(hits[i]) - avgDiff) < dabs(PA(ht) - PB(ht) - avgDiff)) {
          top[tag] = (unsigned short)i;
          link[i] = tag > 0 ? top[tag - 1] : (unsigned short)0xFFFF;
        }
      }
    }
  }
  int k = top[ret - 1];
  for (int i = ret - 1; i >= 0; --i) { LIS[i] = hits[k]; k = link[k]; }
  // collapse equal-b runs (keep the least divergent, first on ties)
  k = 0;
  for (int i = 0; i < ret;) {
    int j;
    for (j = i + 1; j < ret; ++j) if (PB(LIS[i]) != PB(LIS[j])) break;
    if (j == i + 1) LIS[k] = LIS[i];
    else {
      int mintag = i; double minDiff = dabs(PA(LIS[i]) - PB(LIS[i]) - avgDiff);
      for (int l = i + 1; l < j; ++l) {
        double d = dabs(PA(LIS[l]) - PB(LIS[l]) - avgDiff);
        if (d < minDiff) { minDiff = d; mintag = l; }
      }
      LIS[k] = LIS[mintag];
    }
    i = j; ++k;
  }
  ret = k;
  // replacement sweep
  int i = 0, j = 0;
  while (i < ret && j < size) {
    unsigned hj = hits[j], li = LIS[i];
    if (PB(hj) < PB(li)) ++j;
    else if (i + 1 < ret && PB(LIS[i + 1]) <= PB(hj)) ++i;
    else if (li == hj) ++j;
    else {
      if (PA(li) <= PA(hj) && (i == ret - 1 || PA(hj) < PA(LIS[i + 1])) &&
          dabs(PA(hj) - PB(hj) - avgDiff) < dabs(PA(li) - PB(li) - avgDiff))
        LIS[i] = hj;
      ++j;
    }
  }
  return ret;
}

// GetTotalHitLengthOnRead / OnSeq (SeqSet.hpp:3330-3367) over a chain
__device__ __forceinline__ int totalHitLen(const unsigned *c, int n, int K, bool onSeq) {
  int ret = 0;
  for (int i = 0; i < n;) {
    int j;
    for (j = i + 1; j < n; ++j) {
      int cur = onSeq ? PB(c[j]) : PA(c[j]), prv = onSeq ? PB(c[j - 1]) : PA(c[j - 1]);
      if (cur > prv + K - 1) break;
    }
    ret += (onSeq ? PB(c[j - 1]) - PB(c[i]) : PA(c[j - 1]) - PA(c[i])) + K;
    i = j;
  }
  return ret;
}

// overlap record while a read is being processed (10 ints)
struct OvRec {
  int seqIdx, rs, re, ss, se;
  int matchCnt;      // final matchCnt
  int indelCnt;
  int chainPos;      // run start index (chain = u32 view of the key array at 2*chainPos)
  int chainLen;
  int flags;         // bit0: strand == 1, bit1: similarity forced to 0, bit2: isRef
};
#define CAND_START_BITS 18
#define CAND_START(c) ((int)((c) & ((1u << CAND_START_BITS) - 1u)))
#define CAND_LEN(c) ((int)((c) >> CAND_START_BITS))
#define OV_PLUS 1
#define OV_SIMZERO 2
#define OV_ISREF 4
#define OV_CUT 8           // cut by the pre-filters of SeqSet.hpp:1705-1794 (mode 4: its scored fields are stashed in chainPos, see ovStashScored)
#define OV_GENETYPE(f) (((f) >> 8) & 255)
#define OV_NAME0(f) ((char)(((f) >> 16) & 255))
#define OV_NAME2(f) ((char)(((unsigned)(f)) >> 24))

__device__ __forceinline__ double ovSim(const OvRec &o) {
  if (o.flags & OV_SIMZERO) return 0.0;
  return (double)o.matchCnt / (double)(o.se - o.ss + 1 + o.re - o.rs + 1);
}
// _overlap::operator< (SeqSet.hpp:104-128) as a three-way comparison (< 0: a sorts first); `scored` = similarity is
// meaningful. The similarity test only runs between records of EQUAL matchCnt m, where the doubles m / den_a and m / den_b
// (den < 2^12) differ exactly when the denominators do and order inversely to them; a zero similarity (m == 0 or the
// SIMZERO flag) is the largest denominator. No floating point, identical order.
__device__ __forceinline__ unsigned ovSimDen(const OvRec &o) {
  return ((o.flags & OV_SIMZERO) || o.matchCnt == 0) ? 0xFFFFFFFFu : (unsigned)(o.se - o.ss + 1 + o.re - o.rs + 1);
}
__device__ __forceinline__ int ovCmp(const OvRec &a, const OvRec &b, bool scored) {
  if (a.matchCnt != b.matchCnt) return a.matchCnt > b.matchCnt ? -1 : 1;
  if (scored) {
    const unsigned da = ovSimDen(a), db = ovSimDen(b);
    if (da != db) return da < db ? -1 : 1;
  }
  if (a.re - a.rs != b.re - b.rs) return a.re - a.rs > b.re - b.rs ? -1 : 1;
  if (a.seqIdx != b.seqIdx) return a.seqIdx < b.seqIdx ? -1 : 1;
  const int sta = (a.flags & OV_PLUS) ? 1 : -1, stb = (b.flags & OV_PLUS) ? 1 : -1;
  if (sta != stb) return sta < stb ? -1 : 1;
  if (a.rs != b.rs) return a.rs < b.rs ? -1 : 1;
  if (a.re != b.re) return a.re < b.re ? -1 : 1;
  if (a.ss != b.ss) return a.ss < b.ss ? -1 : 1;
  if (a.se != b.se) return a.se < b.se ? -1 : 1;
  return 0;
}
__device__ __forceinline__ bool ovLess(const OvRec &a, const OvRec &b, bool scored) { return ovCmp(a, b, scored) < 0; }

// A pre-filter cut leaves the overlap as the reference's `continue` does (matchCnt of GetOverlapsFromHits, similarity 0). The scored
// fields it had are kept in chainPos (dead once the overlaps are scored): the candidate store of the ordered builder hands them to
// the host, whose replay of the scan may find the overlap uncut after another contig's candidates changed (T4QueryArgs::candOut).
// (chainPos of a scored overlap of a contig: bits 21-30 the size of the run it was chained from -- the number novelMinHitRequired is
// compared with, SeqSet.hpp:923-925 --, and once the pre-filters cut it, bits 0-9 / 10-19 / 20 its scored matchCnt / indelCnt / zero flag)
#define OV_RUN_SHIFT 21
// (a scored matchCnt is at most the two spans of the overlap, 2 * T4_MAXL: ten bits hold it only while reads stay this short -- and
// T4Cand carries rs / re / m0 as 16-bit fields)
static_assert(2 * T4_MAXL < 1024, "ovCutKeepScored packs the scored matchCnt into ten bits: widen the packing before T4_MAXL grows");
__device__ __forceinline__ void ovKeepRunSize(OvRec &o, int runSize) { o.chainPos = (runSize > 1023 ? 1023 : runSize) << OV_RUN_SHIFT; }
__device__ __forceinline__ void ovCutKeepScored(OvRec &o) {
  const int ind = o.indelCnt > 1023 ? 1023 : o.indelCnt;
  o.chainPos = (o.chainPos & (0x3FF << OV_RUN_SHIFT)) | (o.matchCnt & 0x3FF) | (ind << 10) | ((o.flags & OV_SIMZERO) ? (1 << 20) : 0);
  o.matchCnt = o.chainLen; o.indelCnt = 0; o.flags |= OV_SIMZERO | OV_CUT;
}
__device__ __forceinline__ T4Cand ovToCand(const OvRec &o) {   // o.chainLen holds the matchCnt of GetOverlapsFromHits (scoreOverlaps leaves it there)
  T4Cand c;
  c.seqIdx = o.seqIdx; c.ss = o.ss; c.se = o.se; c.rs = (short)o.rs; c.re = (short)o.re; c.m0 = (short)o.chainLen;
  const bool cut = (o.flags & OV_CUT) != 0;
  c.matchCnt = (short)(cut ? (o.chainPos & 0x3FF) : o.matchCnt);
  c.indelCnt = (short)(cut ? ((o.chainPos >> 10) & 0x3FF) : (o.indelCnt > 32767 ? 32767 : o.indelCnt));
  const bool simzero = cut ? ((o.chainPos >> 20) & 1) != 0 : (o.flags & OV_SIMZERO) != 0;
  const int runSize = (o.chainPos >> OV_RUN_SHIFT) & 0x3FF;
  c.flags = (unsigned short)(((o.flags & OV_PLUS) ? T4_CAND_PLUS : 0) | (simzero ? T4_CAND_SIMZERO : 0) | (cut ? T4_CAND_CUT : 0) | (runSize << T4_CAND_RUN_SHIFT));
  return c;
}

// ------------------------------------------------------------------------------------------------
// per-wave working set. CAP = hit capacity, MAXOV = overlap capacity. LDS tiers use static
// __shared__ arrays; the last tier (CAP == 0) works out of per-block global scratch.
// ------------------------------------------------------------------------------------------------
// optional per-phase cycle accounting (build with -DT4_PHASE_TIMING; read back through T4Work.phase)
#ifdef T4_PHASE_TIMING
#define T4_NPHASE 64   // ids 0-31: reads in the LDS tiers; the same + 32: reads in global scratch (WaveState::phaseBase)
__device__ unsigned long long g_phaseCycles[T4_NPHASE];
__device__ unsigned long long g_dbgCount[8];   // 0 jobs, 1 pending (banded) jobs, 2 wave-DP steps, 3 scratch fallbacks, 4 fallback cells, 5 fallback cycles, 6 overhang DPs (four per wavefront), 7 of which leave the diagonal
#define DBG_ADD(i, v) do { atomicAdd(&g_dbgCount[i], (unsigned long long)(v)); } while (0)
__device__ unsigned long long g_phaseByOverlaps[4 * T4_NPHASE];   // the same, per read of an AddRead query, by its number of overlaps: < 5, < 20, <= 64, more
#define PHASE_MARK(ws, id)                                                                  \
  do { if (threadIdx.x == 0) { long long now_ = clock64(); atomicAdd(&g_phaseCycles[(ws)->curPhase], (unsigned long long)(now_ - (ws)->phaseT0)); (ws)->phaseLocal[(ws)->curPhase & (T4_NPHASE - 1)] += (unsigned)(now_ - (ws)->phaseT0); (ws)->phaseT0 = now_; (ws)->curPhase = (id) + (ws)->phaseBase; } } while (0)
#define PHASE_MARK_RED(red, id) PHASE_MARK((WaveState *)((char *)(red) - offsetof(WaveState, red)), id)
#else
#define PHASE_MARK(ws, id) do { } while (0)
#define PHASE_MARK_RED(red, id) do { } while (0)
#define DBG_ADD(i, v) do { } while (0)
#endif

struct WaveMem {
  unsigned long long *keys;  // [cap]   hit keys; later per-run {chain u32[n], top u16[n], link u16[n]}
  unsigned *pairs;           // [cap]   (b << 12 | a) of candidate runs; first: posPref
  OvRec *ov;                 // [maxOv] overlaps of the current pass; first: posStart
  OvRec *fin;                // [maxFin] accumulated final overlaps (annotate) / result (overlaps)
  unsigned short *ord;       // [maxOv] sort order
  unsigned *cand;            // [cap / 3 + 1] candidate runs: start | len << 18 (hits of one pass: at most 2^18, the global-scratch tier's capacity)
  char *seg, *rc;            // [T4_MAXL + 8] current segment, forward and reverse complement
  int cap, maxOv, maxFin, candCap;
  unsigned long long *ldsSort;     // global-scratch mode: LDS staging buffer of the hit sort (null: none)
  int ldsSortCap;
  int hitLimit;                    // hits of one pass the arrays take: cap, or less under the testing aid T4Work::capLimit
  int ldsArrays;                   // keys / pairs / ov live in LDS (every tier but the global-scratch one)
  unsigned char *dirBuf;           // buffer of the overhang alignments (ExtendOverlap): LDS in every tier (the key array or the
  int dirBytes;                    // sort's staging block, both dead by then)
  const unsigned *pkRow, *nmRow;   // the read's packed words (global), set by loadSegment
  int segAbs, segLen;              // position of the current segment inside the read
};

struct WaveState { // wave-uniform scalars kept in LDS
  int ovCount, candCount, jobCount, overflow, unsupported, finCount, nContig, sortBad;
  int novelMin[2];
  int red[16];
  short contigA[64], contigB[64];
  int nvPossible[2], nvLongest[2];   // novel group statistics of GetOverlapsFromHits (filter 1)
  int nvN4[2], nvN5[2], nvSmax[2];   // groups of at least 4 / 5 hits and the largest group, TRUE sizes (the statistics measure a group one short or in full)
  int statsStable;                   // no pass of this read so far whose novelMinHitRequired could move (see overlapsFromKeys)
  int hullLo[2], hullHi[2];          // restricted re-query: per strand, hull of the read's projections along the diagonals that hold three or more hits with the contig (lo > hi: none)
  int useMarks;                      // restricted re-query: read the contig's postings off its posting marks (T4CandArgs::useMarks)
  int forceMin[2];                   // restricted re-query: novelMinHitRequired per strand as the entry's whole query had it (0: three hits), T4QueryArgs::forceMin
  int vjRescue;                      // the pass ended in the VJ-junction rescue (GetVJOverlapsFromHits looks ACROSS sequences: such a result is not the sum of per-contig parts)
  int nAll, nOther, strand0;          // GetOverlapsFromRead: overlaps on the strand of the best one (before the similarity cut), on the other strand, that strand
  int wideWant;                      // mode 4, nonzero: a pass that emits more hits than this (or outgrows the global-scratch tier) is handed to the wide query (t4_wide.h)
  unsigned hhBest[2];        // HasHitInSet: per strand, (distinct read offsets << 16) | (0xFFFF - bucket rank) of the best bucket
  long long phaseT0; int curPhase, phaseBase;
#ifdef T4_PHASE_TIMING
  unsigned phaseLocal[T4_NPHASE];
#endif
};

// Build segment chars (forward + reverse complement of the segment) from the packed read.
__device__ void loadSegment(const T4BatchView &bv, long long r, int segStart, int segLen, WaveMem &wm) {
  const unsigned *pk = bv.pk + r * bv.wpk;
  const unsigned *nm = bv.nm + r * bv.wnm;
  for (int i = tid(); i < segLen; i += nthr()) {
    int g = segStart + i;
    unsigned w = pk[g >> 4], m = nm[g >> 5];
    int code = (w >> ((g & 15) * 2)) & 3;
    bool isN = (m >> (g & 31)) & 1;
    char c = isN ? 'N' : (code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T');
    wm.seg[i] = c;
    wm.rc[segLen - 1 - i] = isN ? 'N' : (code == 0 ? 'T' : code == 1 ? 'G' : code == 2 ? 'C' : 'A');
  }
  if (tid() == 0) { wm.seg[segLen] = 0; wm.rc[segLen] = 0; }
  wm.pkRow = pk; wm.nmRow = nm; wm.segAbs = segStart; wm.segLen = segLen;
  __syncthreads();
}

// k-mer code at position p of chars S (N -> 0), and whether the window holds an N
__device__ __forceinline__ unsigned long long kmerAt(const char *S, int p, int K, bool &valid) {
  unsigned long long code = 0;
  valid = true;
  for (int j = 0; j < K; ++j) {
    char c = S[p + j];
    if (c == 'N') valid = false;
    code = (code << 2) | (unsigned long long)nuc2(c);
  }
  return code;
}

// The same code from the read's 2-bit packed words (K <= 16): position p of the forward segment, or of its reverse
// complement (rc == true). The words hold base j at bits 2j (little-endian digits); the index wants the first base in the
// most significant digit, so the forward code is the digit reversal of the extracted window, while the reverse-complement
// code is the window itself with every digit complemented (its digits already come in reverse order). N is packed as 0,
// which is also what nuc2('N') gives; its complement must stay 0.
__device__ __forceinline__ unsigned long long kmerPacked(const WaveMem &wm, bool rc, int p, int segLen, int K, bool &valid) {
  const int g = wm.segAbs + (rc ? segLen - K - p : p);
  const int w = g >> 4, sh = (g & 15) * 2;
  const unsigned long long win = ((unsigned long long)wm.pkRow[w] | ((unsigned long long)wm.pkRow[w + 1] << 32)) >> sh;
  const int wn = g >> 5;
  const unsigned nwin = (unsigned)((((unsigned long long)wm.nmRow[wn] | ((unsigned long long)wm.nmRow[wn + 1] << 32)) >> (g & 31)) & ((1ull << K) - 1ull));
  valid = nwin == 0;
  const unsigned long long m2 = (1ull << (2 * K)) - 1ull;
  if (rc) {
    unsigned long long nd = nwin;                       // N flags -> both bits of their digits
    nd = (nd | (nd << 16)) & 0x0000FFFF0000FFFFull; nd = (nd | (nd << 8)) & 0x00FF00FF00FF00FFull;
    nd = (nd | (nd << 4)) & 0x0F0F0F0F0F0F0F0Full; nd = (nd | (nd << 2)) & 0x3333333333333333ull;
    nd = (nd | (nd << 1)) & 0x5555555555555555ull; nd |= nd << 1;
    return (~win & m2) & ~nd;
  }
  unsigned long long r = __brevll(win & m2);            // bit reversal, then the two bits of every digit back in order
  r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
  return r >> (64 - 2 * K);
}

// Seed stage of one pass: fills posStart/posPref (aliased on wm.ov / wm.pairs) and returns the number
// of hit records H that GetHitsFromRead would emit (before the barcode filter). Wave-uniform result.
// vjOnly only changes the later expansion.
#ifdef __HIPCC__
#define T4_UNIFORM(v) (__builtin_amdgcn_readfirstlane((int)(v)))   // a value every lane holds alike, declared so
#else
#define T4_UNIFORM(v) ((int)(v))
#endif
// codeBuf (nullable): 2 * nk 64-bit words of scratch (the k-mer code of every position) for the wave-wide replay of the repeat-skip
// rule (below); without it the rule is replayed by one lane.
__device__ int seedPositions(const T4IndexView &ix, WaveMem &wm, int segLen, int strandArg, int barcode,
                             bool allowTotalSkip, unsigned *posStart, unsigned *posPref, int *red) {
  const int K = ix.k, lane = tid(), NT = nthr();
  const int nk = segLen - K + 1;           // k-mers per strand
  const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1ull) : ~0ull;
  int skipLimit = ix.firstIsRef ? 0 : K / 2;
  // raw list sizes (0 for invalid k-mers) and starts, for both strands; bit 31 of the size marks a k-mer whose code
  // equals the code of the previous position (K + 1 equal bases): the repeat test of SeqSet.hpp:1380 without a second
  // and third pass over the characters
  const unsigned SAME = 0x80000000u;
  int big = 0;
  for (int q = lane; q < 2 * nk; q += NT) {
    int st = q >= nk, p = st ? q - nk : q;
    bool active = st ? (strandArg != 1) : (strandArg != -1);
    const char *S = st ? wm.rc : wm.seg;
    unsigned start = 0, cnt = 0;
    bool valid;
    const unsigned long long code = (K <= 16 ? kmerPacked(wm, st != 0, p, segLen, K, valid) : kmerAt(S, p, K, valid)) & mask;
    if (active && valid) indexLookup(ix, code, barcode, start, cnt);
    bool same = false;
    if (p > 0) same = (((code >> 2) | ((unsigned long long)nuc2(S[p - 1]) << (2 * (K - 1)))) == code);
    posStart[q] = start; posPref[q] = cnt | (same ? SAME : 0u);
    if (cnt >= 100) big = 1;
  }
  big = blockSum(big, red) != 0;
  if ((skipLimit == 0 && !allowTotalSkip) || !big) {
    // no `continue` can fire: prevKmerCode is always the code of the previous position
    for (int q = lane; q < 2 * nk; q += NT) {
      const unsigned v = posPref[q];
      posPref[q] = (v & SAME) ? 0u : v;
    }
  } else {
    for (int q = lane; q < 2 * nk; q += NT) posPref[q] &= ~SAME;
    __syncthreads();
    if (lane == 0) {
    // sequential replay of the skip state machine (SeqSet.hpp:1370-1425, 1437-1498)
    unsigned long long prev = 0;
    for (int st = 0; st < 2; ++st) {
      bool active = st ? (strandArg != 1) : (strandArg != -1);
      if (!active) continue;
      const char *S = st ? wm.rc : wm.seg;
      unsigned long long code = 0;
      int skipCnt = 0;
      for (int i = 0; i < K - 1; ++i) code = ((code << 2) & mask) | (unsigned long long)nuc2(S[i]);
      for (int i = K - 1; i < segLen; ++i) {
        code = ((code << 2) & mask) | (unsigned long long)nuc2(S[i]);
        int p = i - K + 1, q = st * nk + p;
        unsigned size = posPref[q];
        bool emit = false;
        if (p == 0 || prev != code) {
          if (size >= 100 && p != 0 && i != segLen - 1 && skipCnt < skipLimit) { ++skipCnt; posPref[q] = 0; continue; }
          if (size >= 100 && allowTotalSkip) { posPref[q] = 0; continue; }
          skipCnt = 0;
          emit = true;
        }
        if (!emit) posPref[q] = 0;
        prev = code;
      }
    }
    }
  }
  __syncthreads();
  // exclusive prefix sums over the 2*nk positions
  int carry = 0;
  for (int q0 = 0; q0 < 2 * nk; q0 += NT) {
    int q = q0 + lane;
    int v = q < 2 * nk ? (int)posPref[q] : 0;
    int tot;
    int inc = blockInclScan(v, red, tot);
    if (q < 2 * nk) posPref[q] = (unsigned)(carry + inc - v);
    carry += tot;
  }
  if (lane == 0) posPref[2 * nk] = (unsigned)carry;
  __syncthreads();
  return carry;
}

// The same for the kernel variants that meet contig sets (repeat-skip rule live): the rule is replayed wave-wide, see below.
__device__ T4_NI int seedPositionsNovel(const T4IndexView &ix, WaveMem &wm, int segLen, int strandArg, int barcode,
                             bool allowTotalSkip, unsigned *posStart, unsigned *posPref, int *red, unsigned long long *codeBuf, WaveState *phaseWs) {
  constexpr bool NOVEL = true;
  const int K = ix.k, lane = tid(), NT = nthr();
  const int nk = segLen - K + 1;           // k-mers per strand
  const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1ull) : ~0ull;
  int skipLimit = ix.firstIsRef ? 0 : K / 2;
  // raw list sizes (0 for invalid k-mers) and starts, for both strands; bit 31 of the size marks a k-mer whose code
  // equals the code of the previous position (K + 1 equal bases): the repeat test of SeqSet.hpp:1380 without a second
  // and third pass over the characters
  const unsigned SAME = 0x80000000u;
  int big = 0;
  for (int q = lane; q < 2 * nk; q += NT) {
    int st = q >= nk, p = st ? q - nk : q;
    bool active = st ? (strandArg != 1) : (strandArg != -1);
    const char *S = st ? wm.rc : wm.seg;
    unsigned start = 0, cnt = 0;
    bool valid;
    const unsigned long long code = (K <= 16 ? kmerPacked(wm, st != 0, p, segLen, K, valid) : kmerAt(S, p, K, valid)) & mask;
    if (active && valid) indexLookup(ix, code, barcode, start, cnt);
    bool same = false;
    if (p > 0) same = (((code >> 2) | ((unsigned long long)nuc2(S[p - 1]) << (2 * (K - 1)))) == code);
    posStart[q] = start; posPref[q] = cnt | (same ? SAME : 0u);
    if (NOVEL && codeBuf) codeBuf[q] = code;
    if (cnt >= 100) big = 1;
  }
  big = blockSum(big, red) != 0;
  if (phaseWs) PHASE_MARK(phaseWs, 22);   // seed: the repeat-skip replay (phase-timing builds; callers whose `red` is not inside a WaveState pass no phaseWs)
  if ((skipLimit == 0 && !allowTotalSkip) || !big) {
    // no `continue` can fire: prevKmerCode is always the code of the previous position
    for (int q = lane; q < 2 * nk; q += NT) {
      const unsigned v = posPref[q];
      posPref[q] = (v & SAME) ? 0u : v;
    }
#ifndef T4_SEED_SERIAL
  } else if (NOVEL && codeBuf && !allowTotalSkip) {
    // The skip rule of GetHitsFromRead (SeqSet.hpp:1381-1391) is a small transducer: a k-mer with 100+ postings is passed over
    // (without becoming the "previous k-mer") while fewer than skipLimit have been passed over since the last emitted one; every
    // other k-mer is compared with the previous one that was not passed over, which lies at most skipLimit + 1 positions back.
    // So a position needs two facts that a lane derives for it from the codes of its neighbours -- is its list big, and which of
    // the skipLimit + 1 k-mers before it have its code -- and the replay itself runs over those bits in scalar registers, 64
    // positions per pass of the first wavefront, instead of one lane walking the arrays in LDS.
    const int D = skipLimit + 1;   // <= 16: k <= 31
    for (int q = lane; q < 2 * nk; q += NT) posPref[q] &= ~SAME;
    __syncthreads();
    if (lane < 64) {
      // everything the replay branches on is made wave-uniform for the compiler (readfirstlane / ballot), so that the loop over the
      // positions of a chunk is scalar code: bit tests on 64-bit masks, counters in SGPRs
      const int nkU = T4_UNIFORM(nk), skipLimitU = T4_UNIFORM(skipLimit), strandU = T4_UNIFORM(strandArg);
      for (int st = 0; st < 2; ++st) {
        const bool active = st ? (strandU != 1) : (strandU != -1);
        if (!active) continue;
        int skipCnt = 0, dist = 1;
        for (int c0 = 0; c0 < nkU; c0 += 64) {
          const int p = c0 + lane;
          unsigned F = 0u;
          if (p < nkU) {
            const int q = st * nkU + p;
            if (posPref[q] >= 100u) F = 1u;
            const unsigned long long code = codeBuf[q];
            for (int d = 1; d <= D && d <= p; ++d) if (codeBuf[q - d] == code) F |= 1u << d;
          }
          const unsigned long long bigM = __ballot((F & 1u) != 0), eq1 = __ballot((F & 2u) != 0);
          unsigned long long curEq = dist == 1 ? eq1 : __ballot(((F >> dist) & 1u) != 0);   // equality with the k-mer `dist` positions back
          unsigned long long emitMask = 0;
          const int lim = nkU - c0 < 64 ? nkU - c0 : 64;
          // ordinary positions -- a new code with a short list -- are emitted and reset the counter; whole runs of them are passed
          // in one step (most of a read), the rule itself is only walked through the others
          const unsigned long long ordinary = ~eq1 & ~bigM;
          for (int t = 0; t < lim; ++t) {
            const int pp = c0 + t;
            if (dist == 1 && pp != 0 && ((ordinary >> t) & 1ull)) {
              const unsigned long long rest = ~(ordinary >> t);                      // first position from t on that is not ordinary
              int run = rest ? (int)__builtin_ctzll(rest) : 64 - t;
              if (run > lim - t) run = lim - t;
              emitMask |= (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << t;
              skipCnt = 0;
              t += run - 1;
              continue;
            }
            if (pp == 0) { skipCnt = 0; emitMask |= 1ull; if (dist != 1) { dist = 1; curEq = eq1; } continue; }
            if (!((curEq >> t) & 1ull)) {   // differs from the previous k-mer that was not passed over
              if (((bigM >> t) & 1ull) && pp != nkU - 1 && skipCnt < skipLimitU) { ++skipCnt; ++dist; curEq = __ballot(((F >> dist) & 1u) != 0); continue; }
              skipCnt = 0;
              emitMask |= 1ull << t;
            }
            if (dist != 1) { dist = 1; curEq = eq1; }
          }
          if (p < nkU && !((emitMask >> lane) & 1ull)) posPref[st * nkU + p] = 0;
        }
      }
    }
#endif
  } else {
    for (int q = lane; q < 2 * nk; q += NT) posPref[q] &= ~SAME;
    __syncthreads();
    if (lane == 0) {
    // sequential replay of the skip state machine (SeqSet.hpp:1370-1425, 1437-1498)
    unsigned long long prev = 0;
    for (int st = 0; st < 2; ++st) {
      bool active = st ? (strandArg != 1) : (strandArg != -1);
      if (!active) continue;
      const char *S = st ? wm.rc : wm.seg;
      unsigned long long code = 0;
      int skipCnt = 0;
      for (int i = 0; i < K - 1; ++i) code = ((code << 2) & mask) | (unsigned long long)nuc2(S[i]);
      for (int i = K - 1; i < segLen; ++i) {
        code = ((code << 2) & mask) | (unsigned long long)nuc2(S[i]);
        int p = i - K + 1, q = st * nk + p;
        unsigned size = posPref[q];
        bool emit = false;
        if (p == 0 || prev != code) {
          if (size >= 100 && p != 0 && i != segLen - 1 && skipCnt < skipLimit) { ++skipCnt; posPref[q] = 0; continue; }
          if (size >= 100 && allowTotalSkip) { posPref[q] = 0; continue; }
          skipCnt = 0;
          emit = true;
        }
        if (!emit) posPref[q] = 0;
        prev = code;
      }
    }
    }
  }
  __syncthreads();
  if (phaseWs) PHASE_MARK(phaseWs, 23);   // seed: prefix sums
  // exclusive prefix sums over the 2*nk positions
  int carry = 0;
  for (int q0 = 0; q0 < 2 * nk; q0 += NT) {
    int q = q0 + lane;
    int v = q < 2 * nk ? (int)posPref[q] : 0;
    int tot;
    int inc = blockInclScan(v, red, tot);
    if (q < 2 * nk) posPref[q] = (unsigned)(carry + inc - v);
    carry += tot;
  }
  if (lane == 0) posPref[2 * nk] = (unsigned)carry;
  __syncthreads();
  return carry;
}

// Expand postings into sortable keys. Returns the number of valid keys (after barcode / VJ filters);
// invalid slots get the all-ones key and sort to the end.
__device__ int expandHits(const T4IndexView &ix, WaveMem &wm, int nk, int H, int barcode, bool vjOnly,
                          const unsigned *posStart, const unsigned *posPref, int *red, int k32 = 0) {
  const int lane = tid(), NT = nthr();
  int dropped = 0;
  const int cBits32 = (k32 >> 8) & 255, bias32 = (1 << cBits32) - 512;
  for (int s = lane; s < H; s += NT) {
    int lo = 0, hi = 2 * nk - 1;   // last q with posPref[q] <= s
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (posPref[mid] <= (unsigned)s) lo = mid; else hi = mid - 1;
    }
    int q = lo;
    int2 po = ix.post[posStart[q] + ((unsigned)s - posPref[q])];
    int st = q >= nk, a = st ? q - nk : q;
    bool keep = true;
    T4SeqInfo si;
    if (barcode != -1 || vjOnly) si = ix.seqs[po.x];   // the plain pass needs nothing from the sequence table here
    if (barcode != -1 && si.barcode != barcode) keep = false;
    if (vjOnly) { // GetVJOverlapsFromHits (SeqSet.hpp:1075-1089)
      if (!si.isRef) keep = false;
      else if (si.name3 == 'V') { if (!(po.y >= si.len - 31)) keep = false; }
      else if (si.name3 == 'J') { if (!(po.y < 31)) keep = false; }
      else keep = false;
    }
    unsigned long long key = ~0ull;
    if (keep) {
      unsigned long long sb = st ? 0ull : 1ull;  // minus strand sorts first
      key = (sb << 63) | ((unsigned long long)po.x << (T4_C_BITS + T4_B_BITS)) |
            ((unsigned long long)(a - po.y + T4_C_BIAS) << T4_B_BITS) | (unsigned long long)po.y;
    } else ++dropped;
    if (k32) {   // 32-bit form of the same order (t4Key32Bits), sorted as such and widened afterwards (seedChainPass)
      unsigned k = ~0u;
      if (keep) k = (st ? 0u : 1u << 31) | ((unsigned)po.x << (cBits32 + 9)) | ((unsigned)(a - po.y + bias32) << 9) | (unsigned)a;
      ((unsigned *)wm.keys)[s] = k;
    } else wm.keys[s] = key;
  }
  dropped = blockSum(dropped, red);
  return H - dropped;
}

// The hits with ONE sequence only (restricted re-query, T4QueryArgs::onlySeq), appended compactly: every posting of every emitted
// k-mer is looked at (H may be far beyond the key array), the few that name `seq` are kept. Returns their number, -1 when they
// outgrow the key array. Uses ws->candCount as the cursor.
__device__ int expandHitsOnly(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int nk, int H, int seq, const unsigned *posStart, const unsigned *posPref) {
  const int lane = tid(), NT = nthr();
  if (lane == 0) ws->candCount = 0;
  __syncthreads();
  for (int s = lane; s < H; s += NT) {
    int lo = 0, hi = 2 * nk - 1;   // last q with posPref[q] <= s
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (posPref[mid] <= (unsigned)s) lo = mid; else hi = mid - 1;
    }
    const int q = lo;
    const int2 po = ix.post[posStart[q] + ((unsigned)s - posPref[q])];
    if (po.x != seq) continue;
    const int st = q >= nk, a = st ? q - nk : q;
    const int at = atomicAdd(&ws->candCount, 1);
    if (at < wm.hitLimit)
      wm.keys[at] = ((st ? 0ull : 1ull) << 63) | ((unsigned long long)po.x << (T4_C_BITS + T4_B_BITS)) |
                    ((unsigned long long)(a - po.y + T4_C_BIAS) << T4_B_BITS) | (unsigned long long)po.y;
  }
  __syncthreads();
  const int n = ws->candCount;
  __syncthreads();
  return n <= wm.hitLimit ? n : -1;
}

// The same hits read off the CONTIG instead of the read's posting lists (restricted re-query with posting marks, T4CandArgs::useMarks):
// a posting (seq, o) exists for every offset o whose predicate byte carries a mark, its code is the contig's k-mer at o (the index
// is kept in step with the consensus: KmerIndex entries are removed and rebuilt whenever a consensus base changes, SeqSet.hpp:4537-
// 4588, 11058-11080), and it is a hit of read position q exactly when q was emitted by the seed stage and holds that code. A heavy
// read's restricted re-query then costs the contig's length, not the read's tens of thousands of postings. codeBuf: the codes of the
// read's 2 nk positions (left in the key array by the seed stage); the hits are built behind them and moved to the front.
// Returns their number, -1 when they outgrow the key array, -2 when the contig's marks are not to be trusted.
__device__ T4_NI int expandHitsContig(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int nk, int seq, const unsigned *posPref) {
  const int lane = tid(), NT = nthr(), K = ix.k;
  const T4SeqInfo si = ix.seqs[seq];
  if (si.len >= 1 && (ix.pw[si.pwOff] & T4_PW_MARK_BAD)) return -2;
  const unsigned long long *codeBuf = wm.keys;
  unsigned *head = wm.cand, *next = wm.cand + 1024;   // 1024 chain heads (q + 1; 0: empty), 2 nk links
  const int HOFF = 1024;                              // the hits start behind the codes (2 nk <= 768 of them)
  if (2 * nk > 768 || wm.candCap < 1024 + 768 || wm.hitLimit <= HOFF + 64) return -2;
  for (int i = lane; i < 1024; i += NT) head[i] = 0u;
  if (lane == 0) ws->candCount = 0;
  __syncthreads();
  for (int q = lane; q < 2 * nk; q += NT) {
    if (posPref[q + 1] == posPref[q]) continue;   // not emitted (or an empty list)
    const unsigned long long code = codeBuf[q];
    unsigned h = (unsigned)((code * 0x9E3779B97F4A7C15ull) >> 54);   // 10 bits
    for (;;) {
      const unsigned cur = head[h];
      if (cur == 0u) { if (atomicCAS(&head[h], 0u, (unsigned)q + 1u) == 0u) { next[q] = 0u; break; } continue; }
      if (codeBuf[cur - 1u] == code) { next[q] = atomicExch(&head[h], (unsigned)q + 1u); break; }   // (the slot stays with this code: every later head holds it too)
      h = (h + 1u) & 1023u;
    }
  }
  __syncthreads();
  const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1ull) : ~0ull;
  for (int o = lane; o + K <= si.len; o += NT) {
    const int m = (ix.pw[si.pwOff + o] >> T4_PW_MARK_SHIFT) & 3;
    if (!m) continue;
    unsigned long long code = 0;
    bool ok = true;
    for (int t = 0; t < K; ++t) { const char ch = ix.cons[si.consOff + o + t]; if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ok = false; code = (code << 2) | (unsigned long long)nuc2(ch); }
    if (!ok) continue;
    code &= mask;
    unsigned h = (unsigned)((code * 0x9E3779B97F4A7C15ull) >> 54);
    for (;;) {
      const unsigned cur = head[h];
      if (cur == 0u) break;
      if (codeBuf[cur - 1u] == code) {
        for (unsigned q1 = cur; q1 != 0u; q1 = next[q1 - 1u]) {
          const int q = (int)q1 - 1, st = q >= nk, a = st ? q - nk : q;
          for (int t = 0; t < m; ++t) {
            const int at = atomicAdd(&ws->candCount, 1);
            if (HOFF + at < wm.hitLimit)
              wm.keys[HOFF + at] = ((st ? 0ull : 1ull) << 63) | ((unsigned long long)seq << (T4_C_BITS + T4_B_BITS)) |
                                   ((unsigned long long)(a - o + T4_C_BIAS) << T4_B_BITS) | (unsigned long long)o;
          }
        }
        break;
      }
      h = (h + 1u) & 1023u;
    }
  }
  __syncthreads();
  const int n = ws->candCount;
  __syncthreads();
  if (HOFF + n > wm.hitLimit) return -1;
  for (int i0 = 0; i0 < n; i0 += NT) {   // to the front, a chunk at a time (the ranges overlap once n passes HOFF)
    const int i = i0 + lane;
    unsigned long long v = 0;
    if (i < n) v = wm.keys[HOFF + i];
    __syncthreads();
    if (i < n) wm.keys[i] = v;
    __syncthreads();
  }
  return n;
}

// Workgroup bitonic sort of keys[0, n) (any n): the all-ascending network on the next power of two with VIRTUAL +inf
// padding -- every compare-exchange moves the minimum to the lower index, so padding never moves and pairs that touch it
// are skipped (no element >= n is ever read or written; up to 2x less work than a padded sort). The compare-exchanges of
// the sub-steps with j <= 64 stay inside aligned chunks of 128 elements: each wavefront owns whole chunks through those
// sub-steps and only needs its own LDS ordering (wave barrier), so a stage costs log2(k) - 6 workgroup barriers, not log2(k).
__device__ __forceinline__ void waveLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <class KeyT>
__device__ void bitonicSort(KeyT *keys, int n) {
  const int lane = tid(), NT = nthr();
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  const int half = n2 >> 1;
  const int nChunk = (half + 63) >> 6;        // chunks of 64 pairs = 128 elements
  const int wave = lane >> 6, nw = NT >> 6, wl = lane & 63;
  for (int k = 2; k <= n2; k <<= 1) {
    // first sub-step of the stage: partner mirrored inside the k-block (i ^ (k - 1))
    int j = k >> 1;
    if (j > 64) {
      for (int t = lane; t < half; t += NT) {
        const int low = t & (j - 1), i = ((t & ~(j - 1)) << 1) | low, p = (i & ~(k - 1)) + (k - 1 - low);
        if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
      }
      __syncthreads();
      for (j >>= 1; j > 64; j >>= 1) {
        for (int t = lane; t < half; t += NT) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
          if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
        }
        __syncthreads();
      }
    }
    // chunk-local sub-steps (j <= 64): wave w owns chunks w, w + nw, ...
    for (int c = wave; c < nChunk; c += nw) {
      const int t = (c << 6) + wl;
      for (int jj = j; jj > 0; jj >>= 1) {
        if (t < half) {
          const int low = t & (jj - 1), i = ((t & ~(jj - 1)) << 1) | low;
          const int p = (jj == (k >> 1)) ? (i & ~(k - 1)) + (k - 1 - low) : (i | jj);
          if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
        }
        waveLdsSync();
      }
    }
    __syncthreads();
  }
}

// The same network for keys in GLOBAL memory (the global-scratch tier: tens of thousands of hits) staged through an LDS
// buffer of B keys (B a power of two): every sub-step whose partners lie inside an aligned block of B keys runs on the block
// in LDS, only the sub-steps with j >= B touch global memory -- 10 global sub-steps instead of 153 for 131072 keys.
// Padding up to the block size is an explicit +inf (all ones: no key is larger; dropped hits carry it too).
template <class KeyT> __device__ void bitonicSortReg(KeyT *keys, int n);   // below: chunk-local sub-steps in registers
template <class KeyT> __device__ void bitonicSortRegLds(KeyT *keys, int n);   // the same for keys known to live in LDS (ds_ instructions)
template <class KeyT> __device__ __forceinline__ void cmpExchReg(KeyT &lo, KeyT &hi, int mask, bool keepMax);
template <class KeyT>
__device__ void bitonicSortBlocked(KeyT *keys, int n, KeyT *lds, int B) {
  const int lane = tid(), NT = nthr();
  const KeyT INF = ~(KeyT)0;
  if (n <= B) {   // fits the buffer: one round trip
    for (int i = lane; i < n; i += NT) lds[i] = keys[i];
    __syncthreads();
    bitonicSortRegLds(lds, n);
    for (int i = lane; i < n; i += NT) keys[i] = lds[i];
    __syncthreads();
    return;
  }
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (int b0 = 0; b0 < n; b0 += B) {   // stages k = 2 .. B: every block sorted ascending
    for (int i = lane; i < B; i += NT) lds[i] = b0 + i < n ? keys[b0 + i] : INF;
    __syncthreads();
    bitonicSortRegLds(lds, B);
    for (int i = lane; i < B; i += NT) if (b0 + i < n) keys[b0 + i] = lds[i];
    __syncthreads();
  }
  const int half = n2 >> 1;
  for (int k = 2 * B; k <= n2; k <<= 1) {
    int j = k >> 1;
    for (int t = lane; t < half; t += NT) {   // first sub-step of the stage: partner mirrored inside the k-block
      const int low = t & (j - 1), i = ((t & ~(j - 1)) << 1) | low, p = (i & ~(k - 1)) + (k - 1 - low);
      if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
    }
    __syncthreads();
    for (j >>= 1; j >= B; j >>= 1) {
      for (int t = lane; t < half; t += NT) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
        if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
      }
      __syncthreads();
    }
    for (int b0 = 0; b0 < n; b0 += B) {   // sub-steps j = B / 2 .. 1 of the stage, block by block in LDS
      for (int i = lane; i < B; i += NT) lds[i] = b0 + i < n ? keys[b0 + i] : INF;
      __syncthreads();
      const int jLow = B >= 128 ? 64 : 0;   // sub-steps j <= 64 stay inside chunks of 128 keys: in registers, one LDS round trip
      for (int jj = B >> 1; jj > jLow; jj >>= 1) {
        for (int t = lane; t < (B >> 1); t += NT) {
          const int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), p = i | jj;
          KeyT a = lds[i], b = lds[p];
          if (a > b) { lds[i] = b; lds[p] = a; }
        }
        __syncthreads();
      }
      if (jLow) {
        const int wave = lane >> 6, nw = NT >> 6, wl = lane & 63;
        for (int c = wave; c < (B >> 7); c += nw) {
          const int base = c << 7;
          KeyT lo = lds[base + wl], hi = lds[base + 64 + wl];
          if (lo > hi) { const KeyT t = lo; lo = hi; hi = t; }
          for (int jj = 32; jj > 0; jj >>= 1) cmpExchReg(lo, hi, jj, (wl & jj) != 0);
          lds[base + wl] = lo; lds[base + 64 + wl] = hi;
        }
        __syncthreads();
      }
      for (int i = lane; i < B; i += NT) if (b0 + i < n) keys[b0 + i] = lds[i];
      __syncthreads();
    }
  }
}

#ifndef T4_OPT_REGSORT
#define T4_OPT_REGSORT 1
#endif
#ifndef T4_OPT_REGSORT64
#define T4_OPT_REGSORT64 1
#endif

#if T4_OPT_REGSORT
// The same network for 32-bit keys with the chunk-local sub-steps in registers: a lane holds elements wl and wl + 64 of its
// wavefront's 128-element chunk, partners are reached by lane exchanges (wl ^ mask), and a chunk is read and written once
// per stage instead of once per sub-step -- 56 of the 66 sub-steps of a 2048-key sort are chunk-local, and the LDS version
// is bound by LDS bandwidth. Out of line: its registers are its own.
template <class KeyT>
__device__ __forceinline__ void cmpExchReg(KeyT &lo, KeyT &hi, int mask, bool keepMax) {
  const KeyT olo = __shfl_xor(lo, mask), ohi = __shfl_xor(hi, mask);
  lo = keepMax ? (olo > lo ? olo : lo) : (olo < lo ? olo : lo);
  hi = keepMax ? (ohi > hi ? ohi : hi) : (ohi < hi ? ohi : hi);
}
template <class KeyT, class KeyPtr>
__device__ T4_NI void bitonicSortRegP(KeyPtr keys, int n) {
  const KeyT INF = ~(KeyT)0;
  const int lane = tid(), NT = nthr();
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  const int half = n2 >> 1;
  const int nChunk = (n + 127) >> 7;          // chunks of 128 elements that hold a real key
  const int wave = lane >> 6, nw = NT >> 6, wl = lane & 63;
  // stages k = 2 .. 128 never leave a chunk
  for (int c = wave; c < nChunk; c += nw) {
    const int base = c << 7;
    KeyT lo = base + wl < n ? keys[base + wl] : INF;
    KeyT hi = base + 64 + wl < n ? keys[base + 64 + wl] : INF;
    for (int k = 2; k <= 64; k <<= 1) {
      cmpExchReg(lo, hi, k - 1, (wl & (k >> 1)) != 0);                      // mirrored partner inside the k-block
      for (int jj = k >> 2; jj > 0; jj >>= 1) cmpExchReg(lo, hi, jj, (wl & jj) != 0);
    }
    {   // k = 128: element wl meets 127 - wl = the upper element of lane 63 - wl, and the other way round
      const KeyT olo = __shfl_xor(lo, 63), ohi = __shfl_xor(hi, 63);
      lo = ohi < lo ? ohi : lo;
      hi = olo > hi ? olo : hi;
      for (int jj = 32; jj > 0; jj >>= 1) cmpExchReg(lo, hi, jj, (wl & jj) != 0);
    }
    if (base + wl < n) keys[base + wl] = lo;
    if (base + 64 + wl < n) keys[base + 64 + wl] = hi;
  }
  __syncthreads();
  for (int k = 256; k <= n2; k <<= 1) {
    int j = k >> 1;
    for (int t = lane; t < half; t += NT) {   // first sub-step of the stage: mirrored partner
      const int low = t & (j - 1), i = ((t & ~(j - 1)) << 1) | low, p = (i & ~(k - 1)) + (k - 1 - low);
      if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
    }
    __syncthreads();
    for (j >>= 1; j > 64; j >>= 1) {
      for (int t = lane; t < half; t += NT) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
        if (p < n) { KeyT a = keys[i], b = keys[p]; if (a > b) { keys[i] = b; keys[p] = a; } }
      }
      __syncthreads();
    }
    for (int c = wave; c < nChunk; c += nw) {   // j = 64 .. 1 in registers
      const int base = c << 7;
      KeyT lo = base + wl < n ? keys[base + wl] : INF;
      KeyT hi = base + 64 + wl < n ? keys[base + 64 + wl] : INF;
      if (lo > hi) { const KeyT t = lo; lo = hi; hi = t; }
      for (int jj = 32; jj > 0; jj >>= 1) cmpExchReg(lo, hi, jj, (wl & jj) != 0);
      if (base + wl < n) keys[base + wl] = lo;
      if (base + 64 + wl < n) keys[base + 64 + wl] = hi;
    }
    __syncthreads();
  }
}
template <class KeyT> __device__ __forceinline__ void bitonicSortReg(KeyT *keys, int n) { bitonicSortRegP<KeyT, KeyT *>(keys, n); }
template <class KeyT> __device__ __forceinline__ void bitonicSortRegLds(KeyT *keys, int n) { bitonicSortRegP<KeyT, T4_LDS_AS KeyT *>((T4_LDS_AS KeyT *)keys, n); }
__device__ __forceinline__ void bitonicSort32(unsigned *keys, int n) { bitonicSortReg<unsigned>(keys, n); }
#endif
#ifndef T4_V0_NOKEYSORT
#define T4_V0_NOKEYSORT 0
#endif
#ifndef T4_V0_SERIALPRE
#define T4_V0_SERIALPRE 0
#endif
// (out of line, the call cost the rough-annotation kernels 9 % -- profiles/r02_annotate_inline_ab.txt -- although they never make it)
#ifndef T4_PREFILTER_NI
#define T4_PREFILTER_NI __forceinline__
#endif
#ifndef T4_OPT_OVKEYSORT
#define T4_OPT_OVKEYSORT 1
#endif
#ifndef T4_OPT_ROWCHAIN
#define T4_OPT_ROWCHAIN 1
#endif
#ifndef T4_OPT_R2CHECK
#define T4_OPT_R2CHECK 1
#endif
#ifndef T4_OPT_KEY32
#define T4_OPT_KEY32 1
#endif
#define KEY_G(k) ((unsigned)((k) >> (T4_C_BITS + T4_B_BITS)))
#define KEY_IDX(k) ((int)(((k) >> (T4_C_BITS + T4_B_BITS)) & ((1u << T4_IDX_BITS) - 1)))
#define KEY_PLUS(k) ((int)((k) >> 63))
#define KEY_C(k) ((int)(((k) >> T4_B_BITS) & ((1u << T4_C_BITS) - 1)))
#define KEY_B(k) ((int)((k) & ((1u << T4_B_BITS) - 1)))

// Chain one run [s, s+n) whose pairs are already in wm.pairs sorted by (b, a); append the overlap.
// Tail of chainRun: `mono` runs arrive with lisOut filled and both hit-length sums known; the others take the lane-serial LIS.
__device__ void chainFinish(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int s, int n, int seqIdx, int plus,
                            bool isRef, int hitLenRequired, bool mono, int hitLen, int hitLenSeq) {
  const int K = ix.k;
  unsigned *lisOut = (unsigned *)(wm.keys + s);
  unsigned short *top = (unsigned short *)(lisOut + n);
  unsigned short *link = top + n;
  int lisSize = n;
  if (!mono) {
    lisSize = lisLane(wm.pairs + s, n, lisOut, top, link);
    if (lisSize * K < hitLenRequired) return;
    hitLen = totalHitLen(lisOut, lisSize, K, false);
    hitLenSeq = totalHitLen(lisOut, lisSize, K, true);
  }
  if (lisSize * K < hitLenRequired) return;
  if (hitLen < hitLenRequired) return;
  if (hitLenSeq < hitLenRequired) return;
  OvRec o;
  o.seqIdx = seqIdx;
  o.rs = PA(lisOut[0]); o.re = PA(lisOut[lisSize - 1]) + K - 1;
  o.ss = PB(lisOut[0]); o.se = PB(lisOut[lisSize - 1]) + K - 1;
  o.matchCnt = 2 * hitLen; o.indelCnt = isRef ? 0 : n;   // (a novel overlap's run size rides here until scoring overwrites the field: ovKeepRunSize)
  o.chainPos = s; o.chainLen = lisSize;
  {
    const T4SeqInfo si = ix.seqs[seqIdx];   // gene class + chain letters ride along for the V/J/C selection
    o.flags = (plus ? OV_PLUS : 0) | (isRef ? OV_ISREF : 0) | OV_SIMZERO | ((int)si.geneType << 8) | ((int)si.name0 << 16) | ((int)si.name2 << 24);
  }
  if (!isRef && hitLen * 2 < o.se - o.ss + 1) return;
  int slot = atomicAdd(&ws->ovCount, 1);
  if (slot < wm.maxOv) wm.ov[slot] = o; else ws->overflow = 1;
}

__device__ void chainRun(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int s, int n, int seqIdx, int plus,
                         bool isRef, int hitLenRequired) {
  const int K = ix.k;
  unsigned *lisOut = (unsigned *)(wm.keys + s);
  // A run whose (b, a)-sorted hits are strictly increasing in both coordinates is its own LIS: the
  // equal-b collapse and the replacement sweep of LongestIncreasingSubsequence are then identities.
  // One pass for the common case: copy the run, test monotonicity and accumulate both GetTotalHitLength sums (for an
  // increasing chain, sum over segments of (last - first + K) == K + sum over neighbours of min(step, K)).
  bool mono = true;
  int hitLen = K, hitLenSeq = K;
  {
    unsigned prev = wm.pairs[s];
    lisOut[0] = prev;
    for (int t = 1; t < n; ++t) {
      const unsigned cur = wm.pairs[s + t];
      const int da = PA(cur) - PA(prev), db = PB(cur) - PB(prev);
      mono = mono && da > 0 && db > 0;
      hitLen += da < K ? da : K;
      hitLenSeq += db < K ? db : K;
      lisOut[t] = cur;
      prev = cur;
    }
  }
  chainFinish(ix, wm, ws, s, n, seqIdx, plus, isRef, hitLenRequired, mono, hitLen, hitLenSeq);
}


// R3 of overlapsFromKeys, one 16-lane row per candidate run (a read has ~65 candidate runs of ~12 hits and a few of 100+: one
// lane per run left the wavefront waiting for its longest run). Extraction, the (b, a) order test, the rank sort of the few
// unordered runs, the monotonicity test and both GetTotalHitLength sums run across the row; only a run that is not its own
// LIS (2 % of them) falls back to the lane-serial LongestIncreasingSubsequence. Loops hold no wave-level operation, so the
// rows of a wavefront may run different trip counts. T4_NI_R3: out of line (own register budget).
#if T4_NI_R3
__device__ T4_NI
#else
__device__ __forceinline__
#endif
void chainRunsRows(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int nCand, int hitLenRequired) {
  const int lane = tid(), NT = nthr(), K = ix.k;
    const int row = lane >> 4, rl = lane & 15, nRows = NT >> 4, rowShift = (lane & 63) & ~15;
    for (int c0 = 0; c0 < nCand; c0 += nRows) {
      const int c = c0 + row;
      const bool has = c < nCand;
      int s = 0, n = 0, idx = 0, plus = 0, adjustRadius = 0;
      bool isRef = false;
      if (has) {
        s = CAND_START(wm.cand[c]); n = CAND_LEN(wm.cand[c]);
        const unsigned long long ks = wm.keys[s];
        idx = KEY_IDX(ks); plus = KEY_PLUS(ks);
        isRef = seqIsRef(ix, idx);
        adjustRadius = isRef ? ix.radius : 0;
      }
      const bool big = adjustRadius > 0 && n > 48;   // ordered by R2 already
      waveLdsSync();                                  // every row has read its keys[s] before anything is overwritten
      if (has && !big)
        for (int t = rl; t < n; t += 16) {
          const unsigned long long kt = wm.keys[s + t];
          const int b = KEY_B(kt), a = KEY_C(kt) - T4_C_BIAS + b;
          wm.pairs[s + t] = ((unsigned)b << 12) | (unsigned)a;
        }
      waveLdsSync();
      bool unsorted = false;
      if (has && !big && adjustRadius > 0)
        for (int t = rl + 1; t < n; t += 16) unsorted = unsorted || wm.pairs[s + t - 1] > wm.pairs[s + t];
      unsorted = ((__ballot(unsorted) >> rowShift) & 0xFFFFull) != 0;
      if (__any(unsorted)) {   // a run is a union of a few diagonals: rank sort across the row (pairs of one run are distinct)
        unsigned *tmp = (unsigned *)(wm.keys + s) + n;
        if (unsorted)
          for (int t = rl; t < n; t += 16) {
            const unsigned v = wm.pairs[s + t];
            int rank = 0;
            for (int u = 0; u < n; ++u) rank += wm.pairs[s + u] < v ? 1 : 0;
            tmp[rank] = v;
          }
        waveLdsSync();
        if (unsorted) for (int t = rl; t < n; t += 16) wm.pairs[s + t] = tmp[t];
        waveLdsSync();
      }
      // a run whose (b, a)-sorted hits increase strictly in both coordinates is its own LIS (see chainRun)
      unsigned *lisOut = (unsigned *)(wm.keys + s);
      bool bad = false;
      int hitLen = 0, hitLenSeq = 0;
      if (has)
        for (int t = rl; t < n; t += 16) {
          const unsigned cur = wm.pairs[s + t];
          if (t > 0) {
            const unsigned prev = wm.pairs[s + t - 1];
            const int da = PA(cur) - PA(prev), db = PB(cur) - PB(prev);
            bad = bad || !(da > 0 && db > 0);
            hitLen += da < K ? da : K;
            hitLenSeq += db < K ? db : K;
          }
          lisOut[t] = cur;
        }
      bad = ((__ballot(bad) >> rowShift) & 0xFFFFull) != 0;
      hitLen = rowSum16(hitLen) + K;
      hitLenSeq = rowSum16(hitLenSeq) + K;
      waveLdsSync();
      if (has && rl == 0) chainFinish(ix, wm, ws, s, n, idx, plus, isRef, hitLenRequired, !bad, hitLen, hitLenSeq);
    }
}

// GetOverlapsFromHits (SeqSet.hpp:763-1063) on the sorted keys [0, Hv). Overlaps land in wm.ov.
// `filter` is the reference's filter argument. removeOnlyRepeats needs a hit with repeats > 10000,
// impossible while H <= 65535 only if ... it is handled by the caller refusing such reads (status).
__device__ void overlapsFromKeys(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int Hv, int hitLenRequired, int filter) {
  const int lane = tid(), NT = nthr(), K = ix.k;
  if (lane == 0) { ws->novelMin[0] = ws->novelMin[1] = 3; ws->candCount = 0; ws->nvPossible[0] = ws->nvPossible[1] = 0; ws->nvLongest[0] = ws->nvLongest[1] = 0;
                   ws->nvN4[0] = ws->nvN4[1] = ws->nvN5[0] = ws->nvN5[1] = ws->nvSmax[0] = ws->nvSmax[1] = 0; }
  if (filter == 1 && ix.hasNovel) {
    // Group statistics of the novel sequences (SeqSet.hpp:784-811). The reference walks the groups with `i = j` followed
    // by the loop's `++i`, i.e. it skips the first hit of the group that follows a measured one (a one-hit group vanishes,
    // and the group after it is measured in full). With n_t the true group sizes: skip_0 = false, skip_(t+1) = !(skip_t &&
    // n_t == 1), which unrolls to skip_t = (t - r - 1 is even) for r = the last index < t that is 0 or has n_r >= 2.
    // So: groups by scan/compaction, r by a max-scan, one lane per group.
    unsigned *gs = wm.pairs;   // dead until R1
    int nG = 0;
    for (int i0 = 0; i0 < Hv; i0 += NT) {
      const int i = i0 + lane;
      const bool st = i < Hv && (i == 0 || KEY_G(wm.keys[i]) != KEY_G(wm.keys[i - 1]));
      int tot;
      const int inc = blockInclScan(st ? 1 : 0, ws->red, tot);
      if (st) gs[nG + inc - 1] = (unsigned)i;
      nG += tot;
    }
    __syncthreads();
    int carry = -1;
    for (int t0 = 0; t0 < nG; t0 += NT) {
      const int t = t0 + lane;
      int u = -1, nT = 0, s0 = 0;
      if (t < nG) {
        s0 = (int)gs[t];
        nT = (t + 1 < nG ? (int)gs[t + 1] : Hv) - s0;
        if (t >= 1) { const int nPrev = s0 - (int)gs[t - 1]; if (t - 1 == 0 || nPrev >= 2) u = t - 1; }
      }
      int tot;
      int r = blockInclMaxScan(u, ws->red, tot);
      if (carry > r) r = carry;
      if (tot > carry) carry = tot;
      if (t < nG) {
        const bool skip = t >= 1 && (((t - r - 1) & 1) == 0);
        const int m = nT - (skip ? 1 : 0);
        const unsigned long long k0 = wm.keys[s0];
        if (m > 0 && !seqIsRef(ix, KEY_IDX(k0))) {
          const int plus = KEY_PLUS(k0);
          if (m > 3) atomicAdd(&ws->nvPossible[plus], 1);
          atomicMax(&ws->nvLongest[plus], m);
        }
        if (nT >= 4 && !seqIsRef(ix, KEY_IDX(k0))) {
          const int plus = KEY_PLUS(k0);
          atomicAdd(&ws->nvN4[plus], 1);
          if (nT >= 5) atomicAdd(&ws->nvN5[plus], 1);
          atomicMax(&ws->nvSmax[plus], nT);
        }
      }
    }
    __syncthreads();
    if (lane == 0)
      for (int t = 0; t <= 1; ++t) {
        const int possible = ws->nvPossible[t], longest = ws->nvLongest[t];
        if (possible > 100000) ws->novelMin[t] = (int)(longest * 0.75);
        else if (possible > 10000) ws->novelMin[t] = longest / 2;
        else if (possible > 1000) ws->novelMin[t] = longest / 3;
        else if (possible > 100) ws->novelMin[t] = longest / 4;
        // Could these thresholds move while every group of three or more hits stays as it is? An index edit that only touches
        // smaller groups changes which groups the walk above measures one short: a group of TRUE size n is measured n or
        // n - 1 whatever happens around it, so `possible` stays within [groups of >= 5, groups of >= 4] and `longest`
        // within [largest - 1, largest]. When both ends give the same threshold (and no smaller group can reach it), the
        // result of this pass does not depend on such edits: the ordered contig builder keeps the read's cached result.
        const int lo = ws->nvN5[t], hi = ws->nvN4[t], big = ws->nvSmax[t];
        const int cLo = lo > 100000 ? 4 : lo > 10000 ? 3 : lo > 1000 ? 2 : lo > 100 ? 1 : 0;
        const int cHi = hi > 100000 ? 4 : hi > 10000 ? 3 : hi > 1000 ? 2 : hi > 100 ? 1 : 0;
        bool ok = cLo == cHi;
        if (ok && cLo > 0) {
          const int a = big - 1 > 0 ? big - 1 : 0;
          const int fa = cLo == 4 ? (int)(a * 0.75) : cLo == 3 ? a / 2 : cLo == 2 ? a / 3 : a / 4;
          const int fb = cLo == 4 ? (int)(big * 0.75) : cLo == 3 ? big / 2 : cLo == 2 ? big / 3 : big / 4;
          ok = fa == fb && fa >= 3;
        }
        if (!ok) ws->statsStable = 0;
      }
  }
  if (filter == 0 && lane == 0) { if (ws->forceMin[0] > 0) ws->novelMin[0] = ws->forceMin[0]; if (ws->forceMin[1] > 0) ws->novelMin[1] = ws->forceMin[1]; }
  __syncthreads();
  PHASE_MARK(ws, 5);
  // R1: run starts are compacted (wm.pairs is dead here); a run is the stretch up to the next start, so nobody walks a
  // run hit by hit. Qualifying runs become candidates.
  unsigned *starts = wm.pairs;
  int nRuns = 0;
  for (int i0 = 0; i0 < Hv; i0 += NT) {
    const int i = i0 + lane;
    bool runStart = false;
    if (i < Hv) {
      const unsigned long long ki = wm.keys[i];
      runStart = true;
      if (i > 0) {
        const unsigned long long kp = wm.keys[i - 1];
        const int adjustRadius = seqIsRef(ix, KEY_IDX(ki)) ? ix.radius : 0;
        if (KEY_G(kp) == KEY_G(ki) && KEY_C(ki) - KEY_C(kp) <= adjustRadius) runStart = false;
      }
    }
    int tot;
    const int inc = blockInclScan(runStart ? 1 : 0, ws->red, tot);
    if (runStart) starts[nRuns + inc - 1] = (unsigned)i;
    nRuns += tot;
  }
  __syncthreads();
  for (int r = lane; r < nRuns; r += NT) {
    const int i = (int)starts[r], e = r + 1 < nRuns ? (int)starts[r + 1] : Hv;
    const int n = e - i;
    const unsigned long long ki = wm.keys[i];
    const bool isRef = seqIsRef(ix, KEY_IDX(ki));
    const int minHit = isRef ? 3 : ws->novelMin[KEY_PLUS(ki)];
    if (n >= minHit && n * K >= hitLenRequired) {
      int slot = atomicAdd(&ws->candCount, 1);
      if (n >= (1 << (32 - CAND_START_BITS))) ws->unsupported = 1;   // a run of 16384 hits: not with reads of a few hundred bases
      else if (slot < wm.candCap) wm.cand[slot] = (unsigned)i | ((unsigned)n << CAND_START_BITS);
      else ws->overflow = 1;
    }
  }
  __syncthreads();
  const int nCand = ws->candCount < wm.candCap ? ws->candCount : wm.candCap;
  PHASE_MARK(ws, 6);
  // R2: long multi-diagonal runs (reference genes only) are ordered by (b, a), one run per wavefront (runs own disjoint
  // slices of pairs / keys, so the wavefronts need no workgroup barrier between them, only their own LDS ordering)
  if (ix.radius > 0) {
    const int wave = lane >> 6, nw = NT >> 6, wl = lane & 63;
    for (int c = wave; c < nCand; c += nw) {
      int s = CAND_START(wm.cand[c]), n = CAND_LEN(wm.cand[c]);
      if (n <= 48) continue;                                  // wave-uniform
      unsigned long long ks = wm.keys[s];
      if (!seqIsRef(ix, KEY_IDX(ks))) continue;               // wave-uniform
      // upper half of the run's own key area; keys[s] itself (u32 words 0,1) stays intact for R3
      unsigned *tmp = (unsigned *)(wm.keys + s) + n;
      for (int t = wl; t < n; t += 64) {
        unsigned long long kt = wm.keys[s + t];
        int b = KEY_B(kt), a = KEY_C(kt) - T4_C_BIAS + b;
        wm.pairs[s + t] = ((unsigned)b << 12) | (unsigned)a;
      }
      waveLdsSync();
#if T4_OPT_R2CHECK
      {   // a long run on a single diagonal (an exact gene match) is in (b, a) order already
        bool unsorted = false;
        for (int t = wl + 1; t < n; t += 64) unsorted = unsorted || wm.pairs[s + t - 1] > wm.pairs[s + t];
        if (!__any(unsorted)) continue;                       // wave-uniform
      }
#endif
      for (int t = wl; t < n; t += 64) {
        unsigned v = wm.pairs[s + t];
        int rank = 0;
        for (int u = 0; u < n; ++u) { unsigned x = wm.pairs[s + u]; rank += (x < v) ? 1 : 0; }
        tmp[rank] = v;                                        // pairs of one run are distinct
      }
      waveLdsSync();
      for (int t = wl; t < n; t += 64) wm.pairs[s + t] = tmp[t];
      waveLdsSync();
    }
  }
  __syncthreads();   // R3 overwrites key areas that R2's wave-uniform tests read
  PHASE_MARK(ws, 7);
#if T4_OPT_ROWCHAIN
  chainRunsRows(ix, wm, ws, nCand, hitLenRequired);
  __syncthreads();
}
#else
  // R3: one lane per candidate run: extract (b << 12 | a), order by (b, a), chain
  for (int c = lane; c < nCand; c += NT) {
    int s = CAND_START(wm.cand[c]), n = CAND_LEN(wm.cand[c]);
    unsigned long long ks = wm.keys[s];
    int idx = KEY_IDX(ks), plus = KEY_PLUS(ks);
    bool isRef = seqIsRef(ix, idx);
    int adjustRadius = isRef ? ix.radius : 0;
    if (!(adjustRadius > 0 && n > 48)) {
      for (int t = s; t < s + n; ++t) {
        unsigned long long kt = wm.keys[t];
        int b = KEY_B(kt), a = KEY_C(kt) - T4_C_BIAS + b;
        wm.pairs[t] = ((unsigned)b << 12) | (unsigned)a;
      }
      if (adjustRadius > 0) { // insertion sort by (b, a): a run is a union of a few diagonals
        for (int t = s + 1; t < s + n; ++t) {
          unsigned v = wm.pairs[t];
          int u = t - 1;
          while (u >= s && wm.pairs[u] > v) { wm.pairs[u + 1] = wm.pairs[u]; --u; }
          wm.pairs[u + 1] = v;
        }
      }
    }
    chainRun(ix, wm, ws, s, n, idx, plus, isRef, hitLenRequired);
  }
  __syncthreads();
}
#endif

// IsOverlapLowComplex (SeqSet.hpp:590-617)
// SeqSet::IsLowComplexity-style test of GetOverlapsFromRead (SeqSet.hpp:2036-2062) on segment positions [rs, re] of the
// strand the overlap is on. Only the multiset of the four base counts matters, and the reverse complement permutes the
// counts of the mirrored forward interval, so both strands count on the read's 2-bit packed words: 16 bases per popcount
// round instead of one LDS byte at a time.
__device__ __forceinline__ bool lowComplex(const WaveMem &wm, bool plus, int rs, int re) {
  int lo = plus ? rs : wm.segLen - 1 - re, hi = plus ? re : wm.segLen - 1 - rs;
  lo += wm.segAbs; hi += wm.segAbs;
  int cA = 0, cC = 0, cG = 0, cT = 0;
  for (int w = lo >> 4; w <= (hi >> 4); ++w) {
    const unsigned x = wm.pkRow[w];
    unsigned nb = (wm.nmRow[w >> 1] >> ((w & 1) * 16)) & 0xFFFFu;   // N flags of these 16 bases -> even bit positions
    nb = (nb | (nb << 8)) & 0x00FF00FFu; nb = (nb | (nb << 4)) & 0x0F0F0F0Fu;
    nb = (nb | (nb << 2)) & 0x33333333u; nb = (nb | (nb << 1)) & 0x55555555u;
    unsigned v = 0x55555555u & ~nb;
    const int b0 = w << 4;
    if (lo > b0) v &= ~0u << ((lo - b0) * 2);
    if (hi < b0 + 15) v &= ~0u >> ((b0 + 15 - hi) * 2);
    const unsigned l = x & 0x55555555u, h = (x >> 1) & 0x55555555u;
    cA += __popc(~h & ~l & v); cC += __popc(~h & l & v); cG += __popc(h & ~l & v); cT += __popc(h & l & v);
  }
  const int cnt[4] = {cA, cC, cG, cT};
  int lowCnt = 0, lowTotal = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) { ++lowCnt; lowTotal += cnt[i]; }
  if (lowTotal * 7 >= re - rs + 1) return false;
  return lowCnt >= 2;
}

// Quick exits of the two gap aligners: empty side, 1 x 1, and the equal-length cases that provably (affine) or
// by the reference's own early return (posWeight, AlignAlgo.hpp:81-103) stay on the main diagonal.
// Returns true when the packed GetAlignStats counts could be produced without a banded DP.
__device__ bool dpAffineQuick(const char *t, int lent, const char *p, int lenp, unsigned &cnt) {
  cnt = 0;
  if (lent == 0 || lenp == 0) return true;
  if (lent != lenp) return false;
  // Equal lengths: every alignment leaving the diagonal opens an insertion and a deletion (2 x -5), so it
  // scores <= 2*i - 12 on a prefix of length i while the diagonal scores 2*i - 4*mm_i: for mm <= 3 the diagonal
  // is optimal on every prefix and the traceback, which tests the diagonal move first (AlignAlgo.hpp:338-349),
  // walks it. (1 x 1 is the reference's own special case and gives the same counts.)
  int mm = 0;
  for (int i = 0; i < lent; ++i) { char a = t[i], b = p[i]; mm += (a == b || a == 'N' || b == 'N') ? 0 : 1; }
  if (lent > 1 && mm > 3) return false;
  cnt = CNT_MATCH * (unsigned)(lent - mm) + CNT_MIS * (unsigned)mm;
  return true;
}
__device__ bool dpPosWeightQuick(const T4PW *w, int lent, const char *p, int lenp, unsigned &cnt) {
  cnt = 0;
  if (lent == 0 || lenp == 0) return true;
  if (lent != lenp) return false;
  int mm = 0;
  for (int i = 0; i < lent; ++i) mm += baseEqualW(w[i], p[i]) ? 0 : 1;
  if (lent > 1 && !((lent - mm) * 2 - mm * 2 >= lent * 2 - 8)) return false;
  cnt = CNT_MATCH * (unsigned)(lent - mm) + CNT_MIS * (unsigned)mm;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative banded gap DP: lane d owns band column d (W = 11 + |lent - lenp| <= 64 columns) and the
// cells are swept in the skewed order s = 2*i + d, in which the left neighbour (i, d-1) and the upper
// neighbour (i-1, d+1) were both produced at step s-1 by the adjacent lanes and the diagonal neighbour
// (i-1, d) is the lane's own previous cell: one alignment costs 2*lenp + W shuffle rounds instead of
// lenp * W serial cell updates. Scores, borders, sentinels and tie-breaks are those of
// AlignAlgo::GlobalAlignment (PW == false) / GlobalAlignment_PosWeight (PW == true); the result is the packed
// GetAlignStats of the reference's traceback (see the forward-count recurrences above dpAffineFwd).
// Every lane of the wave must call it with the same arguments. tbuf: >= lent bytes of LDS (affine only).
// ------------------------------------------------------------------------------------------------
template <bool PW>
__device__ unsigned dpWave(const char *t, const T4PW *w, int lent, const char *p, int lenp, char *tbuf) {
  const int d = laneId();
  if (lent == 0 || lenp == 0) return 0u;
  if (lent == 1 && lenp == 1) {
    bool eq = PW ? baseEqualW(w[0], p[0]) : (t[0] == p[0] || t[0] == 'N' || p[0] == 'N');
    return eq ? CNT_MATCH : CNT_MIS;
  }
  if (PW && lent == lenp) {   // the reference's ungapped early return (AlignAlgo.hpp:81-103)
    int mm = 0;
    for (int i = d; i < lent; i += 64) mm += baseEqualW(w[i], p[i]) ? 0 : 1;
    mm = waveSum(mm);
    if ((lent - mm) * 2 - mm * 2 >= lent * 2 - 8) return CNT_MATCH * (unsigned)(lent - mm) + CNT_MIS * (unsigned)mm;
  }
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  const int W = leftBand + rightBand + 1;
  if (W > 64 || lent > T4_MAXGAP || lenp > T4_MAXGAP) return DP_FAIL;
  // the target's characters / predicate bytes next to the wavefront (one byte per lane per step: read from global memory its
  // latency would be the cost of the step)
  for (int i = d; i < lent; i += 64) tbuf[i] = PW ? (char)w[i] : t[i];
  const int negInf = (lent + 1) * (lenp + 1) * (-4);
  const int e0 = -4 + (lenp + 1) * (-4);
  const int q4 = 4 * (lenp + 1);
  // the lane's cell of row 0
  int M = negInf, E = negInf, F = negInf;
  unsigned C0 = 0, C1 = 0, C2 = 0;
  {
    int j0 = d - leftBand;
    if (d < W && j0 >= 0 && j0 <= lent) {
      if (j0 == 0) { M = 0; E = 0; F = 0; }
      else if (PW) { M = -4 - 4 * j0; C0 = CNT_MATCH + CNT_INDEL * (unsigned)(j0 - 1); }
      else { M = -4 - 4 * j0; E = e0; F = -4 - j0; C0 = CNT_INDEL * (unsigned)(j0 + (j0 > q4 ? 1 : 0)); C1 = CNT_INDEL * (unsigned)(1 + j0); C2 = CNT_INDEL * (unsigned)j0; }
    }
  }
  const int lastStep = 2 * lenp + W - 1;
  for (int s = 2; s <= lastStep; ++s) {
    int lM = waveUp1(M), uM = waveDown1(M);
    unsigned lC0 = waveUp1(C0), uC0 = waveDown1(C0);
    int lF = 0, uE = 0; unsigned lC2 = 0, uC1 = 0;
    if (!PW) { lF = waveUp1(F); lC2 = waveUp1(C2); uE = waveDown1(E); uC1 = waveDown1(C1); }
    const int i2 = s - d, i = i2 >> 1, j = i - leftBand + d;
    if (d < W && (i2 & 1) == 0 && i >= 1 && i <= lenp && j >= 1 && j <= lent) {
      if (j == 1) {             // left neighbour is column 0
        lM = -4 - 4 * i;
        if (PW) lC0 = CNT_MATCH + CNT_INDEL * (unsigned)(i - 1);
        else { lF = -4 - 4 * i; lC0 = CNT_INDEL * (unsigned)i; lC2 = CNT_INDEL * (unsigned)(1 + i); }
      } else if (d == 0) { lM = negInf; lF = negInf; lC0 = 0; lC2 = 0; }
      if (i == 1) {             // upper neighbour is row 0
        uM = -4 - 4 * j;
        if (PW) uC0 = CNT_MATCH + CNT_INDEL * (unsigned)(j - 1);
        else { uE = e0; uC0 = CNT_INDEL * (unsigned)(j + (j > q4 ? 1 : 0)); uC1 = CNT_INDEL * (unsigned)(1 + j); }
      } else if (d + 1 >= W) { uM = negInf; uE = negInf; uC0 = 0; uC1 = 0; }
      int dM; unsigned dC0;     // diagonal neighbour
      if (i == 1) {
        int jj = j - 1;
        dM = jj == 0 ? 0 : -4 - 4 * jj;
        dC0 = jj == 0 ? 0u : (PW ? CNT_MATCH + CNT_INDEL * (unsigned)(jj - 1) : CNT_INDEL * (unsigned)(jj + (jj > q4 ? 1 : 0)));
      } else if (j == 1) {
        dM = -4 - 4 * (i - 1);
        dC0 = PW ? CNT_MATCH + CNT_INDEL * (unsigned)(i - 2) : CNT_INDEL * (unsigned)(i - 1);
      } else { dM = M; dC0 = C0; }
      const char pc = p[i - 1];
      if (PW) {
        const bool eq = baseEqualW((T4PW)tbuf[j - 1], pc);
        const int dsc = dM + (eq ? 2 : -2);
        int m = dsc;
        if (lM - 4 > m) m = lM - 4;
        if (uM - 4 > m) m = uM - 4;
        unsigned c;
        if (dsc == m) c = dC0 + (eq ? CNT_MATCH : CNT_MIS);
        else if (uM - 4 == m) c = uC0 + CNT_INDEL;
        else c = lC0 + CNT_INDEL;
        M = m; C0 = c;
      } else {
        const char tc = tbuf[j - 1];
        const bool eq = (tc == pc || tc == 'N' || pc == 'N');
        int e = uE - 1, eo = uM - 5;
        if (eo > e) e = eo;
        int f = lF - 1, fo = lM - 5;
        if (fo > f) f = fo;
        const int dsc = dM + (eq ? 2 : -2);
        int m = dsc;
        if (e > m) m = e;
        if (f > m) m = f;
        const unsigned c1 = CNT_INDEL + ((eo == e) ? uC0 : uC1);
        const unsigned c2 = CNT_INDEL + ((fo == f) ? lC0 : lC2);
        unsigned c0;
        if (dsc == m) c0 = dC0 + (eq ? CNT_MATCH : CNT_MIS);
        else c0 = (f >= e) ? c2 : c1;
        M = m; E = e; F = f; C0 = c0; C1 = c1; C2 = c2;
      }
    }
  }
  return __shfl(C0, lent - lenp + leftBand);
}


template <bool LDS> struct T4TbufPtr { typedef const char *type; };
template <> struct T4TbufPtr<true> { typedef const T4_LDS_AS char *type; };

// The same recurrences with EIGHT alignments per wavefront and no idle cell slot. A band is 11 + |lent - lenp| columns,
// almost always <= 16; in the skewed order s = 2i + d a lane that owns ONE column has a cell only every other step. So
// lane L of a group of 8 owns the column pair (2L, 2L+1): in step pair u it computes row i = u - L of both columns,
//   A = (i, 2L):   left (i, 2L-1) = lane L-1's B of the previous pair (one DPP row shift), upper = own old B, diagonal = own old A
//   B = (i, 2L+1): left = own new A, upper (i-1, 2L+2) = lane L+1's new A (one DPP row shift), diagonal = own old B
// i.e. 8 DPP moves and two full cell updates per pair, every lane busy, two alignments per 16-lane DPP row. Every lane
// passes the job of its group of 8 (has == false: none); jobs whose band is wider than 16 columns or whose sides exceed
// T4_MAXGAP (or the group's tcap bytes of LDS at tbuf) get DP_FAIL and are left to dpWave. All 64 lanes call it.
// p always lives in LDS (the read's segment); TLDS says that tbuf does too (every tier but the global-scratch one): the loop then
// reads its characters with ds_read (lgkmcnt only, in order) instead of flat loads, which the prefetch of the next step needs.
template <bool PW, bool TLDS>
__device__ T4_NI unsigned dpOct(bool has, const char *t, const T4PW *w, int lent, const char *p, int lenp, char *tbuf, int tcap) {
  const int L = laneId() & 7, grpBase = laneId() & 56;
  const T4_LDS_AS char *const pl = (const T4_LDS_AS char *)p;
  typedef typename T4TbufPtr<TLDS>::type TbufPtr;
  const TbufPtr tl = (TbufPtr)tbuf;
  unsigned result = 0u;
  bool run = has;
  if (run && (lent == 0 || lenp == 0)) { result = 0u; run = false; }
  if (run && lent == 1 && lenp == 1) {
    bool eq = PW ? baseEqualW(w[0], p[0]) : (t[0] == p[0] || t[0] == 'N' || p[0] == 'N');
    result = eq ? CNT_MATCH : CNT_MIS; run = false;
  }
  if (PW) {   // the reference's ungapped early return (AlignAlgo.hpp:81-103)
    const bool chk = run && lent == lenp;
    int mm = 0;
    if (chk) for (int i = L; i < lent; i += 8) mm += baseEqualW(w[i], p[i]) ? 0 : 1;
    mm += __shfl_xor(mm, 1); mm += __shfl_xor(mm, 2); mm += __shfl_xor(mm, 4);
    if (chk && (lent - mm) * 2 - mm * 2 >= lent * 2 - 8) { result = CNT_MATCH * (unsigned)(lent - mm) + CNT_MIS * (unsigned)mm; run = false; }
  }
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  const int W = leftBand + rightBand + 1;
  if (run && (W > 16 || lent > T4_MAXGAP || lenp > T4_MAXGAP || lent > tcap)) { result = DP_FAIL; run = false; }
  if (run) { for (int i = L; i < lent; i += 8) tbuf[i] = PW ? (char)w[i] : t[i]; }
  const int negInf = (lent + 1) * (lenp + 1) * (-4);
  const int e0 = -4 + (lenp + 1) * (-4);
  const int q4 = 4 * (lenp + 1);
  // row 0 of the lane's two columns
  int MA = negInf, EA = negInf, FA = negInf, MB = negInf, EB = negInf, FB = negInf;
  unsigned C0A = 0, C1A = 0, C2A = 0, C0B = 0, C1B = 0, C2B = 0;
  if (run) {
    for (int c = 0; c < 2; ++c) {
      const int dcol = 2 * L + c, j0 = dcol - leftBand;
      int m = negInf, e = negInf, f = negInf; unsigned c0 = 0, c1 = 0, c2 = 0;
      if (dcol < W && j0 >= 0 && j0 <= lent) {
        if (j0 == 0) { m = 0; e = 0; f = 0; }
        else if (PW) { m = -4 - 4 * j0; c0 = CNT_MATCH + CNT_INDEL * (unsigned)(j0 - 1); }
        else { m = -4 - 4 * j0; e = e0; f = -4 - j0; c0 = CNT_INDEL * (unsigned)(j0 + (j0 > q4 ? 1 : 0)); c1 = CNT_INDEL * (unsigned)(1 + j0); c2 = CNT_INDEL * (unsigned)j0; }
      }
      if (c == 0) { MA = m; EA = e; FA = f; C0A = c0; C1A = c1; C2A = c2; } else { MB = m; EB = e; FB = f; C0B = c0; C1B = c1; C2B = c2; }
    }
  }
  const int lastPair = run ? lenp + ((W - 1) >> 1) : 0;
  int maxPair = lastPair;
  for (int o = 8; o < 64; o <<= 1) { int v = __shfl_xor(maxPair, o); if (v > maxPair) maxPair = v; }
  for (int u = 1; u <= maxPair; ++u) {
    const int i = u - L;
    const bool rowOk = run && u <= lastPair && i >= 1 && i <= lenp;
    const char pc = rowOk ? pl[i - 1] : 'A';
    // ---- column A = 2L: left neighbour from lane L-1's B (previous pair)
    int lM = rowUp1E(MB, negInf); unsigned lC0 = rowUp1E(C0B, 0u);
    int lF = 0; unsigned lC2 = 0;
    if (!PW) { lF = rowUp1E(FB, negInf); lC2 = rowUp1E(C2B, 0u); }
    if (L == 0) { lM = negInf; lC0 = 0; lF = negInf; lC2 = 0; }   // the group's first lane: column -1 (the DPP edge only covers the row's)
    const int oMA = MA; const unsigned oC0A = C0A;                 // A's diagonal neighbour (i-1, 2L): own A of the previous pair
    const int oMB = MB, oEB = EB; const unsigned oC0B = C0B, oC1B = C1B;   // A's upper and B's diagonal neighbour: own B of the previous pair
    {
      const int dcol = 2 * L, j = i - leftBand + dcol;
      if (rowOk && dcol < W && j >= 1 && j <= lent) {
        int uM = oMB, uE = oEB; unsigned uC0 = oC0B, uC1 = oC1B;   // upper neighbour (i-1, 2L+1): own B of the previous pair
        int dM = oMA; unsigned dC0 = oC0A;
        if (i == 1 || j == 1) {
          if (j == 1) {
            lM = -4 - 4 * i;
            if (PW) lC0 = CNT_MATCH + CNT_INDEL * (unsigned)(i - 1);
            else { lF = -4 - 4 * i; lC0 = CNT_INDEL * (unsigned)i; lC2 = CNT_INDEL * (unsigned)(1 + i); }
          }
          if (i == 1) {
            uM = -4 - 4 * j;
            if (PW) uC0 = CNT_MATCH + CNT_INDEL * (unsigned)(j - 1);
            else { uE = e0; uC0 = CNT_INDEL * (unsigned)(j + (j > q4 ? 1 : 0)); uC1 = CNT_INDEL * (unsigned)(1 + j); }
            const int jj = j - 1;
            dM = jj == 0 ? 0 : -4 - 4 * jj;
            dC0 = jj == 0 ? 0u : (PW ? CNT_MATCH + CNT_INDEL * (unsigned)(jj - 1) : CNT_INDEL * (unsigned)(jj + (jj > q4 ? 1 : 0)));
          } else {
            dM = -4 - 4 * (i - 1);
            dC0 = PW ? CNT_MATCH + CNT_INDEL * (unsigned)(i - 2) : CNT_INDEL * (unsigned)(i - 1);
          }
        }
        if (PW) {
          const bool eq = baseEqualW((T4PW)tl[j - 1], pc);
          const int dsc = dM + (eq ? 2 : -2);
          int m = dsc;
          if (lM - 4 > m) m = lM - 4;
          if (uM - 4 > m) m = uM - 4;
          unsigned c;
          if (dsc == m) c = dC0 + (eq ? CNT_MATCH : CNT_MIS);
          else if (uM - 4 == m) c = uC0 + CNT_INDEL;
          else c = lC0 + CNT_INDEL;
          MA = m; C0A = c;
        } else {
          const char tc = tl[j - 1];
          const bool eq = (tc == pc || tc == 'N' || pc == 'N');
          int e = uE - 1, eo = uM - 5;
          if (eo > e) e = eo;
          int f = lF - 1, fo = lM - 5;
          if (fo > f) f = fo;
          const int dsc = dM + (eq ? 2 : -2);
          int m = dsc;
          if (e > m) m = e;
          if (f > m) m = f;
          const unsigned c1 = CNT_INDEL + ((eo == e) ? uC0 : uC1);
          const unsigned c2 = CNT_INDEL + ((fo == f) ? lC0 : lC2);
          unsigned c0;
          if (dsc == m) c0 = dC0 + (eq ? CNT_MATCH : CNT_MIS);
          else c0 = (f >= e) ? c2 : c1;
          MA = m; EA = e; FA = f; C0A = c0; C1A = c1; C2A = c2;
        }
      }
    }
    // ---- column B = 2L+1: upper neighbour from lane L+1's A (this pair)
    int uM = rowDown1E(MA, negInf); unsigned uC0 = rowDown1E(C0A, 0u);
    int uE = 0; unsigned uC1 = 0;
    if (!PW) { uE = rowDown1E(EA, negInf); uC1 = rowDown1E(C1A, 0u); }
    if (L == 7) { uM = negInf; uC0 = 0; uE = negInf; uC1 = 0; }    // column 16 is never inside a band of <= 16 columns
    {
      const int dcol = 2 * L + 1, j = i - leftBand + dcol;
      if (rowOk && dcol < W && j >= 1 && j <= lent) {
        int bM = MA, bF = FA; unsigned bC0 = C0A, bC2 = C2A;       // left neighbour (i, 2L): own A of this pair
        int dM = oMB; unsigned dC0 = oC0B;
        if (i == 1 || j == 1) {
          if (j == 1) {
            bM = -4 - 4 * i;
            if (PW) bC0 = CNT_MATCH + CNT_INDEL * (unsigned)(i - 1);
            else { bF = -4 - 4 * i; bC0 = CNT_INDEL * (unsigned)i; bC2 = CNT_INDEL * (unsigned)(1 + i); }
          }
          if (i == 1) {
            uM = -4 - 4 * j;
            if (PW) uC0 = CNT_MATCH + CNT_INDEL * (unsigned)(j - 1);
            else { uE = e0; uC0 = CNT_INDEL * (unsigned)(j + (j > q4 ? 1 : 0)); uC1 = CNT_INDEL * (unsigned)(1 + j); }
            const int jj = j - 1;
            dM = jj == 0 ? 0 : -4 - 4 * jj;
            dC0 = jj == 0 ? 0u : (PW ? CNT_MATCH + CNT_INDEL * (unsigned)(jj - 1) : CNT_INDEL * (unsigned)(jj + (jj > q4 ? 1 : 0)));
          } else {
            dM = -4 - 4 * (i - 1);
            dC0 = PW ? CNT_MATCH + CNT_INDEL * (unsigned)(i - 2) : CNT_INDEL * (unsigned)(i - 1);
          }
        }
        if (PW) {
          const bool eq = baseEqualW((T4PW)tl[j - 1], pc);
          const int dsc = dM + (eq ? 2 : -2);
          int m = dsc;
          if (bM - 4 > m) m = bM - 4;
          if (uM - 4 > m) m = uM - 4;
          unsigned c;
          if (dsc == m) c = dC0 + (eq ? CNT_MATCH : CNT_MIS);
          else if (uM - 4 == m) c = uC0 + CNT_INDEL;
          else c = bC0 + CNT_INDEL;
          MB = m; C0B = c;
        } else {
          const char tc = tl[j - 1];
          const bool eq = (tc == pc || tc == 'N' || pc == 'N');
          int e = uE - 1, eo = uM - 5;
          if (eo > e) e = eo;
          int f = bF - 1, fo = bM - 5;
          if (fo > f) f = fo;
          const int dsc = dM + (eq ? 2 : -2);
          int m = dsc;
          if (e > m) m = e;
          if (f > m) m = f;
          const unsigned c1 = CNT_INDEL + ((eo == e) ? uC0 : uC1);
          const unsigned c2 = CNT_INDEL + ((fo == f) ? bC0 : bC2);
          unsigned c0;
          if (dsc == m) c0 = dC0 + (eq ? CNT_MATCH : CNT_MIS);
          else c0 = (f >= e) ? c2 : c1;
          MB = m; EB = e; FB = f; C0B = c0; C1B = c1; C2B = c2;
        }
      }
    }
  }
  int dF = lent - lenp + leftBand;
  if (dF < 0 || dF > 15) dF = 0;
  const unsigned finA = __shfl(C0A, grpBase + (dF >> 1)), finB = __shfl(C0B, grpBase + (dF >> 1));
  return run ? ((dF & 1) ? finB : finA) : result;
}

// Anchor walk of one overlap (SeqSet.hpp:1829-2019). One lane. The gap alignments themselves are hoisted out
// of this data-dependent loop (a wavefront would otherwise execute them one lane at a time):
//   collect == true : append one job (ordIdx | j << 16) per gap that needs an alignment
//   collect == false: consume the job results stored behind the chain and finish the overlap
__device__ void walkOverlap(const T4IndexView &ix, WaveMem &wm, WaveState *ws, OvRec &o, int ordIdx, bool collect) {
  const int K = ix.k;
  const bool isRef = (o.flags & OV_ISREF) != 0;
  const unsigned *hc = (const unsigned *)(wm.keys + o.chainPos);
  const unsigned *res = hc + o.chainLen;
  int matchCnt = 2 * K, indelCnt = 0;
  bool simOne = true;
  unsigned prevPair = hc[0];
  for (int j = 1; j < o.chainLen; ++j) {
    const unsigned curPair = hc[j];
    int pa = PA(prevPair), pb = PB(prevPair), qa = PA(curPair), qb = PB(curPair);
    prevPair = curPair;
    int doDP = 0;
    if (pb - pa == qb - qa) {
      if (pa + K - 1 >= qa) matchCnt += 2 * (qa - pa);
      else { matchCnt += 2 * K; doDP = 1; }
    } else {
      if (ix.radius == 0 || !isRef) { simOne = false; break; }
      if (pa + K - 1 >= qa && pb + K - 1 < qb) { matchCnt += 2 * (qa - pa); indelCnt += (qb - (pb + K) + (qa + K - pa)); }
      else if (pa + K - 1 < qa && pb + K - 1 >= qb) { matchCnt += 2 * (qb - pb); indelCnt += (qa - (pa + K) + (qb + K - pb)); }
      else if (pa + K - 1 >= qa && pb + K - 1 >= qb) {
        int da = qa - pa, db = qb - pb;
        matchCnt += 2 * (da < db ? da : db);
        int d = (qa - qb) - (pa - pb);
        indelCnt += d < 0 ? -d : d;
      } else { matchCnt += 2 * K; doDP = 2; }
    }
    if (doDP) {
      int lent = qb - (pb + K), lenp = qa - (pa + K);
      if (lent > ix.nomatchGapLimit || lenp > ix.nomatchGapLimit) { simOne = false; break; }
      if (collect) {
        int slot = atomicAdd(&ws->jobCount, 1);
        if (slot < wm.candCap) wm.cand[slot] = (unsigned)ordIdx | ((unsigned)j << 16); else ws->overflow = 1;
        continue;
      }
      unsigned c = res[j];
      if (c == DP_FAIL) { ws->unsupported = 1; simOne = false; break; }
      matchCnt += 2 * (int)(c & 1023u); indelCnt += (int)(c >> 20);
      if (doDP == 1) { if ((ix.radius == 0 || !isRef) && indelCnt > 0) { simOne = false; break; } }
      else { if (!isRef && indelCnt > 0) { simOne = false; break; } }
    }
  }
  if (collect) {
    // Reference-gene overlaps under radius > 0 never break on an alignment result (only on geometry, which this pass sees):
    // their matchCnt / indelCnt are plain sums, so the alignment results are ADDED by whoever computes them (res[0], which no
    // gap uses, and o.indelCnt) and the finishing pass does not walk the chain again.
    if (isRef && ix.radius > 0) {
      ((unsigned *)res)[0] = (unsigned)matchCnt;
      OvRec &dst = wm.ov[wm.ord[ordIdx]];
      dst.indelCnt = indelCnt;
      if (simOne) dst.flags &= ~OV_SIMZERO; else dst.flags |= OV_SIMZERO;
    }
    return;
  }
  o.matchCnt = matchCnt; o.indelCnt = indelCnt;
  if (simOne) o.flags &= ~OV_SIMZERO; else o.flags |= OV_SIMZERO;
  if (lowComplex(wm, (o.flags & OV_PLUS) != 0, o.rs, o.re)) o.flags |= OV_SIMZERO;
}

#if T4_OPT_ROWWALK
// The collecting walk of every kept overlap, one 16-lane row per overlap (experiment; out of line so that its registers do
// not add to the kernel's pressure). For a reference-gene overlap under radius > 0 the walk only ever stops on geometry (a
// gap beyond nomatchGapLimit), so its sums are prefix sums up to that anchor pair j*: find j*, add the pairs before it (pair
// j* itself has added its 2K before the reference looks at the gap), list their gap jobs.
// ALL: the row walk covers the other class as well when the list is short (the AddRead / AssignRead kernels).
template <bool ALL>
__device__ T4_NI void walkOverlapsRows(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int overlapCnt) {
  const int lane = tid(), NT = nthr(), K = ix.k;
  const int row = lane >> 4, rl = lane & 15, nRows = NT >> 4;
  // the other class (novel sequences -- every contig of an AddRead query -- or radius 0) stops on alignment geometry too:
  // the data-dependent walk, one lane per overlap (a row per overlap left 15 of 16 lanes idle through thousands of them)
  // -- unless the list is short: then a row per overlap walks its chain sixteen pairs at a time here too. That walk stops at
  // the first pair off the diagonal or whose gap is beyond nomatchGapLimit, and lists the gaps before it.
  const bool rowsForAll = ALL && overlapCnt <= 4 * nRows;
  int anyFast = 0;
  if (ALL && !rowsForAll) {
    for (int i = lane; i < overlapCnt; i += NT) {
      OvRec oc = wm.ov[wm.ord[i]];
      if ((oc.flags & OV_ISREF) != 0 && ix.radius > 0) anyFast = 1;
      else walkOverlap(ix, wm, ws, oc, i, true);
    }
    if (blockSum(anyFast, ws->red) == 0) return;
  }
  for (int i0 = 0; i0 < overlapCnt; i0 += nRows) {
    const int i = i0 + row;
    const bool has = i < overlapCnt;
    int chainPos = 0, chainLen = 0, ovSlot = 0;
    bool fast = false;
    if (has) {
      ovSlot = wm.ord[i];
      const OvRec &o = wm.ov[ovSlot];
      chainPos = o.chainPos; chainLen = o.chainLen;
      fast = (o.flags & OV_ISREF) != 0 && ix.radius > 0;
      if (!ALL && !fast && rl == 0) { OvRec oc = o; walkOverlap(ix, wm, ws, oc, i, true); }   // (rough annotation: the rare novel overlap of a mixed set)
    }
    const unsigned *hc = (const unsigned *)(wm.keys + chainPos);
    int jStop = 0x7FFFFFFF;
    const bool slow = ALL && has && !fast && rowsForAll;
    if (slow)
      for (int j = 1 + rl; j < chainLen; j += 16) {
        const unsigned pp = hc[j - 1], cp = hc[j];
        const int pa = PA(pp), pb = PB(pp), qa = PA(cp), qb = PB(cp);
        if (pb - pa != qb - qa || (pa + K - 1 < qa && (qb - (pb + K) > ix.nomatchGapLimit || qa - (pa + K) > ix.nomatchGapLimit))) { jStop = j; break; }
      }
    if (fast)
      for (int j = 1 + rl; j < chainLen; j += 16) {
        const unsigned pp = hc[j - 1], cp = hc[j];
        const int pa = PA(pp), pb = PB(pp), qa = PA(cp), qb = PB(cp);
        const bool gapA = pa + K - 1 < qa, gapB = pb + K - 1 < qb;
        const bool doDP = (pb - pa == qb - qa) ? gapA : (gapA && gapB);
        if (doDP && (qb - (pb + K) > ix.nomatchGapLimit || qa - (pa + K) > ix.nomatchGapLimit)) { jStop = j; break; }
      }
    for (int m = 1; m < 16; m <<= 1) { const int other = __shfl_xor(jStop, m); jStop = other < jStop ? other : jStop; }
    int matchCnt = 0, indelCnt = 0;
    if (slow)
      for (int j = 1 + rl; j < chainLen && j < jStop; j += 16) {
        if (PA(hc[j - 1]) + K - 1 < PA(hc[j])) {
          const int slot = atomicAdd(&ws->jobCount, 1);
          if (slot < wm.candCap) wm.cand[slot] = (unsigned)i | ((unsigned)j << 16); else ws->overflow = 1;
        }
      }
    if (fast)
      for (int j = 1 + rl; j < chainLen && j <= jStop; j += 16) {
        if (j == jStop) { matchCnt += 2 * K; break; }
        const unsigned pp = hc[j - 1], cp = hc[j];
        const int pa = PA(pp), pb = PB(pp), qa = PA(cp), qb = PB(cp);
        bool doDP = false;
        if (pb - pa == qb - qa) {
          if (pa + K - 1 >= qa) matchCnt += 2 * (qa - pa);
          else { matchCnt += 2 * K; doDP = true; }
        } else if (pa + K - 1 >= qa && pb + K - 1 < qb) { matchCnt += 2 * (qa - pa); indelCnt += (qb - (pb + K) + (qa + K - pa)); }
        else if (pa + K - 1 < qa && pb + K - 1 >= qb) { matchCnt += 2 * (qb - pb); indelCnt += (qa - (pa + K) + (qb + K - pb)); }
        else if (pa + K - 1 >= qa && pb + K - 1 >= qb) {
          const int da = qa - pa, db = qb - pb;
          matchCnt += 2 * (da < db ? da : db);
          const int d = (qa - qb) - (pa - pb);
          indelCnt += d < 0 ? -d : d;
        } else { matchCnt += 2 * K; doDP = true; }
        if (doDP) {
          const int slot = atomicAdd(&ws->jobCount, 1);
          if (slot < wm.candCap) wm.cand[slot] = (unsigned)i | ((unsigned)j << 16); else ws->overflow = 1;
        }
      }
    matchCnt = rowSum16(matchCnt);
    indelCnt = rowSum16(indelCnt);
    if (fast && rl == 0) {
      ((unsigned *)hc)[chainLen] = (unsigned)(matchCnt + 2 * K);
      OvRec &dst = wm.ov[ovSlot];
      dst.indelCnt = indelCnt;
      if (jStop == 0x7FFFFFFF) dst.flags &= ~OV_SIMZERO; else dst.flags |= OV_SIMZERO;
    }
  }
}

// The finishing pass of the same class for a short list, a row per overlap: with the gap results in place the sequential walk
// (walkOverlap, collect == false) stops at the first pair that is off the diagonal, has a gap beyond nomatchGapLimit, has a
// failed alignment or an alignment with indels; its sums are those of the pairs before that one plus what that pair had added
// before the stop. Reference-gene overlaps under radius > 0 take their sums as the lane-per-overlap loop does.
__device__ T4_NI void finishOverlapsRows(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int overlapCnt) {
  const int lane = tid(), NT = nthr(), K = ix.k;
  const int row = lane >> 4, rl = lane & 15, nRows = NT >> 4;
  for (int i0 = 0; i0 < overlapCnt; i0 += nRows) {
    const int i = i0 + row;
    const bool has = i < overlapCnt;
    OvRec o = wm.ov[wm.ord[has ? i : 0]];
    const bool fast = (o.flags & OV_ISREF) != 0 && ix.radius > 0;
    const int runSize = (o.flags & OV_ISREF) ? 0 : o.indelCnt;   // (chainFinish left it there)
    const unsigned *hc = (const unsigned *)(wm.keys + o.chainPos);
    const unsigned *res = hc + o.chainLen;
    int jBreak = 0x7FFFFFFF, add = 0, ind = 0, failed = 0;
    if (has && !fast)
      for (int j = 1 + rl; j < o.chainLen; j += 16) {
        const unsigned pp = hc[j - 1], cp = hc[j];
        const int pa = PA(pp), pb = PB(pp), qa = PA(cp), qb = PB(cp);
        bool stop = false;
        int a = 0;
        if (pb - pa != qb - qa) stop = true;
        else if (pa + K - 1 >= qa) a = 2 * (qa - pa);
        else {
          a = 2 * K;
          if (qb - (pb + K) > ix.nomatchGapLimit || qa - (pa + K) > ix.nomatchGapLimit) stop = true;
          else {
            const unsigned c = res[j];
            if (c == DP_FAIL) { stop = true; failed = 1; }
            else { a += 2 * (int)(c & 1023u); if (c >> 20) { stop = true; ind = (int)(c >> 20); } }
          }
        }
        if (stop) { jBreak = j; add += a; break; }   // a lane's later pairs lie beyond the stop: not walked
        add += a;
      }
    int jb = jBreak;
    for (int m = 1; m < 16; m <<= 1) { const int other = __shfl_xor(jb, m); jb = other < jb ? other : jb; }
    // what the lanes added for pairs beyond the row's stop does not count
    if (has && !fast && jb != 0x7FFFFFFF) {
      add = 0; 
      for (int j = 1 + rl; j < o.chainLen && j <= jb; j += 16) {
        const unsigned pp = hc[j - 1], cp = hc[j];
        const int pa = PA(pp), pb = PB(pp), qa = PA(cp), qb = PB(cp);
        if (pb - pa != qb - qa) continue;                      // only the stop itself can be off the diagonal: adds nothing
        if (pa + K - 1 >= qa) { add += 2 * (qa - pa); continue; }
        add += 2 * K;
        if (qb - (pb + K) > ix.nomatchGapLimit || qa - (pa + K) > ix.nomatchGapLimit) continue;
        const unsigned c = res[j];
        if (c != DP_FAIL) add += 2 * (int)(c & 1023u);
      }
      if (jBreak != jb) { ind = 0; failed = 0; }
    }
    add = rowSum16(add);
    ind = rowSum16(ind);
    failed = rowSum16(failed);
    if (has && rl == 0) {
      const int m0 = o.matchCnt;
      if (fast) {
        o.matchCnt = (int)res[0];
        if (lowComplex(wm, (o.flags & OV_PLUS) != 0, o.rs, o.re)) o.flags |= OV_SIMZERO;
      } else {
        if (failed) ws->unsupported = 1;
        o.matchCnt = 2 * K + add; o.indelCnt = ind;
        if (jb == 0x7FFFFFFF) o.flags &= ~OV_SIMZERO; else o.flags |= OV_SIMZERO;
        if (lowComplex(wm, (o.flags & OV_PLUS) != 0, o.rs, o.re)) o.flags |= OV_SIMZERO;
      }
#ifdef T4_PATHDBG
      if (!fast) {   // cross-check against the sequential walk
        OvRec chk = wm.ov[wm.ord[i]];
        walkOverlap(ix, wm, ws, chk, i, false);
        const int kind = jb == 0x7FFFFFFF ? 0 : failed ? 3 : ind ? 2 : 1;
        printf("PATHDBG finishrows kind %d %s\n", kind, (chk.matchCnt == o.matchCnt && chk.indelCnt == o.indelCnt && chk.flags == o.flags) ? "ok" : "MISMATCH");
      }
#endif
      o.chainLen = m0;
      if (!(o.flags & OV_ISREF)) ovKeepRunSize(o, runSize);   // (the chain is done with: its index makes room)
      wm.ov[wm.ord[i]] = o;
    }
  }
}
#endif

// result of one gap alignment of overlap slot `ovSlot` (fast class of walkOverlap): add it where the finishing pass expects it
__device__ __forceinline__ void addGapResult(WaveMem &wm, WaveState *ws, int ovSlot, unsigned c) {
  OvRec &o = wm.ov[ovSlot];
  if (c == DP_FAIL) { ws->unsupported = 1; return; }
  unsigned *res0 = (unsigned *)(wm.keys + o.chainPos) + o.chainLen;
  atomicAdd(res0, 2u * (c & 1023u));
  if (c >> 20) atomicAdd(&o.indelCnt, (int)(c >> 20));
}

// Quick exits of one gap job (any lane); jobs that need the banded DP are marked DP_PENDING.
__device__ void runGapJobQuick(const T4IndexView &ix, WaveMem &wm, WaveState *ws, unsigned job) {
  const int K = ix.k;
  const OvRec &o = wm.ov[wm.ord[job & 0xFFFF]];
  const int j = (int)(job >> 16);
  unsigned *hc = (unsigned *)(wm.keys + o.chainPos);
  const int pa = PA(hc[j - 1]), pb = PB(hc[j - 1]), qa = PA(hc[j]), qb = PB(hc[j]);
  const int lent = qb - (pb + K), lenp = qa - (pa + K);
  const char *r = ((o.flags & OV_PLUS) ? wm.seg : wm.rc) + pa + K;
  const T4SeqInfo si = ix.seqs[o.seqIdx];
  unsigned cnt;
  bool done = si.isRef ? dpAffineQuick(ix.cons + si.consOff + pb + K, lent, r, lenp, cnt)
                       : dpPosWeightQuick(ix.pw + si.pwOff + pb + K, lent, r, lenp, cnt);
  hc[o.chainLen + j] = done ? cnt : DP_PENDING;
  if (done && si.isRef && ix.radius > 0) addGapResult(wm, ws, wm.ord[job & 0xFFFF], cnt);
}

// GetVJOverlapsFromHits' pair selection (SeqSet.hpp:1093-1160) on wm.ov[0..n). One lane. Returns 0 or 2
// and leaves the chosen pair in wm.ov[0], wm.ov[1].
__device__ int selectVJPair(const T4IndexView &ix, WaveMem &wm, int n) {
  int maxMatch = 0, tagi = 0, tagj = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      T4SeqInfo a = ix.seqs[wm.ov[i].seqIdx], b = ix.seqs[wm.ov[j].seqIdx];
      if (a.name0 != b.name0 || a.name1 != b.name1 || a.name2 != b.name2 || a.name3 == b.name3) continue;
      if (a.name3 == 'V') { if (wm.ov[i].rs > wm.ov[j].rs) continue; }
      else { if (wm.ov[i].rs < wm.ov[j].rs) continue; }
      if (wm.ov[i].matchCnt + wm.ov[j].matchCnt > maxMatch) { maxMatch = wm.ov[i].matchCnt + wm.ov[j].matchCnt; tagi = i; tagj = j; }
    }
  if (maxMatch == 0) return 0;
  OvRec a = wm.ov[tagi], b = wm.ov[tagj];
  wm.ov[0] = a; wm.ov[1] = b;
  return 2;
}

// One GetHitsFromRead + GetOverlapsFromHits pass over the current segment. Returns H (hit records
// emitted by the seed stage) or -1 on capacity overflow. Overlaps are left in wm.ov / ws->ovCount.
template <bool NOVEL>   // NOVEL: the kernel variants that meet contig sets (repeat-skip rule live); the reference-set variants keep their register budget
__device__ int seedChainPass(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int segLen, int strandArg, int barcode,
                             bool allowTotalSkip, bool vjOnly, int hitLenRequired, int filter, int onlySeq = -1) {
  const int lane = tid(), NT = nthr();
  unsigned *posStart = (unsigned *)wm.ov;   // dead before the first overlap record is written
  unsigned *posPref = wm.pairs;             // dead before the first pair is written
  const int nk = segLen - ix.k + 1;
  if (lane == 0) ws->ovCount = 0;
  PHASE_MARK(ws, 1);
  int H = NOVEL ? seedPositionsNovel(ix, wm, segLen, strandArg, barcode, allowTotalSkip, posStart, posPref, ws->red, wm.keys, ws)
                : seedPositions(ix, wm, segLen, strandArg, barcode, allowTotalSkip, posStart, posPref, ws->red);   // the key array is free until the hits are expanded
  if (NOVEL && onlySeq >= 0) {   // restricted re-query: the hits with one contig (a few hundred at most), whatever the read's total
    int Hv = -2;
    if (ws->useMarks) { __syncthreads(); Hv = expandHitsContig(ix, wm, ws, nk, onlySeq, posPref); }
    const bool offContig = Hv != -2;
    if (!offContig) Hv = expandHitsOnly(ix, wm, ws, nk, H, onlySeq, posStart, posPref);
    if (Hv < 0) return -1;
    if (offContig) H = Hv;   // (the postings this pass looked at)
    if (Hv > 1) { if (wm.ldsArrays) bitonicSortRegLds(wm.keys, Hv); else bitonicSort(wm.keys, Hv); }
    __syncthreads();
    // the contig's two groups (true sizes): what the caller's bookkeeping of the group statistics needs (T4QueryArgs::stats8)
    int nPlus = 0;
    for (int i = lane; i < Hv; i += NT) nPlus += (int)(wm.keys[i] >> 63);
    nPlus = blockSum(nPlus, ws->red);
    // ... and the stretch of the contig the read lies on along every diagonal with three or more hits (the entry's dependency record
    // of this contig: what t4_assembler::rebuildGroup derived from the consensus, a superset, on the host)
    if (lane == 0) { ws->hullLo[0] = ws->hullLo[1] = 0x7FFFFFFF; ws->hullHi[0] = ws->hullHi[1] = -0x7FFFFFFF; }
    __syncthreads();
    for (int i = lane; i < Hv; i += NT) {
      const unsigned long long ki = wm.keys[i];
      const unsigned long long dg = ki >> T4_B_BITS;   // (strand, contig, diagonal)
      if (i > 0 && (wm.keys[i - 1] >> T4_B_BITS) == dg) continue;   // not the first hit of its diagonal
      if (i + 2 < Hv && (wm.keys[i + 2] >> T4_B_BITS) == dg) {
        const int at = -(KEY_C(ki) - T4_C_BIAS), st = KEY_PLUS(ki);
        atomicMin(&ws->hullLo[st], at);
        atomicMax(&ws->hullHi[st], at + segLen - 1);
      }
    }
    __syncthreads();
    overlapsFromKeys(ix, wm, ws, Hv, hitLenRequired, 0);   // (filter 0: the thresholds are the caller's -- ws->forceMin, else three hits: it has made sure the group statistics leave them there)
    if (lane == 0) {
      const int g[2] = {Hv - nPlus, nPlus};
      for (int t = 0; t < 2; ++t) { ws->nvN4[t] = g[t] >= 4 ? 1 : 0; ws->nvN5[t] = g[t] >= 5 ? 1 : 0; ws->nvSmax[t] = g[t]; }
    }
    __syncthreads();
    return H;
  }
  if (NOVEL && !allowTotalSkip && !vjOnly && filter == 1 && H > 10000) {
    // a posting list beyond 10000 entries switches on removeOnlyRepeats and the run-relative repeats test (SeqSet.hpp:802, 876,
    // 934-940): the wide query replays them, the single-workgroup tiers do not
    int huge = 0;
    for (int q = lane; q < 2 * nk; q += NT) if (posPref[q + 1] - posPref[q] > 10000u) huge = 1;
    if (blockSum(huge, ws->red)) { if (ws->wideWant) return -3; if (lane == 0) ws->unsupported = 1; }
  }
  if (NOVEL && ws->wideWant && H > ws->wideWant && !allowTotalSkip && !vjOnly && filter == 1) return -3;   // heavy enough for the wide query (t4_wide.h)
  if (H > wm.hitLimit) return (NOVEL && ws->wideWant && !wm.ldsArrays && !allowTotalSkip && !vjOnly && filter == 1) ? -3 : -1;
  PHASE_MARK(ws, 2);
#if T4_OPT_KEY32
  const int k32 = ix.key32;
#else
  const int k32 = 0;
#endif
  int Hv = expandHits(ix, wm, nk, H, barcode, vjOnly, posStart, posPref, ws->red, k32);
  __syncthreads();
  PHASE_MARK(ws, 3);
  if (k32) {
    // half the LDS traffic of the sort; the keys are widened to the 64-bit layout the later stages read (through pairs,
    // whose prefix sums died with expandHits: element i of the wide array overlays elements 2i, 2i + 1 of the narrow one)
    unsigned *k32v = (unsigned *)wm.keys;
#if T4_OPT_REGSORT
    if (H > 1) { if (NOVEL && wm.ldsArrays) bitonicSortRegLds<unsigned>(k32v, H); else bitonicSort32(k32v, H); }
#else
    if (H > 1) bitonicSort(k32v, H);
#endif
    for (int i = lane; i < H; i += NT) wm.pairs[i] = k32v[i];
    __syncthreads();
    const int cBits = (k32 >> 8) & 255, bias32 = (1 << cBits) - 512;
    for (int i = lane; i < H; i += NT) {
      const unsigned v = wm.pairs[i];
      unsigned long long key = ~0ull;
      if (v != ~0u) {
        const int a = (int)(v & 511u), c = (int)((v >> 9) & ((1u << cBits) - 1u)) - bias32;
        const unsigned long long idx = (v & 0x7FFFFFFFu) >> (cBits + 9);
        key = ((unsigned long long)(v >> 31) << 63) | (idx << (T4_C_BITS + T4_B_BITS)) |
              ((unsigned long long)(c + T4_C_BIAS) << T4_B_BITS) | (unsigned long long)(a - c);
      }
      wm.keys[i] = key;
    }
    __syncthreads();
  } else if (H > 1) {
    if (!wm.ldsArrays && wm.ldsSort) bitonicSortBlocked(wm.keys, H, wm.ldsSort, wm.ldsSortCap);
#if T4_OPT_REGSORT64
    else if (NOVEL && wm.ldsArrays) bitonicSortRegLds(wm.keys, H);   // 64-bit keys of a big set
    else if (wm.ldsArrays) bitonicSortReg(wm.keys, H);   // chunk-local sub-steps in registers as for the 32-bit keys
#endif
    else bitonicSort(wm.keys, H);
  }
  PHASE_MARK(ws, 4);
  overlapsFromKeys(ix, wm, ws, Hv, hitLenRequired, filter);
  PHASE_MARK(ws, 0);
  return H;
}

// std::sort(overlaps) for thousands of overlaps (a read inside a gene segment that every contig carries; the rank sort of
// overlapsFromSegment is quadratic). The first four criteria of _overlap::operator< (matchCnt desc, read span desc, seqIdx,
// strand) fit 45 bits: sort (those | index) with the hit sorter, then settle the rare groups that tie on all four with the full
// comparison. Returns false (nothing done) when a field does not fit. Out of line: its registers are its own.
__device__ T4_NI bool sortOverlapsByKey(WaveMem &wm, WaveState *ws, int overlapCnt) {
  const int lane = tid(), NT = nthr();
  bool keySorted = false;
  if (lane == 0) ws->sortBad = 0;
  __syncthreads();
  unsigned long long *sk = overlapCnt <= wm.ldsSortCap ? wm.ldsSort : (unsigned long long *)wm.pairs;   // pairs: dead until the DPs
  for (int i = lane; i < overlapCnt; i += NT) {
    const OvRec o = wm.ov[i];
    const int span = o.re - o.rs;
    if (o.matchCnt < 0 || o.matchCnt > 4095 || span < 0 || span > 1023) ws->sortBad = 1;
    sk[i] = ((unsigned long long)(4095 - (o.matchCnt & 4095)) << 47) | ((unsigned long long)(1023 - (span & 1023)) << 37) |
            ((unsigned long long)(unsigned)o.seqIdx << 15) | ((unsigned long long)((o.flags & OV_PLUS) ? 1 : 0) << 14) |
            (unsigned long long)i;
  }
  __syncthreads();
  if (!ws->sortBad) {
    if (overlapCnt <= wm.ldsSortCap) bitonicSortReg(sk, overlapCnt);
    else bitonicSortBlocked(sk, overlapCnt, wm.ldsSort, wm.ldsSortCap);
    __syncthreads();
    for (int p = lane; p < overlapCnt; p += NT) {
      const unsigned long long key = sk[p], pre = key >> 14;
      const int i = (int)(key & 16383);
      int gs = p, ge = p + 1;
      while (gs > 0 && (sk[gs - 1] >> 14) == pre) --gs;
      while (ge < overlapCnt && (sk[ge] >> 14) == pre) ++ge;
      int rank = 0;
      if (ge - gs > 1) {
        const OvRec me = wm.ov[i];
        for (int q = gs; q < ge; ++q) {
          const int j = (int)(sk[q] & 16383);
          if (j == i) continue;
          const int cm = ovCmp(wm.ov[j], me, false);
          if (cm < 0 || (cm == 0 && j < i)) ++rank;
        }
      }
      wm.ord[gs + rank] = (unsigned short)i;
#ifdef T4_PATHDBG
      if (ge - gs > 1 && p == gs) printf("PATHDBG keysort tie group %d of %d\n", ge - gs, overlapCnt);
      if (p == 0) printf("PATHDBG keysort n %d %s\n", overlapCnt, overlapCnt <= wm.ldsSortCap ? "lds" : "blocked");
#endif
    }
    keySorted = true;
  }
  __syncthreads();
  return keySorted;
}

// The fast pre-filters of GetOverlapsFromRead against the best novel overlap so far (SeqSet.hpp:1705-1794), on the scored list
// wm.ov[wm.ord[0 .. overlapCnt)].
__device__ T4_PREFILTER_NI void prefilterNovel(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int overlapCnt, int segLen) {
  const int lane = tid(), NT = nthr();
  // the fast pre-filters against the best novel overlap (SeqSet.hpp:1705-1794) are order dependent: overlap i is judged
  // against the best scored novel overlap among 0 .. i-1. That best changes a handful of times over thousands of overlaps,
  // so the list is replayed a workgroup-wide chunk at a time: every undecided overlap is judged against the current best,
  // the first one that would replace it is found, everything before it is final, everything after it is judged again.
  int best = -1;
  OvRec bn = wm.ov[wm.ord[0]];
  const int len = segLen;
  for (int i0 = 0; i0 < overlapCnt; i0 += NT) {
    const int i = i0 + lane;
    const bool has = i < overlapCnt;
    const int slot = has ? wm.ord[i] : 0;
    OvRec o = wm.ov[slot];
    const bool novel = has && !(o.flags & OV_ISREF);
    int from = i0;
    for (;;) {
      bool cut = false, cand = false;
      if (novel && i >= from) {
        const int m0 = o.chainLen;
        if (best != -1) {
          const double bs = ovSim(bn);
          if (bn.rs == 0 && bn.re == len - 1) {
            if (bs == 1) cut = true;
            else if (bs > ix.repeatSim && m0 < 0.9 * bn.matchCnt) cut = true;
          }
          if (!cut && bn.rs + len - 1 - bn.re < ix.radius) {
            if (bs == 1 && m0 < 0.9 * bn.matchCnt) cut = true;
            else if (bs > ix.repeatSim && m0 < 0.8 * bn.matchCnt) cut = true;
          }
          if (!cut && o.ss - o.rs >= ix.radius && o.se + (len - 1 - o.re) + ix.radius < ix.seqs[o.seqIdx].len &&
              bn.matchCnt > 0.97 * (2 * len) && bs > ix.repeatSim && m0 < 0.9 * bn.matchCnt) cut = true;
          if (!cut && m0 < 0.4 * bn.matchCnt) cut = true;
          if (!cut && overlapCnt > 1000 && m0 < 0.9 * bn.matchCnt) cut = true;
        }
        if (!cut && !(o.flags & OV_SIMZERO) && ovSim(o) > 0 && (best == -1 || ovLess(o, bn, true))) cand = true;
      }
      if (lane == 0) ws->red[0] = 0x7FFFFFFF;
      __syncthreads();
      if (cand) atomicMin(&ws->red[0], i);
      __syncthreads();
      const int f = ws->red[0];
      if (cut && i < f) {   // final: the best it was judged against is the one the sequential pass would have used
        ovCutKeepScored(o);
        wm.ov[slot] = o;
        from = 0x7FFFFFFF;
      }
      __syncthreads();
      if (f == 0x7FFFFFFF) break;
#ifdef T4_PATHDBG
      if (lane == 0) printf("PATHDBG prefilter best %d -> %d of %d\n", best, f, overlapCnt);
#endif
      best = f;
      bn = wm.ov[wm.ord[f]];
      from = from == 0x7FFFFFFF ? from : f + 1;
    }
  }
  __syncthreads();
}

// The scoring of GetOverlapsFromRead (SeqSet.hpp:1673-2094 without its pre-filters, which are replayed afterwards): every overlap
// wm.ov[wm.ord[0 .. overlapCnt)] walks its chain, the gap alignments run eight (or one) per wavefront, the overlaps take their
// matchCnt / indelCnt / similarity-zero flag; chainLen is left holding the matchCnt of GetOverlapsFromHits. false: a capacity ran out.
template <bool ROWS>
__device__ __forceinline__ bool scoreOverlaps(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int overlapCnt, DPScratch sc) {
  const int lane = tid(), NT = nthr();
  // score every kept overlap (one lane each)
  // (1) collect the gap-alignment jobs of every kept overlap (job list = the dead cand array)
  if (lane == 0) ws->jobCount = 0;
  __syncthreads();
#if T4_OPT_ROWWALK
  walkOverlapsRows<ROWS>(ix, wm, ws, overlapCnt);
#else
  for (int i = lane; i < overlapCnt; i += NT) {
    OvRec o = wm.ov[wm.ord[i]];
    walkOverlap(ix, wm, ws, o, i, true);
  }
#endif
  __syncthreads();
  if (ws->overflow) return false;
  const int nJobs = ws->jobCount;
  if (lane == 0) DBG_ADD(0, nJobs);
  // (2) quick exits, one job per lane
  PHASE_MARK(ws, 13);
  for (int q = lane; q < nJobs; q += NT) runGapJobQuick(ix, wm, ws, wm.cand[q]);
  __syncthreads();
  // (3) banded DPs: compact the pending jobs, then one wavefront per alignment
  PHASE_MARK(ws, 14);
  int nPend = 0;
  for (int q0 = 0; q0 < nJobs; q0 += NT) {
    int q = q0 + lane;
    unsigned job = q < nJobs ? wm.cand[q] : 0u;
    bool pend = false;
    if (q < nJobs) {
      const OvRec &o = wm.ov[wm.ord[job & 0xFFFF]];
      pend = ((const unsigned *)(wm.keys + o.chainPos))[o.chainLen + (job >> 16)] == DP_PENDING;
    }
    int tot;
    int inc = blockInclScan(pend ? 1 : 0, ws->red, tot);   // barriers inside: the chunk's jobs are all read
    if (pend) wm.cand[nPend + inc - 1] = job;               // target <= q: in-place, chunk by chunk
    nPend += tot;
    __syncthreads();
  }
#if T4_OPT_JOBSORT
  // Longest jobs first: a wavefront steps its eight alignments together, i.e. for as long as the longest of them, so batches
  // of similar length waste the fewest steps (results are sums per overlap: the order of the jobs is not observable).
  if (nPend > 8 && nPend <= NT) {
    unsigned job = 0u;
    if (lane < nPend) {
      job = wm.cand[lane];
      const OvRec &o = wm.ov[wm.ord[job & 0xFFFF]];
      const unsigned *hc = (const unsigned *)(wm.keys + o.chainPos);
      const int jj = (int)(job >> 16);
      const int lenp = PA(hc[jj]) - (PA(hc[jj - 1]) + ix.k), lent = PB(hc[jj]) - (PB(hc[jj - 1]) + ix.k);
      const int steps = lenp + ((10 + (lent > lenp ? lent - lenp : lenp - lent)) >> 1);
      wm.pairs[lane] = ((unsigned)steps << 16) | (unsigned)(0xFFFF - lane);
    }
    __syncthreads();
    if (lane < nPend) {
      const unsigned mine = wm.pairs[lane];
      int rank = 0;
      for (int j = 0; j < nPend; ++j) rank += wm.pairs[j] > mine ? 1 : 0;
      wm.cand[rank] = job;
    }
    __syncthreads();
  }
#endif
  {
    const int wave = lane >> 6, nw = NT >> 6, grp = (lane >> 3) & 7;
    // target buffers in LDS, one slice per wave, cut into the eight groups' buffers: pairs is dead here (cap >= 1024 ints); the
    // global-scratch tier, whose pairs are global memory, takes the LDS block the sorts staged through
    char *const tbufAll = wm.ldsArrays ? (char *)wm.pairs : (char *)wm.dirBuf;
    const int rowCap = (((wm.ldsArrays ? wm.cap * 4 : wm.dirBytes)) / (nw * 8)) & ~15;
    char *tbufWave = tbufAll + wave * 8 * rowCap;
    // (3a) eight alignments per wavefront, one per group of 8 lanes (bands of <= 16 columns: almost all of them)
    for (int q0 = wave * 8; q0 < nPend; q0 += nw * 8) {
      const int q = q0 + grp;
      const bool has = q < nPend;
      const unsigned job = has ? wm.cand[q] : 0u;
      const OvRec o = wm.ov[wm.ord[job & 0xFFFF]];
      const int jj = has ? (int)(job >> 16) : 1;
      unsigned *hc = (unsigned *)(wm.keys + o.chainPos);
      int lent = 0, lenp = 0, pa = 0, pb = 0;
      if (has) {
        pa = PA(hc[jj - 1]); pb = PB(hc[jj - 1]);
        const int qa = PA(hc[jj]), qb = PB(hc[jj]);
        lent = qb - (pb + ix.k); lenp = qa - (pa + ix.k);
      }
      const char *r = ((o.flags & OV_PLUS) ? wm.seg : wm.rc) + pa + ix.k;
      const T4SeqInfo si = ix.seqs[o.seqIdx];
      const bool asRef = has && si.isRef, asPw = has && !si.isRef;
      unsigned c = DP_FAIL;
      if (__any(asRef)) {
        const char *tq = ix.cons + si.consOff + pb + ix.k;
        unsigned v = dpOct<false, true>(asRef, tq, (const T4PW *)0, lent, r, lenp, tbufWave + grp * rowCap, rowCap);
        if (asRef) c = v;
      }
      if (__any(asPw)) {
        const T4PW *wq = ix.pw + si.pwOff + pb + ix.k;
        unsigned v = dpOct<true, true>(asPw, (const char *)0, wq, lent, r, lenp, tbufWave + grp * rowCap, rowCap);
        if (asPw) c = v;
      }
      if (has && (lane & 7) == 0) {
        DBG_ADD(1, 1);
        if (c != DP_FAIL) {
          hc[o.chainLen + jj] = c;
          if (si.isRef && ix.radius > 0) addGapResult(wm, ws, wm.ord[job & 0xFFFF], c);
          DBG_ADD(2, 2 * lenp + 11 + (lent > lenp ? lent - lenp : lenp - lent));
        }
      }
    }
    __syncthreads();
    // (3b) the wider bands: one wavefront per alignment
    char *tbuf = tbufWave;
    for (int q = wave; q < nPend; q += nw) {
      const unsigned job = wm.cand[q];
      const OvRec &o = wm.ov[wm.ord[job & 0xFFFF]];
      const int jj = (int)(job >> 16);
      unsigned *hc = (unsigned *)(wm.keys + o.chainPos);
      if (hc[o.chainLen + jj] != DP_PENDING) continue;   // wave-uniform
      const int pa = PA(hc[jj - 1]), pb = PB(hc[jj - 1]), qa = PA(hc[jj]), qb = PB(hc[jj]);
      const int lent = qb - (pb + ix.k), lenp = qa - (pa + ix.k);
      const char *r = ((o.flags & OV_PLUS) ? wm.seg : wm.rc) + pa + ix.k;
      const T4SeqInfo si = ix.seqs[o.seqIdx];
      unsigned c = DP_FAIL;
      if (lent <= 8 * rowCap)
        c = si.isRef ? dpWave<false>(ix.cons + si.consOff + pb + ix.k, (const T4PW *)0, lent, r, lenp, tbuf)
                     : dpWave<true>((const char *)0, ix.pw + si.pwOff + pb + ix.k, lent, r, lenp, tbuf);
      if (c == DP_FAIL && laneId() == 0) {   // band wider than a wavefront: lane-serial scratch version
        DBG_ADD(3, 1); DBG_ADD(4, (long long)lent * lenp);
        int c0, c1, c2;
        bool ok = si.isRef ? dpAffine(ix.cons + si.consOff + pb + ix.k, lent, r, lenp, sc, laneId(), c0, c1, c2)
                           : dpPosWeight(ix.pw + si.pwOff + pb + ix.k, lent, r, lenp, sc, laneId(), c0, c1, c2, (signed char *)0);
        if (ok) c = CNT_MATCH * (unsigned)c0 + CNT_MIS * (unsigned)c1 + CNT_INDEL * (unsigned)c2;
      }
      if (laneId() == 0) {
        hc[o.chainLen + jj] = c;
        if (si.isRef && ix.radius > 0) addGapResult(wm, ws, wm.ord[job & 0xFFFF], c);
      }
    }
  }
  __syncthreads();
  // (4) finish the overlaps
  PHASE_MARK(ws, 15);
#if T4_OPT_ROWWALK
  if (ROWS && overlapCnt <= 4 * (NT >> 4)) finishOverlapsRows(ix, wm, ws, overlapCnt);
  else
#endif
  for (int i = lane; i < overlapCnt; i += NT) {
    OvRec o = wm.ov[wm.ord[i]];
    int m0 = o.matchCnt;
    const int runSize = (o.flags & OV_ISREF) ? 0 : o.indelCnt;   // (chainFinish left it there)
    if ((o.flags & OV_ISREF) && ix.radius > 0) {   // sums are complete (see walkOverlap): no second walk
      o.matchCnt = (int)((const unsigned *)(wm.keys + o.chainPos))[o.chainLen];
      if (lowComplex(wm, (o.flags & OV_PLUS) != 0, o.rs, o.re)) o.flags |= OV_SIMZERO;
    } else walkOverlap(ix, wm, ws, o, i, false);
    o.chainLen = m0;                       // chain no longer needed: keep the pre-score matchCnt here
    if (!(o.flags & OV_ISREF)) ovKeepRunSize(o, runSize);
    wm.ov[wm.ord[i]] = o;
  }
  __syncthreads();
  return true;
}

// SeqSet::GetOverlapsFromRead (SeqSet.hpp:1508-2124, readType 0) for the segment in wm.seg / wm.rc.
// Scored and filtered overlaps are appended to wm.fin (coordinates shifted by `shift`).
// Returns the reference's return value (-1, 0 or the overlap count); -2 on capacity overflow.
// ROWS: the row-per-overlap finishing pass for short lists (the AddRead / AssignRead kernels; the rough-annotation kernels, whose
// overlaps are reference genes, keep their register budget).
template <bool ROWS>
__device__ int overlapsFromSegment(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int segLen, int strandArg, int barcode,
                                   bool skipRepeats, int shift, DPScratch sc, unsigned long long &hitTotal, int onlySeq = -1) {
  const int lane = tid(), NT = nthr();
  if (segLen < ix.k) return -1;
  int overlapCnt = 0;
  if (lane == 0) { ws->nAll = 0; ws->nOther = 0; ws->strand0 = 0; ws->vjRescue = 0; }
  if (ROWS && onlySeq >= 0) {
    // Restricted re-query (T4QueryArgs::onlySeq): every overlap of the read with ONE contig, scored; nothing that looks across
    // contigs is applied (strand of the best overlap, pre-filters, similarity cut: the caller merges with what it holds)
    const int H = seedChainPass<ROWS>(ix, wm, ws, segLen, strandArg, barcode, false, false, ix.hitLenRequired, 0, onlySeq);
    if (H < 0) return -2;
    hitTotal += (unsigned long long)H;
    __syncthreads();
    overlapCnt = ws->ovCount;
    if (ws->overflow || overlapCnt > wm.maxOv || overlapCnt > wm.maxFin) return -2;
    if (overlapCnt == 0) return 0;
    for (int i = lane; i < overlapCnt; i += NT) wm.ord[i] = (unsigned short)i;
    __syncthreads();
    if (!scoreOverlaps<ROWS>(ix, wm, ws, overlapCnt, sc)) return -2;
    for (int i = lane; i < overlapCnt; i += NT) { OvRec o = wm.ov[i]; o.rs += shift; o.re += shift; o.chainLen = 0; wm.fin[i] = o; }
    __syncthreads();
    if (lane == 0) ws->finCount = overlapCnt;
    __syncthreads();
    return overlapCnt;
  }
  if (skipRepeats) {
    int H = seedChainPass<ROWS>(ix, wm, ws, segLen, strandArg, barcode, true, false, ix.hitLenRequired, 0);
    if (H < 0) return -2;
    hitTotal += (unsigned long long)H;
    __syncthreads();
    overlapCnt = ws->ovCount;
  }
  if (overlapCnt == 0) {
    __syncthreads();
    int H = seedChainPass<ROWS>(ix, wm, ws, segLen, strandArg, barcode, false, false, ix.hitLenRequired, 1);
    if (H < 0) return H == -3 ? -3 : -2;
    hitTotal += (unsigned long long)H;
    __syncthreads();
    overlapCnt = ws->ovCount;
    if (overlapCnt == 0) {
      // VJ junction rescue on the hits of this pass (SeqSet.hpp:1570-1575)
      __syncthreads();
      int H2 = seedChainPass<ROWS>(ix, wm, ws, segLen, strandArg, barcode, false, true, 17, 0);
      if (H2 < 0) return -2;
      __syncthreads();
      int n = ws->ovCount;
      if (n > wm.maxOv) return -2;
      __syncthreads();
      if (lane == 0) ws->ovCount = selectVJPair(ix, wm, n);
      __syncthreads();
      overlapCnt = ws->ovCount;
      if (overlapCnt == 0) return 0;
      if (lane == 0) ws->vjRescue = 1;   // (only reference genes can pair up there: never in a set of contigs alone, where restricted re-queries live -- enforced, not assumed: ADVICE r4)
    }
  }
#ifdef T4_DEBUG
  if (lane == 0) printf("DBG seg len %d overlapCnt %d overflow %d\n", segLen, overlapCnt, ws->overflow);
#endif
  if (ws->overflow || overlapCnt > wm.maxOv) return -2;
  PHASE_MARK(ws, 8);
  // std::sort(overlaps) by operator< (the order is total on distinct overlaps)
  bool keySorted = false;
#if T4_OPT_OVKEYSORT
  // (threshold: 256 overlaps; lower when the testing aid T4Work::capLimit has shrunk the staging block, so that small cases come here)
  if ((ROWS || !T4_V0_NOKEYSORT) && !wm.ldsArrays && wm.ldsSort && overlapCnt > (wm.ldsSortCap >= 4096 ? 256 : wm.ldsSortCap / 16) && overlapCnt <= 16384)
    keySorted = sortOverlapsByKey(wm, ws, overlapCnt);
#endif
  if (!keySorted)
  for (int i = lane; i < overlapCnt; i += NT) {
    OvRec me = wm.ov[i];
    int rank = 0;
    for (int j = 0; j < overlapCnt; ++j) {
      if (j == i) continue;
      OvRec ot = wm.ov[j];
      const int cm = ovCmp(ot, me, false);
      if (cm < 0 || (cm == 0 && j < i)) ++rank;
    }
    wm.ord[rank] = (unsigned short)i;
  }
  __syncthreads();
  // keep the overlaps on the strand of the best one (SeqSet.hpp:1601-1616), order preserved
  int strand0 = wm.ov[wm.ord[0]].flags & OV_PLUS;
  int kept = 0;
  for (int i0 = 0; i0 < overlapCnt; i0 += NT) {
    int i = i0 + lane;
    int o = i < overlapCnt ? wm.ord[i] : 0;
    bool keep = i < overlapCnt && ((wm.ov[o].flags & OV_PLUS) == strand0);
    int tot;
    int inc = blockInclScan(keep ? 1 : 0, ws->red, tot);   // barriers inside: every ord[i] of the chunk is read
    if (keep) wm.ord[kept + inc - 1] = (unsigned short)o;   // target <= i: compaction in place, chunk by chunk
    kept += tot;
    __syncthreads();
  }
  if (lane == 0) { ws->nAll = kept; ws->nOther = overlapCnt - kept; ws->strand0 = strand0 ? 1 : 0; }
  overlapCnt = kept;
  PHASE_MARK(ws, 9);
  if (!scoreOverlaps<ROWS>(ix, wm, ws, overlapCnt, sc)) return -2;
  PHASE_MARK(ws, 10);
  if (ix.hasNovel && overlapCnt > 50) {
    if (ROWS || !T4_V0_SERIALPRE) prefilterNovel(ix, wm, ws, overlapCnt, segLen);
    else {
      // (rough-annotation kernels: the sequential form, inline -- the out-of-line call cost them a tenth of their speed)
      if (lane == 0) {
        int best = -1;
        const int len = segLen;
        for (int i = 0; i < overlapCnt; ++i) {
          OvRec &o = wm.ov[wm.ord[i]];
          bool isRef = (o.flags & OV_ISREF) != 0;
          if (!isRef && best != -1) {
            const OvRec &bn = wm.ov[wm.ord[best]];
            double bs = ovSim(bn);
            int m0 = o.chainLen;
            bool cut = false;
            if (bn.rs == 0 && bn.re == len - 1) {
              if (bs == 1) cut = true;
              else if (bs > ix.repeatSim && m0 < 0.9 * bn.matchCnt) cut = true;
            }
            if (!cut && bn.rs + len - 1 - bn.re < ix.radius) {
              if (bs == 1 && m0 < 0.9 * bn.matchCnt) cut = true;
              else if (bs > ix.repeatSim && m0 < 0.8 * bn.matchCnt) cut = true;
            }
            if (!cut && o.ss - o.rs >= ix.radius && o.se + (len - 1 - o.re) + ix.radius < ix.seqs[o.seqIdx].len &&
                bn.matchCnt > 0.97 * (2 * len) && bs > ix.repeatSim && m0 < 0.9 * bn.matchCnt) cut = true;
            if (!cut && m0 < 0.4 * bn.matchCnt) cut = true;
            if (!cut && overlapCnt > 1000 && m0 < 0.9 * bn.matchCnt) cut = true;
            if (cut) { o.matchCnt = m0; o.indelCnt = 0; o.flags |= OV_SIMZERO; continue; }
          }
          if (!isRef && !(o.flags & OV_SIMZERO) && ovSim(o) > 0) {
            if (best == -1 || ovLess(o, wm.ov[wm.ord[best]], true)) best = i;
          }
        }
      }
      __syncthreads();
    }
  }
#ifdef T4_DEBUG
  if (lane == 0) for (int i = 0; i < overlapCnt; ++i) { OvRec o = wm.ov[wm.ord[i]]; printf("DBG ov %d seq %d %d-%d %d-%d m %d ind %d fl %d sim %f\n", i, o.seqIdx, o.rs, o.re, o.ss, o.se, o.matchCnt, o.indelCnt, o.flags, ovSim(o)); }
#endif
  PHASE_MARK(ws, 11);
  // similarity thresholds (SeqSet.hpp:2105-2119), order preserved; append to fin
  int base = ws->finCount;
  int outCnt = 0;
  for (int i0 = 0; i0 < overlapCnt; i0 += NT) {
    int i = i0 + lane;
    bool keep = false;
    OvRec o;
    if (i < overlapCnt) {
      o = wm.ov[wm.ord[i]];
      double sim = ovSim(o);
      keep = (o.flags & OV_ISREF) ? !(sim < ix.refSim) : !(sim < ix.novelSim);
    }
    int tot;
    int inc = blockInclScan(keep ? 1 : 0, ws->red, tot);
    int pos = base + outCnt + inc - 1;
    if (keep) {
      if (pos < wm.maxFin) { o.rs += shift; o.re += shift; o.chainLen = 0; wm.fin[pos] = o; }
      else ws->overflow = 1;
    }
    outCnt += tot;
  }
  __syncthreads();
  if (lane == 0) ws->finCount = base + outCnt;
  __syncthreads();
  PHASE_MARK(ws, 0);
  if (ws->overflow) return -2;
  return outCnt;
}


__device__ __forceinline__ void storeOverlap(T4OverlapOut *dst, const OvRec &o) {
  T4OverlapOut t;
  t.seqIdx = o.seqIdx; t.readStart = o.rs; t.readEnd = o.re; t.seqStart = o.ss; t.seqEnd = o.se;
  t.strand = (o.flags & OV_PLUS) ? 1 : -1; t.matchCnt = o.matchCnt; t.indelCnt = o.indelCnt;
  t.similarity = ovSim(o);
  *dst = t;
}

// GetContigIntervals (SeqSet.hpp:5289-5321) on the whole read in wm.seg. One lane.
__device__ void contigIntervals(const char *read, int gapN, WaveState *ws) {
  int n = 0;
  for (int i = 0; read[i];) {
    int NCnt = 0, j;
    for (j = i + 1; read[j]; ++j) {
      if (j >= i + gapN && read[j - gapN] == 'N') --NCnt;
      if (read[j] == 'N') ++NCnt;
      if (NCnt >= gapN) break;
    }
    if (n < 64) { ws->contigA[n] = (short)i; ws->contigB[n] = (short)(read[j] ? j - gapN : j - 1); }
    ++n;
    if (!read[j]) break;
    i = j + 1;
  }
  ws->nContig = n;
}

// The level-0 part of SeqSet::AnnotateRead after the per-contig overlaps are known
// (SeqSet.hpp:6167-6321). fin[0..n) holds all contigs' overlaps; ord = their sorted order.
// `kept` is scratch for n ints. One lane. Writes the four gene overlaps.
// `first[i]` (precomputed by all lanes, see annotatePrepare): sorted position of the first overlap of the same sequence that
// qualifies for the list (V/J/C gene, similarity >= 0.8), or -1. The reference's linear searches over the kept list
// (SeqSet.hpp:6160-6215) then cost O(1) per overlap: the list slot of a sequence is slotOf[first], and the first kept C
// gene is remembered when it is appended.
__device__ void annotateSelect(const T4IndexView &ix, WaveMem &wm, int n, int readLen, int *kept, T4OverlapOut *out) {
  int g[4] = {-1, -1, -1, -1};
  int k = 0, cSlot = -1;
  const int *first = kept + n;
  int *slotOf = kept + 2 * n;
  for (int i = 0; i < n; ++i) {
    const int f = first[i];
    if (f < 0) continue;                       // no overlap of this sequence is ever kept
    const int oi = wm.ord[i];
    const OvRec &o = wm.fin[oi];
    const int gt = OV_GENETYPE(o.flags) == 255 ? -1 : OV_GENETYPE(o.flags);
    if (gt < 0 || gt == 1) continue;
    if (f == i) {                              // used == -1 && sim >= 0.8
      slotOf[i] = k;
      if (gt == 3 && cSlot < 0) cSlot = k;
      kept[k++] = oi;
    } else if (f < i && gt == 2) {             // used != -1: a J gene may replace the kept overlap of its sequence
      const int used = slotOf[f];
      const OvRec &base = wm.fin[kept[used]];
      if (o.matchCnt == base.matchCnt && ovSimDen(o) == ovSimDen(base) && cSlot >= 0) {
        const OvRec &c = wm.fin[kept[cSlot]];
        if (o.re <= c.rs + 3) {
          int d1 = o.re - c.rs, d2 = base.re - c.rs;
          if (d1 < 0) d1 = -d1;
          if (d2 < 0) d2 = -d2;
          if (base.re > c.rs + 3 || d1 < d2) kept[used] = oi;
        }
      }
    }
  }
  if (k > 0) {
    char BT = 0, chain = 0;
    for (int i = 0; i < k; ++i) {
      const OvRec &o = wm.fin[kept[i]];
      char n0 = OV_NAME0(o.flags), n2 = OV_NAME2(o.flags);
      if (BT && n0 != BT) continue;
      BT = n0;
      if (chain && !(n2 == chain || (n2 == 'D' && chain == 'A') || (n2 == 'A' && chain == 'D'))) continue;
      chain = n2;
      int gt = OV_GENETYPE(o.flags) == 255 ? -1 : OV_GENETYPE(o.flags);
      if (gt >= 0 && g[gt] == -1) g[gt] = kept[i];
    }
    if (g[3] != -1) {
      const OvRec &c = wm.fin[g[3]];
      if (c.re - c.rs + 1 <= readLen / 2 && c.re - c.rs + 1 <= 50) {
        for (int i = 0; i < 3; ++i) {
          if (g[i] < 0) continue;
          const OvRec &x = wm.fin[g[i]];
          if ((x.re - 17 > c.rs || c.re < x.re) && c.ss >= 100) { g[3] = -1; break; }
        }
      }
    }
  }
  for (int t = 0; t < 4; ++t) {
    if (g[t] >= 0) storeOverlap(out + t, wm.fin[g[t]]);
    else {
      T4OverlapOut z;
      z.seqIdx = -1; z.readStart = z.readEnd = z.seqStart = z.seqEnd = -1; z.strand = 1; z.matchCnt = 0; z.indelCnt = 0; z.similarity = 0;
      out[t] = z;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SeqSet::ExtendOverlap (SeqSet.hpp:1165-1277) for a list of overlaps of the current read.
// Both overhang alignments are GlobalAlignment_PosWeight calls with lent == lenp, so the reference's
// ungapped early return (<= 2 mismatches) is the common case and is evaluated one (overlap, side) per lane;
// the others run the wave-cooperative banded DP with one direction byte per cell in LDS and a lane-0
// traceback that reproduces the reference's edit string (needed for the "good overhang" scans).
// ------------------------------------------------------------------------------------------------
struct ExtSide { short size, good, match, mis, indel, pending; };
struct ExtOut { int ret, rs, re, ss, se, matchCnt, simFail, den; };   // similarity = matchCnt / den unless simFail

// Bytes of one overhang alignment's buffer: direction bytes (L + 1) * 11, the edit string 2 * L + 8, and the L predicate bytes
// of the target staged beside them -- every step of the anti-diagonal sweep reads one target byte per lane, and from global
// memory that load's latency (not its bandwidth) was the whole cost of a step.
#define T4_EXT_BYTES(L) (((L) + 1) * 11 + 3 * (L) + 8)
#define T4_EXT_WST(buf, L) ((buf) + ((L) + 1) * 11 + 2 * (L) + 8)
// What ExtendOverlap wants to know about the traceback path from the origin to a cell, carried along with the score so that no
// walk is needed for it: has the path met an indel yet, and if not, over its k steps so far, the number of matches `tmp` and the
// "good" length (the largest k' with more than 3/4 matches among the first k' steps, at a match; SeqSet.hpp:1224-1235). A cell
// inherits the state of the predecessor the traceback would take from it (diagonal before insert before delete,
// AlignAlgo.hpp:172-190). Packed: good | tmp << 9 | k << 18 | indel << 27.
#define PS_INDEL (1u << 27)
__device__ __forceinline__ unsigned pathStepDiag(unsigned s, bool eq) {
  if (s & PS_INDEL) return s;
  unsigned good = s & 511u, tmp = (s >> 9) & 511u, k = ((s >> 18) & 511u) + 1u;
  if (eq) { ++tmp; if (4u * tmp > 3u * k) good = k; }
  return good | tmp << 9 | k << 18;
}
// Path state of a border cell (i0, 0) or (0, j0), n = i0 + j0 >= 1. The reference's traceback walks a border by indels -- except
// for its last step: at (1, 0) and (0, 1) the border recurrence does not reproduce m[0][0] = 0, so the step is recorded as a MATCH
// that consumes both sequences (AlignAlgo.hpp:191-203; the walk then ends below zero). An alignment through a border cell thus
// starts with one match; further out on the border, indels follow it.
__device__ __forceinline__ unsigned pathBorder(int n) {
  const unsigned one = 1u | 1u << 9 | 1u << 18;   // k = 1, tmp = 1, good = 1
  return n == 1 ? one : (one | PS_INDEL);
}
// banded posWeight DP of an L x L problem (W = 11) by one wavefront; dir bytes -> dirbuf[i * 11 + d]; returns (to every lane) the
// path state of the end cell (L, L)
__device__ unsigned dpWaveTracePW(const T4PW *w, int L, const char *p, unsigned char *dirbuf) {
  const int d = laneId(), W = 11, leftBand = 5;
  unsigned char *wst = T4_EXT_WST(dirbuf, L);
  for (int t = d; t < L; t += 64) wst[t] = w[t];
  waveLdsSync();
  const int negInf = (L + 1) * (L + 1) * (-4);
  int M = negInf;
  unsigned S = PS_INDEL, fin = PS_INDEL;
  { int j0 = d - leftBand; if (d < W && j0 >= 0 && j0 <= L) M = j0 == 0 ? 0 : -4 - 4 * j0; }
  const int lastStep = 2 * L + W - 1;
  for (int s = 2; s <= lastStep; ++s) {
    int lM = waveUp1(M), uM = waveDown1(M);
    const unsigned lS = waveUp1(S), uS = waveDown1(S);
    const int i2 = s - d, i = i2 >> 1, j = i - leftBand + d;
    if (d < W && (i2 & 1) == 0 && i >= 1 && i <= L && j >= 1 && j <= L) {
      if (j == 1) lM = -4 - 4 * i; else if (d == 0) lM = negInf;
      if (i == 1) uM = -4 - 4 * j; else if (d + 1 >= W) uM = negInf;
      int dM;
      if (i == 1) dM = (j - 1 == 0) ? 0 : -4 - 4 * (j - 1);
      else if (j == 1) dM = -4 - 4 * (i - 1);
      else dM = M;
      const bool eq = baseEqualW(wst[j - 1], p[i - 1]);
      const int dsc = dM + (eq ? 2 : -2);
      int m = dsc;
      if (lM - 4 > m) m = lM - 4;
      if (uM - 4 > m) m = uM - 4;
      dirbuf[i * W + d] = (unsigned char)((lM - 4 == m ? 1 : 0) | (uM - 4 == m ? 2 : 0) | (dsc == m ? 4 : 0) | (eq ? 8 : 0));
      // the predecessor's path state
      if (dsc == m) S = pathStepDiag((i == 1 && j == 1) ? 0u : ((i == 1 || j == 1) ? pathBorder(i + j - 2) : S), eq);
      else if (uM - 4 == m) S = (i == 1 ? pathBorder(j) : uS) | PS_INDEL;
      else S = (j == 1 ? pathBorder(i) : lS) | PS_INDEL;
      if (i == L && j == L) fin = S;
      M = m;
    }
  }
  return (unsigned)__shfl((int)fin, leftBand);   // the end cell (L, L) is lane leftBand's
}
// The same DP for FOUR problems per wavefront, one per 16-lane DPP row (the band is 11 columns wide, so a whole wavefront per
// problem left 53 lanes idle): every row has its own (w, L, p, dirbuf); Lmax = the longest L of the wavefront's rows
// (wave-uniform trip count; a row with L == 0 computes nothing).
__device__ unsigned dpRowTracePW(const T4PW *w, int L, const char *p, unsigned char *dirbuf, int Lmax) {
  const int d = laneId() & 15, W = 11, leftBand = 5;
  unsigned char *wst = T4_EXT_WST(dirbuf, L);
  for (int t = d; t < L; t += 16) wst[t] = w[t];
  waveLdsSync();
  const int negInf = (L + 1) * (L + 1) * (-4);
  int M = negInf;
  unsigned S = PS_INDEL, fin = PS_INDEL;
  { int j0 = d - leftBand; if (d < W && j0 >= 0 && j0 <= L) M = j0 == 0 ? 0 : -4 - 4 * j0; }
  const int lastStep = 2 * Lmax + W - 1;
  for (int s = 2; s <= lastStep; ++s) {
    int lM = rowUp1(M), uM = rowDown1(M);
    const unsigned lS = rowUp1(S), uS = rowDown1(S);
    const int i2 = s - d, i = i2 >> 1, j = i - leftBand + d;
    if (d < W && (i2 & 1) == 0 && i >= 1 && i <= L && j >= 1 && j <= L) {
      if (j == 1) lM = -4 - 4 * i; else if (d == 0) lM = negInf;
      if (i == 1) uM = -4 - 4 * j; else if (d + 1 >= W) uM = negInf;
      int dM;
      if (i == 1) dM = (j - 1 == 0) ? 0 : -4 - 4 * (j - 1);
      else if (j == 1) dM = -4 - 4 * (i - 1);
      else dM = M;
      const bool eq = baseEqualW(wst[j - 1], p[i - 1]);
      const int dsc = dM + (eq ? 2 : -2);
      int m = dsc;
      if (lM - 4 > m) m = lM - 4;
      if (uM - 4 > m) m = uM - 4;
      dirbuf[i * W + d] = (unsigned char)((lM - 4 == m ? 1 : 0) | (uM - 4 == m ? 2 : 0) | (dsc == m ? 4 : 0) | (eq ? 8 : 0));
      if (dsc == m) S = pathStepDiag((i == 1 && j == 1) ? 0u : ((i == 1 || j == 1) ? pathBorder(i + j - 2) : S), eq);   // see dpWaveTracePW
      else if (uM - 4 == m) S = (i == 1 ? pathBorder(j) : uS) | PS_INDEL;
      else S = (j == 1 ? pathBorder(i) : lS) | PS_INDEL;
      if (i == L && j == L) fin = S;
      M = m;
    }
  }
  return (unsigned)__shfl((int)fin, (laneId() & ~15) + leftBand);   // the end cell (L, L) is the row's lane leftBand
}
// The lean form for the ordered contig builder (extendOverlaps, lean): EIGHT problems per wavefront and nothing written but scores.
// An anti-diagonal sweep keeps a lane busy every other step only (lane d owns cell i = (s - d) / 2 when s - d is even), so a
// second problem of the same 16-lane row takes the odd steps: lane d computes for problem par = (s - d) & 1, and its three
// neighbours' values of the step before (and its own of two steps before) belong to the same problem by parity. What ExtendOverlap
// needs comes out of the sweep itself: the path state of the end cell (pathStepDiag) and, for a left overhang -- anchored at the
// END of the alignment -- the number of diagonal steps the traceback takes from (L, L) before its first indel, i.e. the run of
// main-diagonal cells ending at (L, L) whose score comes from the diagonal. wstA / wstB: the target stretches staged in LDS.
// Returns, to the row's lanes, state | run << 32 of problem A in outA and of problem B in outB.
__device__ void dpRowPairLean(const unsigned char *wstA, int LA, const char *pA, const unsigned char *wstB, int LB, const char *pB,
                              int Lmax, unsigned long long &outA, unsigned long long &outB) {
  const int d = laneId() & 15, W = 11, leftBand = 5;
  // The sweep is instruction-bound (one VALU instruction of a wavefront is four cycles whatever the number of active lanes), so the
  // loop is written for few instructions per step: a lane's problem on EVEN steps is fixed (problem d & 1), on odd steps it is the
  // other one, which makes every per-problem quantity loop-invariant per lane; the row index advances by one per pair of steps;
  // a lane keeps its last value per problem (E / O registers), and its neighbours' values of the step before are their registers
  // of the other parity. The end cell (L, L) of either problem is the last cell lane leftBand computes for it.
  typedef const T4_LDS_AS unsigned char *LdsB;
  typedef const T4_LDS_AS char *LdsC;
  const bool odd = (d & 1) != 0;
  const int LE = odd ? LB : LA, LO = odd ? LA : LB;
  const LdsB wE = (LdsB)(odd ? wstB : wstA), wO = (LdsB)(odd ? wstA : wstB);
  const LdsC pE = (LdsC)(odd ? pB : pA), pO = (LdsC)(odd ? pA : pB);
  const int negE = (LE + 1) * (LE + 1) * (-4), negO = (LO + 1) * (LO + 1) * (-4);
  const bool inBand = d < W, first = d == 0, last = d + 1 >= W;
  int ME = 0, MO = 0, RE = 0, RO = 0;
  unsigned SE = PS_INDEL, SO = PS_INDEL;
  // step s = 2, 3, ...: even steps s = 2u serve cell i = u - (d + odd) / 2 ... in closed form i = (s - d - par) >> 1
  int iE = (2 - d - (odd ? 1 : 0)) >> 1;          // row of the even-step problem at s = 2
  int iO = (3 - d - (odd ? 0 : 1)) >> 1;          // row of the odd-step problem at s = 3
  const int nPair = (2 * Lmax + W) / 2 + 1;
#define T4_PAIR_STEP(Mself, Mnb, Sself, Snb, Rself, L_, w_, p_, neg_, i_)                                                               \
  {                                                                                                                                   \
    int lM = rowUp1(Mnb), uM = rowDown1(Mnb);                                                                                          \
    const unsigned lS = rowUp1(Snb), uS = rowDown1(Snb);                                                                               \
    const int i = (i_), j = i - leftBand + d;                                                                                          \
    if (inBand && (unsigned)(i - 1) < (unsigned)(L_) && (unsigned)(j - 1) < (unsigned)(L_)) {                                          \
      if (j == 1) lM = -4 - 4 * i; else if (first) lM = (neg_);                                                                        \
      if (i == 1) uM = -4 - 4 * j; else if (last) uM = (neg_);                                                                         \
      const int dM = i == 1 ? (j == 1 ? 0 : -4 * j) : (j == 1 ? -4 * i : Mself);                                                       \
      const bool eq = baseEqualW((w_)[j - 1], (p_)[i - 1]);                                                                            \
      const int dsc = dM + (eq ? 2 : -2);                                                                                              \
      int m = dsc;                                                                                                                     \
      if (lM - 4 > m) m = lM - 4;                                                                                                      \
      if (uM - 4 > m) m = uM - 4;                                                                                                      \
      if (dsc == m) { Sself = pathStepDiag((i == 1 && j == 1) ? 0u : ((i == 1 || j == 1) ? pathBorder(i + j - 2) : Sself), eq); Rself = (i == 1 ? 0 : Rself) + 1; } \
      else if (uM - 4 == m) { Sself = (i == 1 ? pathBorder(j) : uS) | PS_INDEL; Rself = 0; }                                           \
      else { Sself = (j == 1 ? pathBorder(i) : lS) | PS_INDEL; Rself = 0; }                                                            \
      Mself = m;                                                                                                                       \
    }                                                                                                                                  \
  }
  for (int u = 0; u < nPair; ++u) {
    T4_PAIR_STEP(ME, MO, SE, SO, RE, LE, wE, pE, negE, iE)   // even step: neighbours computed this problem on the odd step before
    T4_PAIR_STEP(MO, ME, SO, SE, RO, LO, wO, pO, negO, iO)   // odd step
    ++iE; ++iO;
  }
#undef T4_PAIR_STEP
  // lane leftBand = 5 is odd: its even-step problem is B, its odd-step problem A
  const int src = (laneId() & ~15) + leftBand;
  const unsigned sA = (unsigned)__shfl((int)SO, src), sB = (unsigned)__shfl((int)SE, src);
  const unsigned rA = (unsigned)__shfl(RO, src), rB = (unsigned)__shfl(RE, src);
  outA = (unsigned long long)sA | ((unsigned long long)rA << 32);
  outB = (unsigned long long)sB | ((unsigned long long)rB << 32);
}
// traceback of the above (AlignAlgo.hpp:160-205); align[] receives the edit string, returns its length. One lane.
__device__ int tracebackPW(const unsigned char *dirbuf, int L, signed char *align) {
  const int W = 11, leftBand = 5;
  int tagi = L, tagj = L, tag = 0;
  while (tagi > 0 || tagj > 0) {
    int a = 0;
    if (tagi > 0 && tagj > 0) {
      unsigned char bits = dirbuf[tagi * W + (tagj - tagi + leftBand)];
      if (bits & 1) a = 3;
      if (bits & 2) a = 2;
      if (bits & 4) a = (bits & 8) ? 0 : 1;
    } else if (tagi == 0) {
      int mL = (tagj - 1 == 0) ? 0 : (-4 - 4 * (tagj - 1));
      if (mL - 4 == -4 - 4 * tagj) a = 3;
    } else {
      int mU = (tagi - 1 == 0) ? 0 : (-4 - 4 * (tagi - 1));
      if (mU - 4 == -4 - 4 * tagi) a = 2;
    }
    align[tag++] = (signed char)a;
    if (a == 3) --tagj; else if (a == 2) --tagi; else { --tagi; --tagj; }
  }
  for (int i = 0, j = tag - 1; i < j; ++i, --j) { signed char x = align[i]; align[i] = align[j]; align[j] = x; }
  return tag;
}

// Extend the n overlaps wm.fin[ord[0..n)] of the read in wm.seg / wm.rc (length len). Results in `res[i]`.
// useFirstStrand: AssignRead aligns every overlap against the strand of overlaps[0] (SeqSet.hpp:4657-4659).
// Scratch: sides = 2 * n ExtSide records, dirbuf = dirBytes >= T4_EXT_BYTES(len) bytes (LDS whenever the caller has any).
// lean: the caller is the ordered contig builder (mode 4). SeqSet::AddRead reads nothing but readStart / readEnd of an overlap whose
// ExtendOverlap returned 0 (SeqSet.hpp:3597-3700: such records only enter its `failedExtendedOverlaps` containment test), and an
// alignment that leaves the diagonal has an indel, which alone makes ExtendOverlap return 0 (SeqSet.hpp:1203-1214). So for such a
// side only the "good" prefix from the anchor outward is derived -- for a left overhang the traceback starts at the anchor and stops
// at the first indel -- and the record of an overlap with a gapped side carries exact coordinates, return value 0, and the
// overlap's own matchCnt / similarity in place of the counts nobody reads.
__device__ void extendOverlaps(const T4IndexView &ix, WaveMem &wm, WaveState *ws, int n, int len, bool useFirstStrand,
                               double factor, ExtSide *sides, unsigned char *dirbuf, int dirBytes, ExtOut *res, bool lean = false) {
  const int lane = tid(), NT = nthr();
  const int plus0 = n > 0 ? (wm.fin[wm.ord[0]].flags & OV_PLUS) : 1;
  // E1: ungapped evaluation of every (overlap, side), one side per wavefront at a time: 64 overhang positions per step
  // (coalesced predicate bytes), mismatches by ballot, the "good" prefix (SeqSet.hpp:1187-1235: the longest stretch from the
  // anchor outward whose matches exceed 3/4 of its length) from prefix popcounts
  {
    const int wave = lane >> 6, nwv = NT >> 6, wl = lane & 63;
    for (int q = wave; q < 2 * n; q += nwv) {
      const OvRec &o = wm.fin[wm.ord[q >> 1]];
      const int side = q & 1;
      const T4SeqInfo si = ix.seqs[o.seqIdx];
      const char *r = (useFirstStrand ? plus0 : (o.flags & OV_PLUS)) ? wm.seg : wm.rc;
      int size, t0, p0;   // overhang length; first target / read position of the overhang
      if (side == 0) { size = o.rs < o.ss ? o.rs : o.ss; t0 = o.ss - size; p0 = o.rs - size; }
      else { int a = len - 1 - o.re, b = si.len - 1 - o.se; size = a < b ? a : b; t0 = o.se + 1; p0 = o.re + 1; }
      const T4PW *w = ix.pw + si.pwOff + t0;
      int mm = 0, good = 0, tmpBase = 0;
      for (int base = 0; base < size; base += 64) {
        const int k = base + wl;                                   // step k + 1 of the scan from the anchor outward
        const int pos = side == 0 ? size - 1 - k : k;              // the left overhang is scanned from its end
        const bool in = k < size;
        const bool eq = in && baseEqualW(w[pos], r[p0 + pos]);
        const unsigned long long bal = __ballot(eq);
        const int tmp = tmpBase + __popcll(bal & ((wl == 63) ? ~0ull : ((2ull << wl) - 1ull)));
        const unsigned long long gb = __ballot(eq && tmp > 0.75 * (k + 1));
        if (gb) good = base + 64 - __clzll((long long)gb);
        mm += __popcll(__ballot(in && !eq));
        tmpBase += __popcll(bal);
      }
      if (wl == 0) {
        ExtSide e;
        e.size = (short)size; e.match = (short)(size - mm); e.mis = (short)mm; e.indel = 0; e.pending = 0;
        if (size > 1 && !((size - mm) * 2 - mm * 2 >= size * 2 - 8)) e.pending = 1;   // needs the banded DP; `good` is what the ungapped
        e.good = (short)good;                                                           // alignment gives, final if the DP's traceback never leaves the diagonal
        sides[q] = e;
      }
    }
  }
  __syncthreads();
  PHASE_MARK(ws, 18);   // extend: list of the gapped sides
  if (lean) {
    // E2, lean form: every pending side goes through dpRowPairLean, eight per wavefront; the only scratch is the staged target bytes
    int nPend = 0, maxL = 0;
    for (int q0 = 0; q0 < 2 * n; q0 += NT) {
      const int q = q0 + lane;
      const bool pend = q < 2 * n && sides[q].pending;
      int tot;
      const int inc = blockInclScan(pend ? 1 : 0, ws->red, tot);
      if (pend) wm.cand[nPend + inc - 1] = (unsigned)q;
      nPend += tot;
      int mx;
      blockInclMaxScan(pend ? (int)sides[q].size : 0, ws->red, mx);
      if (mx > maxL) maxL = mx;
    }
    __syncthreads();
    PHASE_MARK(ws, 19);
    const int nw = NT >> 6, slice = (maxL + 15) & ~15;
    int perChunk = nw * 8;
    if (slice > 0 && perChunk * slice > dirBytes) perChunk = (dirBytes / slice) & ~7;
    if (nPend > 0 && perChunk < 8) { if (lane == 0) ws->unsupported = 1; nPend = 0; }
    const int wave = lane >> 6, wl = lane & 63, row = wl >> 4, l16 = wl & 15;
    for (int c0 = 0; c0 < nPend; c0 += perChunk) {
      // this row's two problems
      int Lq[2] = {0, 0}, qq[2] = {-1, -1};
      const unsigned char *wst[2] = {dirbuf, dirbuf};
      const char *pr[2] = {wm.seg, wm.seg};
      for (int z = 0; z < 2; ++z) {
        const int slot = wave * 8 + row * 2 + z, t = c0 + slot;
        if (slot < perChunk && t < nPend) {
          const int q = (int)wm.cand[t];
          const OvRec &o = wm.fin[wm.ord[q >> 1]];
          const int side = q & 1, L = sides[q].size;
          const T4SeqInfo si = ix.seqs[o.seqIdx];
          const char *r = (useFirstStrand ? plus0 : (o.flags & OV_PLUS)) ? wm.seg : wm.rc;
          const T4PW *w = ix.pw + si.pwOff + (side == 0 ? o.ss - L : o.se + 1);
          unsigned char *dst = dirbuf + (size_t)slot * slice;
          for (int x = l16; x < L; x += 16) dst[x] = w[x];
          Lq[z] = L; qq[z] = q; wst[z] = dst; pr[z] = r + (side == 0 ? o.rs - L : o.re + 1);
        }
      }
      waveLdsSync();
      int Lmax = Lq[0] > Lq[1] ? Lq[0] : Lq[1];
      { int x = __shfl_xor(Lmax, 16); if (x > Lmax) Lmax = x; x = __shfl_xor(Lmax, 32); if (x > Lmax) Lmax = x; }
      unsigned long long out[2];
      dpRowPairLean(wst[0], Lq[0], pr[0], wst[1], Lq[1], pr[1], Lmax, out[0], out[1]);
      for (int z = 0; z < 2; ++z) {   // (wave-uniform control flow: the ballots below are taken by whole wavefronts)
        const bool have = qq[z] >= 0;
        const unsigned pst = (unsigned)out[z];
        const int q = have ? qq[z] : 0, side = q & 1, L = Lq[z];
        const bool gapped = have && (pst & PS_INDEL) != 0;
        DBG_ADD(6, (have && l16 == 0) ? 1 : 0); DBG_ADD(7, (gapped && l16 == 0) ? 1 : 0);
        // left overhang: its last `run` steps are the main-diagonal cells next to the anchor; "good" is the ungapped scan of E1 cut
        // off there (steps k = 1 .. run from the anchor outward, position L - k of the overhang)
        const int run = (gapped && side == 0) ? (int)(out[z] >> 32) : 0;
        int runMax = run;
        { int x = __shfl_xor(runMax, 16); if (x > runMax) runMax = x; x = __shfl_xor(runMax, 32); if (x > runMax) runMax = x; }
        int good = 0, tmpBase = 0;
        for (int base = 0; base < runMax; base += 16) {
          const int k = base + l16;
          const bool eq = k < run && baseEqualW(wst[z][L - 1 - k], pr[z][L - 1 - k]);
          const unsigned bal = (unsigned)((__ballot(eq) >> (wl & 48)) & 0xFFFFull);
          const int tmp = tmpBase + __popc(bal & ((2u << l16) - 1u));
          const unsigned gb = (unsigned)((__ballot(eq && 4 * tmp > 3 * (k + 1)) >> (wl & 48)) & 0xFFFFull);
          if (gb) good = base + 32 - __clz(gb);
          tmpBase += __popc(bal);
        }
        if (gapped && side == 1) good = (int)(pst & 511u);
        if (have && l16 == 0) {
          ExtSide e = sides[q];
          if (gapped) { e.match = 0; e.mis = 0; e.indel = 1; e.good = (short)good; }   // else: the ungapped alignment, the counts of E1 stand
          e.pending = 0;
          sides[q] = e;
        }
      }
      __syncthreads();
    }
    PHASE_MARK(ws, 21);
  } else {
  // E2: gapped sides, compacted into a list (wm.cand is dead here): first those whose direction bytes fit a quarter of a
  // wavefront's share of the buffer -- four per wavefront (one per 16-lane row), their tracebacks one lane per side -- then the
  // longer ones, one wavefront per side with its own slice; a side that fits no slice waits for the serial pass that owns the
  // whole buffer
  const int nwE = NT >> 6, perChunk = nwE * 4;
  const int qslice = (dirBytes / perChunk) & ~15;
  int nFit = 0, nPendSides = 0;
  for (int part = 0; part < 2; ++part) {
    for (int q0 = 0; q0 < 2 * n; q0 += NT) {
      const int q = q0 + lane;
      bool pend = q < 2 * n && sides[q].pending;
      if (pend) { const int size = sides[q].size; const bool fitsQ = T4_EXT_BYTES(size) <= qslice; pend = (part == 0) == fitsQ; }
      int tot;
      const int inc = blockInclScan(pend ? 1 : 0, ws->red, tot);
      if (pend) wm.cand[nPendSides + inc - 1] = (unsigned)q;
      nPendSides += tot;
    }
    if (part == 0) nFit = nPendSides;
  }
  __syncthreads();
  // finish one side from its edit string (any lane)
  auto finishSide = [&](int q, const signed char *align, int alen) {
    const int side = q & 1;
    int m = 0, mm = 0, ind = 0, good = 0, tmp = 0;
    for (int k = 0; k < alen; ++k) { if (align[k] == 0) ++m; else if (align[k] == 1) ++mm; else ++ind; }
    if (side == 0) {
      for (int i = alen - 1, k = 1; i >= 0; --i, ++k) {
        if (align[i] == 0) { ++tmp; if (tmp > 0.75 * k) good = k; }
        else if (align[i] != 1) break;
      }
    } else {
      for (int i = 0; i < alen; ++i) {
        if (align[i] == 0) { ++tmp; if (tmp > 0.75 * (i + 1)) good = i + 1; }
        else if (align[i] != 1) break;
      }
    }
    ExtSide e = sides[q];
    e.match = (short)m; e.mis = (short)mm; e.indel = (short)ind; e.good = (short)good; e.pending = 0;
    sides[q] = e;
  };
  // lean form of traceback + finishSide for a side whose path leaves the diagonal: only `good` (and the fact that there is an indel)
  auto leanSide = [&](int q, const unsigned char *buf, int L, signed char *align) {
    const int side = q & 1;
    int good = 0, tmp = 0;
    if (side == 0) {   // the anchor is where the traceback starts: walk until the first indel
      const int W = 11, leftBand = 5;
      int tagi = L, tagj = L, k = 0;
      while (tagi > 0 && tagj > 0) {
        const unsigned char bits = buf[tagi * W + (tagj - tagi + leftBand)];
        if (!(bits & 4)) break;            // insert or delete (AlignAlgo.hpp:172-190: the diagonal wins whenever it gives the score)
        ++k;
        if (bits & 8) { ++tmp; if (tmp > 0.75 * k) good = k; }
        --tagi; --tagj;
      }
    } else {           // the anchor is where the traceback ends: the whole walk, then the prefix up to the first indel
      const int alen = tracebackPW(buf, L, align);
      for (int i = 0; i < alen; ++i) {
        if (align[i] == 0) { ++tmp; if (tmp > 0.75 * (i + 1)) good = i + 1; }
        else if (align[i] != 1) break;
      }
    }
    ExtSide e = sides[q];
    e.match = 0; e.mis = 0; e.indel = 1; e.good = (short)good; e.pending = 0;
    sides[q] = e;
  };
  PHASE_MARK(ws, 19);   // extend: four overhang DPs per wavefront + tracebacks
  {
    const int wave = lane >> 6, wl = lane & 63, row = wl >> 4;
    for (int c0 = 0; c0 < nFit; c0 += perChunk) {
      const int t = c0 + wave * 4 + row;
      int L = 0;
      const T4PW *w = ix.pw;
      const char *pr = wm.seg;
      if (t < nFit) {
        const int q = (int)wm.cand[t];
        const OvRec &o = wm.fin[wm.ord[q >> 1]];
        const int side = q & 1;
        L = sides[q].size;
        const T4SeqInfo si = ix.seqs[o.seqIdx];
        const char *r = (useFirstStrand ? plus0 : (o.flags & OV_PLUS)) ? wm.seg : wm.rc;
        w = ix.pw + si.pwOff + (side == 0 ? o.ss - L : o.se + 1);
        pr = r + (side == 0 ? o.rs - L : o.re + 1);
      }
      int Lmax = L;
      { int x = __shfl_xor(Lmax, 16); if (x > Lmax) Lmax = x; x = __shfl_xor(Lmax, 32); if (x > Lmax) Lmax = x; }
      const unsigned pst = dpRowTracePW(w, L, pr, dirbuf + (size_t)(wave * 4 + row) * qslice, Lmax);
      // The path state of the end cell says whether the traceback (AlignAlgo.hpp:160-205) meets an indel. If not, the alignment IS
      // the ungapped one, whose counts and "good" prefix the ungapped evaluation already left in sides[q]. If it does and the
      // caller is the ordered builder (lean), a right overhang -- anchored where the path starts -- has its "good" length in the
      // state as well; a left overhang is walked from the anchor to its first indel, any other case is walked whole, by one lane.
      if ((wl & 15) == 0 && t < nFit) {
        const int q = (int)wm.cand[t];
        ExtSide e = sides[q];
        if (!(pst & PS_INDEL)) e.pending = 0;
        else if (lean && (q & 1)) { e.match = 0; e.mis = 0; e.indel = 1; e.good = (short)(pst & 511u); e.pending = 0; }
        else e.pending = 2;
        sides[q] = e;
        DBG_ADD(6, 1); DBG_ADD(7, (pst & PS_INDEL) ? 1 : 0);
      }
      __syncthreads();
      if (lane < perChunk && c0 + lane < nFit) {
        const int q = (int)wm.cand[c0 + lane];
        if (sides[q].pending == 2) {
          const int size = sides[q].size;
          unsigned char *buf = dirbuf + (size_t)lane * qslice;
          signed char *align = (signed char *)(buf + (size + 1) * 11);
          if (lean) leanSide(q, buf, size, align);
          else {
            const int alen = tracebackPW(buf, size, align);
            finishSide(q, align, alen);
          }
        }
      }
      __syncthreads();
    }
  }
  PHASE_MARK(ws, 20);   // extend: the longer sides, one per wavefront
  {
    const int wave = lane >> 6, nw = NT >> 6, wl = lane & 63;
    const int slice = (dirBytes / nw) & ~15;
    const int nBig = nPendSides - nFit;
    for (int pass = 0; pass < 2 && nBig > 0; ++pass) {
      // pass 0: every wavefront takes sides that fit its slice; pass 1: wavefront 0 takes the rest with the whole buffer
      unsigned char *buf = pass == 0 ? dirbuf + wave * slice : dirbuf;
      const int room = pass == 0 ? slice : dirBytes;
      for (int t = pass == 0 ? wave : 0; t < nBig; t += pass == 0 ? nw : 1) {
        if (pass == 1 && wave != 0) break;
        const int q = (int)wm.cand[nFit + t];
        const int size = sides[q].size;
        const bool fits = T4_EXT_BYTES(size) <= slice;
        if ((pass == 0) != fits) continue;       // wave-uniform
        if (T4_EXT_BYTES(size) > room) { if (wl == 0) ws->unsupported = 1; continue; }
        const OvRec &o = wm.fin[wm.ord[q >> 1]];
        const int side = q & 1;
        const T4SeqInfo si = ix.seqs[o.seqIdx];
        const char *r = (useFirstStrand ? plus0 : (o.flags & OV_PLUS)) ? wm.seg : wm.rc;
        const int t0 = side == 0 ? o.ss - size : o.se + 1, p0 = side == 0 ? o.rs - size : o.re + 1;
        const unsigned pst = dpWaveTracePW(ix.pw + si.pwOff + t0, size, r + p0, buf);
        waveLdsSync();
        if (wl == 0) {
          if (!(pst & PS_INDEL)) { ExtSide e = sides[q]; e.pending = 0; sides[q] = e; }   // the ungapped alignment (see above)
          else if (lean && side == 1) { ExtSide e = sides[q]; e.match = 0; e.mis = 0; e.indel = 1; e.good = (short)(pst & 511u); e.pending = 0; sides[q] = e; }
          else {
            signed char *align = (signed char *)(buf + (size + 1) * 11);
            if (lean) leanSide(q, buf, size, align);
            else {
              const int alen = tracebackPW(buf, size, align);
              finishSide(q, align, alen);
            }
          }
        }
      }
      __syncthreads();
    }
  }
  }   // !lean
  __syncthreads();
  PHASE_MARK(ws, 21);   // extend: combine
  // E3: combine (SeqSet.hpp:1179-1266)
  for (int i = lane; i < n; i += NT) {
    const OvRec &o = wm.fin[wm.ord[i]];
    const ExtSide L = sides[2 * i], R = sides[2 * i + 1];
    int ret = 1;
    int left = L.size, right = R.size;
    int matchCnt = L.match + R.match, mismatchCnt = L.mis + R.mis;
    if (L.indel > 0) { left = 0; ret = 0; }
    if (R.indel > 0) { right = 0; ret = 0; }
    int thr = 2;
    if (left >= 2) ++thr;
    if (right >= 2) ++thr;
    const double density = 1.5 / ix.k;
    thr = (int)(thr * factor);
    if (mismatchCnt > thr && (double)mismatchCnt / (left + right) > density) ret = 0;
    ExtOut e;
    e.rs = o.rs - left; e.re = o.re + right; e.ss = o.ss - left; e.se = o.se + right;
    e.matchCnt = 2 * matchCnt + o.matchCnt; e.simFail = 0;
    e.den = e.re - e.rs + 1 + e.se - e.ss + 1;
    double sim = (double)e.matchCnt / (double)e.den;
    bool isRef = (o.flags & OV_ISREF) != 0;
    if ((isRef && sim < ix.refSim) || (!isRef && sim < ix.novelSim)) { e.simFail = 1; e.matchCnt = o.matchCnt; ret = 0; }
    if (lean && (L.indel > 0 || R.indel > 0)) { e.simFail = 1; e.matchCnt = o.matchCnt; ret = 0; }   // the counts of a gapped side were not derived
    if (ret == 0) { e.rs = o.rs - L.good; e.re = o.re + R.good; e.ss = o.ss - L.good; e.se = o.se + R.good; }
    e.ret = ret;
    res[i] = e;
  }
  __syncthreads();
}

// Process one read in one wavefront. Returns false when the read has to move to a larger tier.
// VARIANT 0: GetOverlapsFromRead / AnnotateRead level 0 (modes 0, 1); VARIANT 1: the modes that also run ExtendOverlap
// (2, 3, 4). Separate instantiations: the extension code must not cost the rough-annotation kernels registers.
#include "t4_wide.h"

// The candidate store (T4QueryArgs::candOut): every scored overlap of the pass, in scan order (wm.ov / wm.ord stand as
// overlapsFromSegment left them; a restricted re-query keeps them in wm.ov[0 .. ovCount) unordered). Out of line: its registers are its own.
__device__ T4_NI void emitCands(const T4CandArgs *cs, WaveMem &wm, WaveState *ws, long long r, int nc, bool identity) {
  const int lane = tid(), NT = nthr();
  if (nc <= 0) return;
  if (lane == 0) {
    const unsigned cb = atomicAdd(cs->candCursor, (unsigned)nc);
    ws->red[13] = (cb + (unsigned)nc > (unsigned)cs->candCap) ? -1 : (int)cb;
    if (ws->red[13] < 0) atomicOr(cs->candOverflow, 1);
  }
  __syncthreads();
  const int cb = ws->red[13];
  if (cb >= 0) {
    T4Cand *out = cs->candOut;
    for (int i = lane; i < nc; i += NT) out[cb + i] = ovToCand(wm.ov[identity ? i : (int)wm.ord[i]]);
    if (lane == 0) { cs->candBase[r] = cb; cs->candCnt[r] = nc; }
  }
  __syncthreads();
}

template <int VARIANT>
__device__ bool processRead(const T4IndexView &ix, const T4BatchView &bv, const T4Work &wk, const T4QueryArgs &qa,
                            WaveMem &wm, WaveState *ws, long long r, DPScratch sc) {
  const int lane = tid(), NT = nthr();
  const int len = bv.len[r];
  unsigned long long hitTotal = 0;
  if (lane == 0) { ws->overflow = 0; ws->unsupported = 0; ws->finCount = 0; ws->nContig = 0; ws->ovCount = 0; ws->statsStable = 1; ws->wideWant = 0; ws->nvN4[0] = ws->nvN4[1] = 0; ws->forceMin[0] = ws->forceMin[1] = 0; ws->useMarks = 0; }
#ifdef T4_PHASE_TIMING
  if (lane == 0) { ws->phaseT0 = clock64(); ws->phaseBase = wm.ldsArrays ? 0 : 32; ws->curPhase = ws->phaseBase; }
#endif
  __syncthreads();
  if (VARIANT == 1 && qa.mode == 4) {
    // the query half of SeqSet::AddRead (SeqSet.hpp:3437 + every ExtendOverlap of 3597 / 3746): one launch, results
    // consumed by the host-side ordered commit (t4_assembler)
    ExtSide *sides = (ExtSide *)wm.pairs;
    ExtOut *res = (ExtOut *)wm.ov;
    unsigned char *dirbuf = wm.dirBuf;
    int barcode = bv.barcode ? bv.barcode[r] : -1;
    // a pass that outgrows this workgroup's arrays (hits or overlaps) is spread over the chip: the wide query (t4_wide.h)
    const int onlySeq = qa.onlySeq ? qa.onlySeq[r] : -1;
    const bool wide = onlySeq < 0 && wk.wide != nullptr && !qa.skipRepeats && barcode == -1 && ix.hasNovel == 2 && !qa.views && qa.extendLater > 0;
    if (lane == 0) ws->wideWant = wide ? (wk.wide->minHits > 0 ? wk.wide->minHits : 1) : 0;
    if (lane == 0 && qa.cs) {
      const T4CandArgs *cs = qa.cs;
      if (onlySeq >= 0 && cs->forceMin) { const int f = cs->forceMin[r]; ws->forceMin[0] = f & 0xFFFF; ws->forceMin[1] = (f >> 16) & 0xFFFF; }
      if (onlySeq >= 0) ws->useMarks = cs->useMarks;
      if (cs->candCnt) cs->candCnt[r] = 0;
    }
    loadSegment(bv, r, 0, len, wm);
    int ret = overlapsFromSegment<VARIANT != 0>(ix, wm, ws, len, qa.strandPerRead[r], barcode, qa.skipRepeats != 0, 0, sc, hitTotal, onlySeq);
    if (onlySeq >= 0 && ret == -2) {   // (one contig's hits or overlaps beyond this workgroup's arrays: the caller asks for the whole query instead)
      if (lane == 0) { wk.status[r] = 5; qa.counts[r] = 0; atomicAdd(wk.hitCounter, hitTotal); }
      return true;
    }
    if (lane == 0 && onlySeq < 0) {
      if (qa.aux) qa.aux[r] = (ret == -2 || ret == -3) ? -1 : ((ws->nAll > 32767 ? 32767 : ws->nAll) | (((ws->nOther > 32767 || ws->vjRescue) ? 32767 : ws->nOther) << 15) | (ws->strand0 << 30));   // (a VJ-rescue result reads as "overlaps on the other strand": never eligible for a restricted re-query)
      if (qa.n4) qa.n4[r] = ws->nvN4[0] + ws->nvN4[1];
    }
    if (wide && (ret == -3 || (ret == -2 && !wm.ldsArrays))) {   // (overlaps beyond the LDS tier's arrays: the global-scratch pass of this workgroup first)
      hitTotal = 0;   // (the pass is counted by the wide query's seed stage)
      wideDeferRead(ix, wm, ws, *wk.wide, len, qa.strandPerRead[r], r, hitTotal);
      if (lane == 0) atomicAdd(wk.hitCounter, hitTotal);
      return true;
    }
    if (ret == -2) return false;
    int n = ret > 0 ? ret : 0;
    if (lane == 0 && qa.statsStable) qa.statsStable[r] = ws->statsStable;
    if (qa.cs && lane < 2) {
      int *s8 = qa.cs->stats8 + T4_QSTATS * r;
      s8[lane] = ws->nvN4[lane]; s8[2 + lane] = ws->nvN5[lane]; s8[4 + lane] = ws->nvSmax[lane]; s8[6 + lane] = ws->novelMin[lane];
      if (onlySeq >= 0) { s8[8 + lane] = ws->hullLo[lane]; s8[10 + lane] = ws->hullHi[lane]; }
    }
    if (qa.cs && qa.cs->candOut && ret >= 0) emitCands(qa.cs, wm, ws, r, onlySeq >= 0 ? ws->ovCount : ws->nAll, onlySeq >= 0);
    if (lane == 0) {   // room for this read's records in the result pool
      const int base = n > 0 ? (int)atomicAdd(qa.poolCursor, (unsigned)n) : 0;
      ws->red[15] = (base + n > qa.poolCap) ? -1 : base;
      ws->red[14] = base;
    }
    __syncthreads();
    const long long base = ws->red[15];
    const int reserved = ws->red[14];
    __syncthreads();
    if (base < 0) {
      // the pool is full: the host grows it and repeats the call. The part of the range this read reserved inside the pool must
      // not look like records to extendKernel (it runs over [0, min(cursor, poolCap)) before the host sees the status)
      if (qa.extendLater && qa.recRead) for (int i = reserved + lane; i < qa.poolCap && i < reserved + n; i += NT) if (i >= 0) qa.recRead[i] = -1;
      if (lane == 0) { wk.status[r] = 3; qa.counts[r] = 0; }
      return true;
    }
    if (lane == 0) qa.outBase[r] = (int)base;
    for (int i = lane; i < n; i += NT) { storeOverlap(qa.out + base + i, wm.fin[i]); wm.ord[i] = (unsigned short)i; }
    if (qa.extendLater) {
      // a read with many overlaps leaves their extensions to extendKernel (spread over the chip); the others extend here,
      // where a separate launch would only add its latency to the round
      const bool defer = n > qa.extendLater;
      for (int i = lane; i < n; i += NT) qa.recRead[base + i] = defer ? (int)r : -1;
      if (defer) {
        for (int i = lane; i < n; i += NT) storeOverlap(qa.outDev + base + i, wm.fin[i]);
        if (lane == 0) qa.counts[r] = ret;
        return true;
      }
    }
    __syncthreads();
    PHASE_MARK(ws, 16);
    extendOverlaps(ix, wm, ws, n, len, false, qa.factorPerRead[r], sides, dirbuf, wm.dirBytes, res, qa.leanExt != 0);
    PHASE_MARK(ws, 17);
    for (int i = lane; i < n; i += NT) {
      const OvRec &o = wm.fin[i];
      T4OverlapOut t;
      t.seqIdx = o.seqIdx; t.readStart = res[i].rs; t.readEnd = res[i].re; t.seqStart = res[i].ss; t.seqEnd = res[i].se;
      t.strand = (o.flags & OV_PLUS) ? 1 : -1; t.matchCnt = res[i].matchCnt;
      if (res[i].simFail) { t.indelCnt = o.indelCnt; t.similarity = ovSim(o); }
      else { t.indelCnt = 0; t.similarity = (double)t.matchCnt / (double)res[i].den; }
      qa.outExt[base + i] = t;
      qa.ret[base + i] = res[i].ret;
    }
    if (lane == 0) qa.counts[r] = ret;
  } else if (VARIANT == 1 && (qa.mode == 2 || qa.mode == 3)) {
    // scratch carved from the arrays that are dead after overlapsFromSegment: pairs (+cand) and the key area
    ExtSide *sides = (ExtSide *)wm.pairs;                 // 2 * maxFin * 12 B  <= cap * 4 B
    ExtOut *res = (ExtOut *)wm.ov;                         // maxFin * 32 B      <= maxOv * 40 B
    unsigned char *dirbuf = wm.dirBuf;                     // T4_EXT_BYTES(len) <= 8192 B in every tier
    int barcode = bv.barcode ? bv.barcode[r] : -1;
    loadSegment(bv, r, 0, len, wm);
    int n;
    if (qa.mode == 2) {
      // SeqSet::AssignRead (SeqSet.hpp:4632-4701)
      int ret = overlapsFromSegment<VARIANT != 0>(ix, wm, ws, len, qa.strandPerRead ? qa.strandPerRead[r] : qa.strand, barcode, false, 0, sc, hitTotal);
      if (ret == -2) return false;
      n = ret > 0 ? ret : 0;
      __syncthreads();
      for (int i = lane; i < n; i += NT) {   // std::sort(overlaps) with the scored similarity
        OvRec me = wm.fin[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
          if (j == i) continue;
          OvRec ot = wm.fin[j];
          const int cm = ovCmp(ot, me, true);
          if (cm < 0 || (cm == 0 && j < i)) ++rank;
        }
        wm.ord[rank] = (unsigned short)i;
      }
      __syncthreads();
      extendOverlaps(ix, wm, ws, n, len, true, barcode == -1 ? 1.0 : 2.0, sides, dirbuf, wm.dirBytes, res);
      if (lane == 0) {
        int hit = -1, staleIndel = 0;
        for (int i = 0; i < n; ++i) {
          if (res[i].ret == 1 && res[i].rs == 0 && res[i].re == len - 1) { hit = i; break; }
          if (res[i].simFail) staleIndel = wm.fin[wm.ord[i]].indelCnt;   // `extendedOverlap = overlap` leaves its fields behind
        }
        T4OverlapOut t;
        if (hit >= 0) {
          const OvRec &o = wm.fin[wm.ord[hit]];
          t.seqIdx = o.seqIdx; t.readStart = res[hit].rs; t.readEnd = res[hit].re; t.seqStart = res[hit].ss; t.seqEnd = res[hit].se;
          t.strand = (o.flags & OV_PLUS) ? 1 : -1; t.matchCnt = res[hit].matchCnt; t.indelCnt = staleIndel;
          t.similarity = (double)t.matchCnt / (double)res[hit].den;
          qa.ret[r] = o.seqIdx;
        } else {
          t.seqIdx = -1; t.readStart = t.readEnd = t.seqStart = t.seqEnd = -1; t.strand = 1; t.matchCnt = 0; t.indelCnt = 0; t.similarity = 0;
          qa.ret[r] = -1;
        }
        qa.out[r] = t;
      }
    } else {
      // ExtendOverlap of caller-supplied overlaps, each against the strand it names
      n = qa.inCounts[r];
      if (n < 0) n = 0;
      if (n > qa.maxPerRead) n = qa.maxPerRead;
      if (n > wm.maxFin) return false;
      for (int i = lane; i < n; i += NT) {
        T4OverlapOut in = qa.in[r * qa.maxPerRead + i];
        OvRec o;
        o.seqIdx = in.seqIdx; o.rs = in.readStart; o.re = in.readEnd; o.ss = in.seqStart; o.se = in.seqEnd;
        o.matchCnt = in.matchCnt; o.indelCnt = in.indelCnt; o.chainPos = 0; o.chainLen = 0;
        o.flags = (in.strand == 1 ? OV_PLUS : 0) | (ix.seqs[in.seqIdx].isRef ? OV_ISREF : 0) | (in.similarity == 0 ? OV_SIMZERO : 0);
        wm.fin[i] = o; wm.ord[i] = (unsigned short)i;
      }
      __syncthreads();
      extendOverlaps(ix, wm, ws, n, len, false, qa.mismatchFactor, sides, dirbuf, wm.dirBytes, res);
      for (int i = lane; i < n; i += NT) {
        const OvRec &o = wm.fin[i];
        T4OverlapOut t;
        t.seqIdx = o.seqIdx; t.readStart = res[i].rs; t.readEnd = res[i].re; t.seqStart = res[i].ss; t.seqEnd = res[i].se;
        t.strand = (o.flags & OV_PLUS) ? 1 : -1; t.matchCnt = res[i].matchCnt;
        if (res[i].simFail) { t.indelCnt = o.indelCnt; t.similarity = qa.in[r * qa.maxPerRead + i].similarity; }
        else { t.indelCnt = 0; t.similarity = (double)t.matchCnt / (double)res[i].den; }
        qa.out[r * qa.maxPerRead + i] = t;
        qa.ret[r * qa.maxPerRead + i] = res[i].ret;
      }
    }
  } else if (VARIANT == 3 && qa.mode == 5) {
    // SeqSet::HasHitInSet(read, 0) (SeqSet.hpp:3144-3327), the stage-0 candidate test (FastqExtractor.cpp:129-134):
    // hits -> buckets per (strand, sequence) -> the bucket with the most distinct read offsets per strand ->
    // GetOverlapsFromHits (filter 1) on the chosen bucket(s). The sorted keys hold every bucket as one contiguous range.
    int result = 0;
    if (len >= ix.k) {
      loadSegment(bv, r, 0, len, wm);
      unsigned *posStart = (unsigned *)wm.ov, *posPref = wm.pairs;
      const int nk = len - ix.k + 1;
      int H = seedPositions(ix, wm, len, 0, -1, false, posStart, posPref, ws->red);
      if (H > wm.hitLimit) return false;
      hitTotal += (unsigned long long)H;
      if (H > 0) {
        expandHits(ix, wm, nk, H, -1, false, posStart, posPref, ws->red);
        __syncthreads();
        if (H > 1) bitonicSort(wm.keys, H);
        __syncthreads();
        // bucket starts (compacted into wm.pairs) and, one lane per bucket, the number of distinct read offsets
        unsigned *starts = wm.pairs;
        int nB = 0;
        if (lane == 0) { ws->hhBest[0] = 0; ws->hhBest[1] = 0; }
        for (int i0 = 0; i0 < H; i0 += NT) {
          const int i = i0 + lane;
          const bool st = i < H && (i == 0 || KEY_G(wm.keys[i]) != KEY_G(wm.keys[i - 1]));
          int tot;
          const int inc = blockInclScan(st ? 1 : 0, ws->red, tot);
          if (st) starts[nB + inc - 1] = (unsigned)i;
          nB += tot;
        }
        __syncthreads();
        for (int b = lane; b < nB; b += NT) {
          const int s0 = (int)starts[b], e0 = b + 1 < nB ? (int)starts[b + 1] : H;
          unsigned m[(T4_MAXL + 31) / 32];
          for (int w = 0; w < (T4_MAXL + 31) / 32; ++w) m[w] = 0;
          for (int t = s0; t < e0; ++t) {
            const unsigned long long kt = wm.keys[t];
            const int a = KEY_C(kt) - T4_C_BIAS + KEY_B(kt);
            for (int w = 0; w < (T4_MAXL + 31) / 32; ++w) if (w == (a >> 5)) m[w] |= 1u << (a & 31);
          }
          int distinct = 0;
          for (int w = 0; w < (T4_MAXL + 31) / 32; ++w) distinct += __popc(m[w]);
          atomicMax(&ws->hhBest[KEY_PLUS(wm.keys[s0])], ((unsigned)distinct << 16) | (unsigned)(0xFFFF - b));
        }
        __syncthreads();
        const unsigned best0 = ws->hhBest[0], best1 = ws->hhBest[1];
        const int max0 = best0 ? (int)(best0 >> 16) : -1, max1 = best1 ? (int)(best1 >> 16) : -1;
        const bool both = max0 + ix.k - 1 >= ix.hitLenRequired && max1 + ix.k - 1 >= ix.hitLenRequired;
        int nOv[2] = {0, 0}, firstMatch[2] = {0, 0};
        const int single = max1 >= max0 ? 1 : 0;
        // bounds of the chosen buckets, read before the bucket list (wm.pairs) is reused by overlapsFromKeys
        int bs[2] = {0, 0}, be[2] = {0, 0};
        for (int tag = 0; tag <= 1; ++tag) {
          const unsigned best = tag ? best1 : best0;
          if (!best) continue;
          const int b = 0xFFFF - (int)(best & 0xFFFF);
          bs[tag] = (int)starts[b]; be[tag] = b + 1 < nB ? (int)starts[b + 1] : H;
        }
        for (int tag = 0; tag <= 1; ++tag) {
          if (!both && tag != single) continue;
          if (be[tag] == bs[tag]) continue;
          __syncthreads();
          if (lane == 0) ws->ovCount = 0;
          __syncthreads();
          WaveMem wb = wm;
          wb.keys = wm.keys + bs[tag];
          overlapsFromKeys(ix, wb, ws, be[tag] - bs[tag], ix.hitLenRequired, 1);
          __syncthreads();
          if (ws->overflow) return false;
          const int n = ws->ovCount < wm.maxOv ? ws->ovCount : wm.maxOv;
          int firstPos = 0x7FFFFFFF, fm = 0;       // the reference's tmpOverlaps[0]: the run lowest in diagonal order
          for (int i = 0; i < n; ++i) if (wm.ov[i].chainPos < firstPos) { firstPos = wm.ov[i].chainPos; fm = wm.ov[i].matchCnt; }
          nOv[tag] = n; firstMatch[tag] = fm;
        }
        int maxTag;
        if (both) {
          if (nOv[0] > 0 && nOv[1] > 0) maxTag = firstMatch[0] >= firstMatch[1] ? 0 : 1;
          else if (nOv[0] > 0) maxTag = 0;
          else maxTag = 1;
        } else maxTag = single;
        result = nOv[maxTag] == 0 ? 0 : (maxTag == 0 ? -1 : 1);
      }
    }
    if (lane == 0) qa.ret[r] = result;
  } else if (VARIANT == 0 && qa.mode == 0) {
    int barcode = bv.barcode ? bv.barcode[r] : -1;
    loadSegment(bv, r, 0, len, wm);
    int ret = overlapsFromSegment<VARIANT != 0>(ix, wm, ws, len, qa.strand, barcode, qa.skipRepeats != 0, 0, sc, hitTotal);
    if (ret == -2) return false;
    int n = ret > 0 ? ret : 0;
    if (lane == 0) qa.counts[r] = ret;
    for (int i = lane; i < n && i < qa.maxPerRead; i += NT) storeOverlap(qa.out + r * qa.maxPerRead + i, wm.fin[i]);
  } else if (VARIANT == 0) {
    loadSegment(bv, r, 0, len, wm);
    if (lane == 0) {   // fewer than 7 N in the whole read: no window of 7 N can exist, the read is one contig
      int nN = 0;
      for (int w = 0; w < bv.wnm; ++w) nN += __popc(wm.nmRow[w]);
      if (nN < 7 && len > 0) { ws->contigA[0] = 0; ws->contigB[0] = (short)(len - 1); ws->nContig = 1; }
      else contigIntervals(wm.seg, 7, ws);
    }
    __syncthreads();
    int nContig = ws->nContig;
    if (nContig > 64) { if (lane == 0) wk.status[r] = 1; return true; }
    for (int c = 0; c < nContig; ++c) {
      int a = ws->contigA[c], b = ws->contigB[c];
      if (nContig > 1 || a != 0 || b != len - 1) { __syncthreads(); loadSegment(bv, r, a, b - a + 1, wm); }
      int ret = overlapsFromSegment<VARIANT != 0>(ix, wm, ws, b - a + 1, 0, -1, false, a, sc, hitTotal);
      if (ret == -2) return false;
      __syncthreads();
    }
    int n = ws->finCount;
    PHASE_MARK(ws, 12);
    for (int i = lane; i < n; i += NT) {
      OvRec me = wm.fin[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        OvRec ot = wm.fin[j];
        const int cm = ovCmp(ot, me, true);
        if (cm < 0 || (cm == 0 && j < i)) ++rank;
      }
      wm.ord[rank] = (unsigned short)i;
    }
    __syncthreads();
    {   // first[i] for annotateSelect: all lanes, one sorted position each
      int *first = (int *)wm.cand + n;
      for (int i = lane; i < n; i += NT) {
        const int seq = wm.fin[wm.ord[i]].seqIdx;
        int f = -1;
        for (int j = 0; j <= i; ++j) {
          const OvRec &x = wm.fin[wm.ord[j]];
          if (x.seqIdx != seq) continue;
          const int gt = OV_GENETYPE(x.flags);
          if (gt == 255 || gt == 1) continue;
          if (ovSim(x) >= 0.8) { f = j; break; }
        }
        first[i] = f;
      }
    }
    __syncthreads();
    if (lane == 0) annotateSelect(ix, wm, n, len, (int *)wm.cand, qa.out + r * 4);
  }
  __syncthreads();
  PHASE_MARK(ws, 0);
  if (lane == 0) {
    if (ws->unsupported) wk.status[r] = 1;
    atomicAdd(wk.hitCounter, hitTotal);
  }
  return true;
}

// The query kernel. CAP > 0: LDS tier; CAP == 0: global-scratch tier. Persistent grid, one wave/block.
// VARIANT 0 / 1: see processRead; VARIANT 3: HasHitInSet (mode 5) alone, so that its bucket bit masks cost the annotation
// kernels no registers; VARIANT 2: mode 4 with every read matched against its own per-barcode image
// (qa.views[qa.viewOf[read]]).
#ifndef T4_WPE_SMALL
#define T4_WPE_SMALL 4
#endif
#ifndef T4_WPE_CELLS
#define T4_WPE_CELLS 1   // first launch of the per-barcode queries (1024-hit tier, VARIANT 2): wavefronts per SIMD the register allocation aims at
#endif
template <int CAP, int MAXOV, int NTHREADS, int VARIANT>
__global__ __launch_bounds__(NTHREADS)
// rough-annotation kernels of the two small tiers: register budget for 4 waves / SIMD (LDS lets that many groups in)
// (and of the 3072-hit tier for 3: its LDS lets three groups of four wavefronts onto a CU; left to itself the allocator took 176
// VGPRs in round 3 -- two groups -- and the tier ran 45 % longer, profiles/r03n_annotate_kernel_ab.txt)
__attribute__((amdgpu_waves_per_eu((VARIANT == 0 && CAP > 0 && CAP <= 2048) ? T4_WPE_SMALL : (VARIANT == 0 && CAP == 3072) ? 3 : (VARIANT == 2 && CAP == 1024) ? T4_WPE_CELLS : 1)))
void queryKernel(T4IndexView ixArg, T4BatchView bvArg, T4Work wkArg, T4QueryArgs qaArg) {
  constexpr int C = CAP > 0 ? CAP : 1;
  constexpr int M = CAP > 0 ? MAXOV : 1;
  __shared__ unsigned long long s_keys[C];
  __shared__ unsigned s_pairs[C + C / 3 + 2];   // pairs[C] followed by cand[C / 3 + 2]
  __shared__ OvRec s_ov[M];
  __shared__ OvRec s_fin[M];
  __shared__ unsigned short s_ord[M];
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ WaveState s_ws;
  __shared__ unsigned long long s_gsort[CAP > 0 ? 1 : 8192];   // global-scratch tier: staging buffer of the hit sort
  // The argument structs and the working-memory description are handed to out-of-line device functions by reference. As by-value
  // kernel parameters / locals that forces a copy of each into every lane's private stack (736 bytes of scratch per lane in the
  // AddRead kernel: ~370 KB written per workgroup, the bulk of the 68 GB of WRITE_SIZE per step that profiles/r03k_bench.json
  // shows). The variants that meet contig sets keep ONE copy per workgroup in LDS instead; the reference-set variants (VARIANT 0),
  // whose register budget is tuned, stay as they were.
  constexpr bool SHARED_ARGS = VARIANT != 0;
#ifdef T4_PHASE_TIMING
  if (threadIdx.x < T4_NPHASE) s_ws.phaseLocal[threadIdx.x] = 0;
#endif
  __shared__ T4IndexView s_ix;
  __shared__ T4BatchView s_bv;
  __shared__ T4Work s_wk;
  __shared__ T4QueryArgs s_qa;
  __shared__ WaveMem s_wm, s_wg;
  if (SHARED_ARGS) {
    if (threadIdx.x == 0) { s_ix = ixArg; s_bv = bvArg; s_wk = wkArg; s_qa = qaArg; }
    __syncthreads();
  }
  T4IndexView &ix = SHARED_ARGS ? s_ix : ixArg;
  const T4BatchView &bv = SHARED_ARGS ? s_bv : bvArg;
  const T4Work &wk = SHARED_ARGS ? s_wk : wkArg;
  const T4QueryArgs &qa = SHARED_ARGS ? s_qa : qaArg;
  WaveMem wmLocal;
  WaveMem &wm = SHARED_ARGS ? s_wm : wmLocal;
  if (!SHARED_ARGS || threadIdx.x == 0) {
    if (CAP > 0) {
      wm.keys = s_keys; wm.pairs = s_pairs; wm.cand = s_pairs + C; wm.ov = s_ov; wm.fin = s_fin; wm.ord = s_ord;
      wm.cap = CAP; wm.maxOv = MAXOV; wm.maxFin = MAXOV; wm.candCap = C / 3 + 2; wm.ldsArrays = 1;
      wm.hitLimit = (wk.capLimit > 0 && wk.capLimit < CAP) ? wk.capLimit : CAP;
      wm.ldsSort = nullptr; wm.ldsSortCap = 0;
      wm.dirBuf = (unsigned char *)s_keys; wm.dirBytes = C * 8;
    } else {
      size_t b = blockIdx.x;
      wm.keys = wk.gKeys + b * (size_t)wk.gCap;
      wm.pairs = wk.gPairs + b * (size_t)wk.gCap * 2;
      wm.cand = wm.pairs + wk.gCap;
      wm.ov = (OvRec *)(wk.gOv + b * (size_t)wk.gMaxOv * 10);
      wm.fin = (OvRec *)(wk.gFin + b * (size_t)wk.gMaxOv * 10);
      wm.ord = wk.gOrd + b * (size_t)wk.gMaxOv;
      wm.cap = wk.gCap; wm.maxOv = wk.gMaxOv; wm.maxFin = wk.gMaxOv; wm.candCap = wk.gCap; wm.ldsArrays = 0;
      wm.hitLimit = wk.gCap;
      wm.ldsSort = s_gsort; wm.ldsSortCap = CAP > 0 ? 0 : 8192;
      wm.dirBuf = (unsigned char *)s_gsort; wm.dirBytes = (int)sizeof s_gsort;
    }
    wm.seg = s_seg; wm.rc = s_rc;
  }
  if (SHARED_ARGS) __syncthreads();
  DPScratch sc;
  sc.rows = wk.dpRows + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (6 * T4_ROWW * 64);
  sc.dir = wk.dpDir + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * T4_DIR_BYTES;
  // Persistent grid; the cost of a read varies by orders of magnitude, so a block fetches its next list position from a counter
  // when it is done with one (a static stride leaves most of the grid waiting for the unluckiest block at the end of a launch).
  __shared__ int s_nextW;
  int w = blockIdx.x;
  while (w < wk.nList) {
    long long r = wk.list[w];
    if (VARIANT == 2) {   // the read's own set image
      if (SHARED_ARGS) { __syncthreads(); if (threadIdx.x == 0) s_ix = qa.views[qa.viewOf[r]]; __syncthreads(); }
      else ix = qa.views[qa.viewOf[r]];
    }
#ifdef __HIPCC__
    const unsigned long long tick0 = (VARIANT == 1 && qa.readTicks) ? wall_clock64() : 0ull;
#endif
    bool done = processRead<(VARIANT == 0 ? 0 : VARIANT == 3 ? 3 : 1)>(ix, bv, wk, qa, wm, &s_ws, r, sc);
    if (CAP == 8192 && VARIANT == 1 && !done && wk.gKeys) {
      // The read outgrew the LDS arrays (known right after its seed stage): the same workgroup goes on in its block's slice of
      // the global-scratch arrays instead of leaving the read to another launch -- an AddRead query round is one launch.
      WaveMem wgLocal;
      WaveMem &wg = SHARED_ARGS ? s_wg : wgLocal;
      __syncthreads();
      if (!SHARED_ARGS || threadIdx.x == 0) {
        wg = wm;
        const size_t b = blockIdx.x;
        wg.keys = wk.gKeys + b * (size_t)wk.gCap;
        wg.pairs = wk.gPairs + b * (size_t)wk.gCap * 2;
        wg.cand = wg.pairs + wk.gCap;
        wg.ov = (OvRec *)(wk.gOv + b * (size_t)wk.gMaxOv * 10);
        wg.fin = (OvRec *)(wk.gFin + b * (size_t)wk.gMaxOv * 10);
        wg.ord = wk.gOrd + b * (size_t)wk.gMaxOv;
        wg.cap = wk.gCap; wg.maxOv = wk.gMaxOv; wg.maxFin = wk.gMaxOv; wg.candCap = wk.gCap; wg.ldsArrays = 0; wg.hitLimit = wk.gCap;
        wg.ldsSort = s_keys; wg.ldsSortCap = C;   // the LDS arrays are free now: the hit sort is staged through the key array
        wg.dirBuf = (unsigned char *)s_keys; wg.dirBytes = C * 8;
        if (wk.capLimit > 0 && wk.capLimit < C) { int bsz = 64; while (bsz * 2 <= wk.capLimit) bsz *= 2; wg.ldsSortCap = bsz; }   // testing aid: small blocks
      }
      __syncthreads();
      done = processRead<1>(ix, bv, wk, qa, wg, &s_ws, r, sc);
      if (tid() == 0 && done && wk.nextCount) atomicAdd(wk.nextCount + 2, 1);   // statistics: reads served this way
    }
    if (tid() == 0) {
#ifdef __HIPCC__
      if (VARIANT == 1 && qa.readTicks) qa.readTicks[r] = (int)(wall_clock64() - tick0);
#endif
#ifdef T4_PHASE_TIMING
      if (VARIANT == 1 && qa.mode == 4) {   // this read's phases into the class of its overlap count
        const int nOv = qa.counts[r], cls = nOv < 5 ? 0 : nOv < 20 ? 1 : nOv <= 64 ? 2 : 3;
        for (int p_ = 0; p_ < T4_NPHASE; ++p_) if (s_ws.phaseLocal[p_]) { atomicAdd(&g_phaseByOverlaps[cls * T4_NPHASE + p_], (unsigned long long)s_ws.phaseLocal[p_]); s_ws.phaseLocal[p_] = 0; }
      }
#endif
      if (!done) {
        if (wk.nextList) { int slot = atomicAdd(wk.nextCount, 1); wk.nextList[slot] = (int)r; }
        else wk.status[r] = 2;
      }
      s_nextW = wk.workNext ? (int)gridDim.x + atomicAdd(wk.workNext, 1) : w + (int)gridDim.x;
    }
    __syncthreads();
    w = s_nextW;
    __syncthreads();
  }
}

// SeqSet::ExtendOverlap (SeqSet.hpp:1165-1277) of every record an AddRead query batch returned (T4QueryArgs::extendLater):
// one wavefront per block takes T4_EXT_NREC consecutive records at a time (of one read: a block's range is cut at read
// boundaries), rebuilds the read's characters from its packed words and runs the same extendOverlaps() the query kernel runs
// inside a workgroup. The records of one read no longer wait for each other in one workgroup's eight wavefronts: a read that
// overlaps two thousand contigs becomes a few hundred independent blocks.
#ifndef T4_EXT_NREC
#define T4_EXT_NREC 4   // (8 until round 5: a wavefront steps eight overhang alignments at a time, so eight records were two passes per block; the chip is empty behind a whole-query round and four records -- one pass -- per block shorten its tail: C2 kernels 26.5 -> 25.5 s, profiles/r05h_c2_ext4_*)
#endif
__global__ __launch_bounds__(64) void extendKernel(T4IndexView ix, T4BatchView bv, T4QueryArgs qa, int recBegin) {
  __shared__ OvRec s_fin[T4_EXT_NREC];
  __shared__ unsigned short s_ord[T4_EXT_NREC];
  __shared__ ExtSide s_sides[2 * T4_EXT_NREC];
  __shared__ ExtOut s_res[T4_EXT_NREC];
  __shared__ unsigned s_cand[2 * T4_EXT_NREC + 2];
  __shared__ unsigned long long s_dir[1024];   // 8 KiB of direction bytes: four sides of up to 170 bases at a time, longer ones one at a time
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ WaveState s_ws;
  WaveMem wm;
  wm.keys = s_dir; wm.pairs = (unsigned *)s_sides; wm.ov = (OvRec *)s_res; wm.fin = s_fin; wm.ord = s_ord; wm.cand = s_cand;
  wm.seg = s_seg; wm.rc = s_rc;
  wm.cap = 1024; wm.maxOv = T4_EXT_NREC; wm.maxFin = T4_EXT_NREC; wm.candCap = 2 * T4_EXT_NREC + 2; wm.ldsArrays = 1;
  wm.ldsSort = nullptr; wm.ldsSortCap = 0; wm.hitLimit = 0; wm.dirBuf = (unsigned char *)s_dir; wm.dirBytes = (int)sizeof s_dir;
  const int lane = tid();
  const int nRec = (int)*qa.poolCursor < qa.poolCap ? (int)*qa.poolCursor : qa.poolCap;
  for (int g = recBegin + (int)blockIdx.x * T4_EXT_NREC; g < nRec; g += (int)gridDim.x * T4_EXT_NREC) {
    int i0 = g;
    const int gEnd = g + T4_EXT_NREC < nRec ? g + T4_EXT_NREC : nRec;
    while (i0 < gEnd) {
      const int r = qa.recRead[i0];
      int i1 = i0 + 1;
      while (i1 < gEnd && qa.recRead[i1] == r) ++i1;
      if (r < 0) { i0 = i1; continue; }   // extended by the query kernel itself
      const int n = i1 - i0, len = bv.len[r];
      if (lane == 0) { s_ws.overflow = 0; s_ws.unsupported = 0; }
#ifdef T4_PHASE_TIMING
      if (lane == 0) { s_ws.phaseT0 = clock64(); s_ws.phaseBase = 0; s_ws.curPhase = 24; }   // phase 24: extension launch (the marks inside extendOverlaps take over)
#endif
      loadSegment(bv, r, 0, len, wm);
      if (lane < n) {
        const T4OverlapOut t = qa.outDev[i0 + lane];
        OvRec o;
        o.seqIdx = t.seqIdx; o.rs = t.readStart; o.re = t.readEnd; o.ss = t.seqStart; o.se = t.seqEnd;
        o.matchCnt = t.matchCnt; o.indelCnt = t.indelCnt; o.chainPos = 0; o.chainLen = 0;
        o.flags = t.strand == 1 ? OV_PLUS : 0;   // contig sets only (t4_add_query refuses reference sets)
        s_fin[lane] = o; s_ord[lane] = (unsigned short)lane;
      }
      __syncthreads();
      extendOverlaps(ix, wm, &s_ws, n, len, false, qa.factorPerRead[r], s_sides, (unsigned char *)s_dir, (int)sizeof s_dir, s_res, qa.leanExt != 0);
      if (lane < n) {
        const T4OverlapOut in = qa.outDev[i0 + lane];
        const ExtOut e = s_res[lane];
        T4OverlapOut t;
        t.seqIdx = in.seqIdx; t.readStart = e.rs; t.readEnd = e.re; t.seqStart = e.ss; t.seqEnd = e.se;
        t.strand = in.strand; t.matchCnt = e.matchCnt;
        if (e.simFail) { t.indelCnt = in.indelCnt; t.similarity = in.similarity; }
        else { t.indelCnt = 0; t.similarity = (double)t.matchCnt / (double)e.den; }
        qa.outExt[i0 + lane] = t;
        qa.ret[i0 + lane] = e.ret;
      }
      if (lane == 0 && s_ws.unsupported) qa.poolCursor[1] = 1u;   // an overhang beyond the direction buffer (reads of more than 600 bases)
      __syncthreads();
      i0 = i1;
    }
  }
}

// SeqSet::RecomputePosWeight (SeqSet.hpp:4705-4738) -- the posWeight columns of every contig rebuilt from the reads assigned to it
// (SeqSet::AssignRead results): UpdatePosWeightFromRead (SeqSet.hpp:2466-2474) of every assigned read, on the strand of its
// assignment, then count 1 on the consensus base of every column no read covers. One wavefront per read, a lane per base, integer
// atomics on the 4 x int32 columns (the columns of a deep clone are hot, the adds of one read are to distinct addresses);
// `mult` = number of identical reads the entry stands for (main.cpp:2080 assigns identical consecutive reads once).
__global__ __launch_bounds__(64) void posWeightAccumulateKernel(T4IndexView ix, T4BatchView bv, const T4OverlapOut *assign, const int *mult, int *counts) {
  const int lane = threadIdx.x;
  for (long long r = blockIdx.x; r < bv.n; r += gridDim.x) {
    const T4OverlapOut a = assign[r];
    if (a.seqIdx < 0 || a.seqIdx >= ix.nseq) continue;
    const T4SeqInfo sq = ix.seqs[a.seqIdx];
    if (sq.pwOff < 0) continue;
    const int len = bv.len[r], m = mult ? mult[r] : 1;
    const unsigned *pk = bv.pk + r * bv.wpk, *nm = bv.nm + r * bv.wnm;
    for (int j = lane; j < len; j += 64) {
      const int p = a.strand == 1 ? j : len - 1 - j;   // base j of the strand that was assigned
      if ((nm[p >> 5] >> (p & 31)) & 1u) continue;     // 'N' adds nothing
      int code = (int)((pk[p >> 4] >> ((p & 15) * 2)) & 3u);
      if (a.strand != 1) code = 3 - code;
      const int col = a.seqStart + j;
      if (col < 0 || col >= sq.len) continue;          // an assignment covers the whole read inside the contig; nothing to add outside it
      atomicAdd(&counts[((long long)sq.pwOff + col) * 4 + code], m);
    }
  }
}
__global__ __launch_bounds__(256) void posWeightFinishKernel(T4IndexView ix, int *counts) {
  for (int c = blockIdx.x; c < ix.nseq; c += gridDim.x) {
    const T4SeqInfo sq = ix.seqs[c];
    if (sq.pwOff < 0) continue;
    for (int j = threadIdx.x; j < sq.len; j += blockDim.x) {
      int *w = counts + ((long long)sq.pwOff + j) * 4;
      const char ch = ix.cons[sq.consOff + j];
      if (ch != 'N' && w[0] + w[1] + w[2] + w[3] == 0) w[nuc2(ch)] = 1;
    }
  }
}

// SeqSet::UpdateConsensus (SeqSet.hpp:4537-4588) of every contig from posWeight columns laid out like the image's (T4SeqInfo::pwOff):
// the base with the largest count (the first of equals) replaces the consensus base when that one is strictly rarer; columns without
// any count keep their base; an 'N' stands for base 0 as in the reference's nucToNum table. consOut: the column space of `counts`.
// Elementwise, 17 bytes read and one written per column.
__global__ __launch_bounds__(256) void consensusArgmaxKernel(T4IndexView ix, const int *counts, char *consOut, unsigned long long *changed) {
  unsigned long long mine = 0;
  for (int c = blockIdx.x; c < ix.nseq; c += gridDim.x) {
    const T4SeqInfo sq = ix.seqs[c];
    if (sq.pwOff < 0) continue;
    for (int j = threadIdx.x; j < sq.len; j += blockDim.x) {
      const int4 w = *(const int4 *)(counts + ((long long)sq.pwOff + j) * 4);
      const char ch = ix.cons[sq.consOff + j];
      int mx = 0, tag = 0;
      if (w.x > mx) { mx = w.x; tag = 0; }
      if (w.y > mx) { mx = w.y; tag = 1; }
      if (w.z > mx) { mx = w.z; tag = 2; }
      if (w.w > mx) { mx = w.w; tag = 3; }
      const int cur = ch == 'N' ? 0 : nuc2(ch);
      const int curCnt = cur == 0 ? w.x : cur == 1 ? w.y : cur == 2 ? w.z : w.w;
      char out = ch;
      if (mx > 0 && cur != tag && curCnt < mx) { out = tag == 0 ? 'A' : tag == 1 ? 'C' : tag == 2 ? 'G' : 'T'; ++mine; }
      consOut[(long long)sq.pwOff + j] = out;
    }
  }
  if (mine) atomicAdd(changed, mine);
}

// Scatter of freshly built per-barcode set images from the staging buffer to their slots (16-byte units).
__global__ __launch_bounds__(256) void scatterKernel(const unsigned char *staging, const T4CopyDesc *desc, int nDesc) {
  for (int d = blockIdx.x; d < nDesc; d += gridDim.x) {
    const T4CopyDesc cd = desc[d];
    const uint4 *src = (const uint4 *)(staging + cd.srcOff);
    uint4 *dst = (uint4 *)cd.dst;
    const unsigned long long n16 = cd.bytes >> 4;
    for (unsigned long long i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
  }
}

// Patches of a live set's image (t4_index_apply_delta): run d copies bytes staging[srcOff ..) -> dst, one wavefront per run.
// Runs never overlap (every destination is written once per delta).
__global__ __launch_bounds__(256) void deltaKernel(const unsigned char *staging, const T4CopyDesc *desc, int nDesc) {
  const int wavesPerBlock = blockDim.x >> 6, lane = threadIdx.x & 63;
  for (int d = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6); d < nDesc; d += gridDim.x * wavesPerBlock) {
    const T4CopyDesc cd = desc[d];
    const unsigned char *src = staging + cd.srcOff;
    unsigned char *dst = cd.dst;
    if ((((unsigned long long)dst | (unsigned long long)src | cd.bytes) & 7ull) == 0) {
      const unsigned long long n8 = cd.bytes >> 3;
      for (unsigned long long i = lane; i < n8; i += 64) ((unsigned long long *)dst)[i] = ((const unsigned long long *)src)[i];
    } else
      for (unsigned long long i = lane; i < cd.bytes; i += 64) dst[i] = src[i];
  }
}

// ---- one AddRead query call without a copy engine (round 6). A round of the ordered builder is a chain of dependent stream
// operations a few tens of microseconds long each; an H2D copy, a memset and a D2H copy between the kernels were half of them, and
// every hop between the compute queue and a copy engine costs more than a kernel boundary. The call's input blob sits in pinned
// host memory the device can read, its header goes back into pinned host memory the device can write:
//   aqPrologueKernel: input blob host -> device (the query kernels read it many times: device memory), header block zeroed
//   aqEpilogueKernel: header block device -> host, then ONE word -- the call's sequence number -- that the host polls for
// (the results themselves are written into pinned pools by the kernels that make them, as before).
__global__ __launch_bounds__(256) void aqPrologueKernel(const unsigned long long *hostIn, unsigned long long *devIn, unsigned long long inWords,
                                                        unsigned long long *devOut, unsigned long long outWords) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = t; i < inWords; i += nt) devIn[i] = hostIn[i];
  for (unsigned long long i = t; i < outWords; i += nt) devOut[i] = 0ull;
}
// `done` counts the blocks of this launch that have written their share (device memory, left at zero by the last block)
__global__ __launch_bounds__(256) void aqEpilogueKernel(const unsigned long long *devOut, unsigned long long *hostOut, unsigned long long outWords,
                                                        unsigned *done, unsigned *hostFlag, unsigned seq) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = t; i < outWords; i += nt) hostOut[i] = devOut[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = atomicAdd(done, 1u);
    if (before + 1u == gridDim.x) {
      *done = 0u;
      __threadfence_system();
      __atomic_store_n(hostFlag, seq, __ATOMIC_RELEASE);
    }
  }
}

// IsBaseEqual flips of resident per-barcode images (the common change between two queries of a cell): single bytes
__global__ __launch_bounds__(256) void patchKernel(const T4BytePatch *patches, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) *patches[i].dst = (unsigned char)patches[i].val;
}

// Tiering: estimate H of every read (whole read, both strands) and bin the reads by capacity.
__global__ __launch_bounds__(64) void binKernel(T4IndexView ix, T4BatchView bv, int useBarcode, T4TierCaps caps,
                                               int *lists, int *counts, long long listStride) {
  __shared__ unsigned s_posStart[T4_MAXPOS + 8];
  __shared__ unsigned s_posPref[T4_MAXPOS + 8];
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ int s_red[16];
  // Reads are appended to their tier's list 64 at a time: one atomicAdd per read on the same few counters bounded this kernel
  // (2 M same-address atomics took ~20 of its 23.5 ms); the order inside a list is not observable.
  __shared__ int s_buf[T4_NTIER][64];
  __shared__ int s_cnt[T4_NTIER];
  WaveMem wm;
  wm.seg = s_seg; wm.rc = s_rc;
  if (laneId() < T4_NTIER) s_cnt[laneId()] = 0;
  __syncthreads();
  for (long long r = blockIdx.x; r < bv.n; r += gridDim.x) {
    int len = bv.len[r];
    int H = 0;
    if (len >= ix.k) {
      loadSegment(bv, r, 0, len, wm);
      int barcode = (useBarcode && bv.barcode) ? bv.barcode[r] : -1;
      H = seedPositions(ix, wm, len, 0, barcode, false, s_posStart, s_posPref, s_red);
    }
    H = __shfl(H, 0);
    int t = 0;
    while (t < T4_NTIER - 1 && H > caps.cap[t]) ++t;
    if (laneId() == 0) { s_buf[t][s_cnt[t]] = (int)r; ++s_cnt[t]; }
    __syncthreads();
    const int filled = s_cnt[t];
    __syncthreads();        // everyone has read the count before lane 0 resets it
    if (filled == 64) {     // uniform
      int base = 0;
      if (laneId() == 0) { base = atomicAdd(&counts[t], 64); s_cnt[t] = 0; }
      base = __shfl(base, 0);
      lists[t * listStride + base + laneId()] = s_buf[t][laneId()];
      __syncthreads();
    }
  }
  for (int t = 0; t < T4_NTIER; ++t) {
    const int n = s_cnt[t];
    if (n == 0) continue;   // uniform
    int base = 0;
    if (laneId() == 0) base = atomicAdd(&counts[t], n);
    base = __shfl(base, 0);
    if (laneId() < n) lists[t * listStride + base + laneId()] = s_buf[t][laneId()];
  }
}

// ---- KmerCount (KmerCount.hpp): canonical k-mer counts of a read set and the per-read count statistics -----------------
__device__ __forceinline__ unsigned long long kcMix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
// canonical code of the k-mer at position p of the read in wm.seg / wm.rc (KmerCode::GetCanonicalKmerCode, KmerCode.hpp:54-67);
// valid == KmerCode::IsValid: no N among the last k characters (Append, KmerCode.hpp:94-109)
__device__ __forceinline__ unsigned long long canonicalAt(const WaveMem &wm, int len, int p, int K, bool &valid) {
  bool v2;
  const unsigned long long f = kmerAt(wm.seg, p, K, valid), c = kmerAt(wm.rc, len - p - K, K, v2);
  return c < f ? c : f;
}

// KmerCount::AddCount (KmerCount.hpp:64-97) for every read of the batch; one wavefront per read, a lane per position.
__global__ __launch_bounds__(64) void kmerAddKernel(T4BatchView bv, T4KmerTable tb, long long rBegin, long long rEnd) {
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  WaveMem wm;
  wm.seg = s_seg; wm.rc = s_rc;
  for (long long r = rBegin + blockIdx.x; r < rEnd; r += gridDim.x) {
    const int len = bv.len[r];
    if (len >= tb.k) {   // block-uniform
      loadSegment(bv, r, 0, len, wm);
      const unsigned long long salt = (tb.perBarcode && bv.barcode) ? ((unsigned long long)(bv.barcode[r] + 1) << 42) : 0ull;
      for (int p = laneId(); p + tb.k <= len; p += 64) {
        bool valid;
        const unsigned long long kc = canonicalAt(wm, len, p, tb.k, valid) | salt;
        if (!valid) continue;
        unsigned long long h = kcMix(kc) & tb.mask, probes = 0;
        bool fresh = false;
        for (; probes <= tb.mask; ++probes) {
          const unsigned long long old = atomicCAS(&tb.keys[h], 0ull, kc + 1ull);
          if (old == 0ull || old == kc + 1ull) { atomicAdd(&tb.cnt[h], 1u); fresh = old == 0ull; break; }
          h = (h + 1ull) & tb.mask;
        }
        if (probes > tb.mask) *tb.overflow = 1;
        if (fresh) atomicAdd(tb.used, 1ull);   // occupied slots (every k-mer is new at first, nearly none later; a wave-wide vote here would sit in divergent code)
      }
    }
    __syncthreads();
  }
}

// KmerCount::AddCountFromFile (KmerCount.hpp:99-120): counts given for k-mer codes as they were written (the host has already let
// the later of two records of one k-mer win): count[code] = value.
__global__ __launch_bounds__(256) void kmerSetKernel(T4KmerTable tb, const unsigned long long *codes, const int *counts, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long kc = codes[i];
    unsigned long long h = kcMix(kc) & tb.mask, probes = 0;
    for (; probes <= tb.mask; ++probes) {
      const unsigned long long old = atomicCAS(&tb.keys[h], 0ull, kc + 1ull);
      if (old == 0ull || old == kc + 1ull) { tb.cnt[h] = (unsigned)counts[i]; if (old == 0ull) atomicAdd(tb.used, 1ull); break; }
      h = (h + 1ull) & tb.mask;
    }
    if (probes > tb.mask) *tb.overflow = 1;
  }
}

// KmerCount::GetCountStatsAndTrim (KmerCount.hpp:177-288) for every read of the batch: min / median / mean count of the read's
// valid k-mers (absent or non-positive counts read as 1), the quality trimming when quals != null (the read's qualities at
// quals + qoff[r], as long as the read), and the N rule on the minimum. lenOut = the length the read is cut to (0: emptied).
// One wavefront per read: lookups a lane per position, the order-dependent scans on lane 0, the median by ranks.
// every (k-mer, count) of a full-ish table into one four times as large (t4_kmer_count_add grows the table as it fills)
__global__ __launch_bounds__(256) void kmerRehashKernel(T4KmerTable from, T4KmerTable to) {
  for (unsigned long long s = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; s <= from.mask; s += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long key = from.keys[s];
    if (!key) continue;
    unsigned long long h = kcMix(key - 1ull) & to.mask;
    for (;;) {
      const unsigned long long old = atomicCAS(&to.keys[h], 0ull, key);
      if (old == 0ull || old == key) { atomicAdd(&to.cnt[h], from.cnt[s]); break; }
      h = (h + 1ull) & to.mask;
    }
  }
}

// every (k-mer key, count) pair of the table side by side, in any order (t4_kmer_count_export: what a rank of a run whose input is
// dealt out by cells hands the other ranks). cursor counts the pairs; only the first cap of them are written.
__global__ __launch_bounds__(256) void kmerExportKernel(T4KmerTable tb, unsigned long long *codes, int *counts, unsigned long long *cursor, unsigned long long cap) {
  for (unsigned long long s = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; s <= tb.mask; s += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long key = tb.keys[s];
    if (!key) continue;
    const unsigned long long at = atomicAdd(cursor, 1ull);
    if (at < cap) { codes[at] = key - 1ull; counts[at] = (int)tb.cnt[s]; }
  }
}

// The counts of another read set added to this table's: KmerCount::AddCount (KmerCount.hpp:64-97) only ever increments, so the counts
// of a union of read sets are the sums of the sets' counts. onlyPresent: a pair whose k-mer this table does not hold is passed over
// (a rank of a sharded run looks up the k-mers of its own reads only, and those are all here).
__global__ __launch_bounds__(256) void kmerMergeKernel(T4KmerTable tb, const unsigned long long *codes, const int *counts, long long n, int onlyPresent) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long key = codes[i] + 1ull;
    unsigned long long h = kcMix(codes[i]) & tb.mask, probes = 0;
    for (; probes <= tb.mask; ++probes) {
      if (onlyPresent) {
        const unsigned long long cur = tb.keys[h];
        if (cur == 0ull) break;
        if (cur == key) { atomicAdd(&tb.cnt[h], (unsigned)counts[i]); break; }
      } else {
        const unsigned long long old = atomicCAS(&tb.keys[h], 0ull, key);
        if (old == 0ull || old == key) { atomicAdd(&tb.cnt[h], (unsigned)counts[i]); if (old == 0ull) atomicAdd(tb.used, 1ull); break; }
      }
      h = (h + 1ull) & tb.mask;
    }
    if (!onlyPresent && probes > tb.mask) *tb.overflow = 1;
  }
}

__global__ __launch_bounds__(64) void kmerStatsKernel(T4BatchView bv, T4KmerTable tb, const char *quals, const long long *qoff,
                                                     int *minOut, int *medOut, float *avgOut, int *lenOut) {
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ int s_c[T4_MAXL + 8];     // count per position, -1 = no valid k-mer there
  __shared__ int s_v[T4_MAXL + 8];     // the reference's c[]: counts of the valid k-mers in read order
  __shared__ int s_k, s_sum, s_nul0, s_nul1, s_newLen, s_med;
  WaveMem wm;
  wm.seg = s_seg; wm.rc = s_rc;
  const int K = tb.k;
  for (long long r = blockIdx.x; r < bv.n; r += gridDim.x) {
    const int len = bv.len[r];
    if (len < K) {   // block-uniform
      if (laneId() == 0) { minOut[r] = -1; medOut[r] = -1; avgOut[r] = -1.0f; lenOut[r] = len; }
      continue;
    }
    loadSegment(bv, r, 0, len, wm);
    const unsigned long long salt = (tb.perBarcode && bv.barcode) ? ((unsigned long long)(bv.barcode[r] + 1) << 42) : 0ull;
    for (int p = laneId(); p + K <= len; p += 64) {
      bool valid;
      const unsigned long long kc = canonicalAt(wm, len, p, K, valid) | salt;
      int v = -1;
      if (valid) {
        v = 0;
        unsigned long long h = kcMix(kc) & tb.mask;
        for (unsigned long long probes = 0; probes <= tb.mask; ++probes) {
          const unsigned long long key = tb.keys[h];
          if (key == 0ull) break;
          if (key == kc + 1ull) { v = (int)tb.cnt[h]; break; }
          h = (h + 1ull) & tb.mask;
        }
        if (v <= 0) v = 1;
      }
      s_c[p] = v;
    }
    for (int p = laneId(); p < len + 1; p += 64) s_v[p] = 0;   // entries the read never writes count as 0 (see the oracle)
    __syncthreads();
    if (laneId() == 0) {
      int n = 0, sum = 0;
      for (int p = 0; p + K <= len; ++p) if (s_c[p] >= 0) { s_v[n++] = s_c[p]; sum += s_c[p]; }
      int kk = n, nul0 = -1, nul1 = -1, newLen = len;
      if (n == 0) { kk = -1; if (quals) newLen = 0; }
      else if (quals) {
        const char *q = quals + qoff[r];
        int i;
        for (i = n - 1; i >= 0; --i) if (s_v[i] > 1) break;
        ++i;
        int badCnt = 0, trimStart = -1;
        for (int j = len - 1; j >= i + K - 1; --j)
          if (q[j] - 32 <= 15) { ++badCnt; if (badCnt >= 0.1 * (len - j)) trimStart = j; }
        if (trimStart > 0) { kk = trimStart - K + 1; newLen = trimStart; nul0 = trimStart; }
        if (trimStart > 0 && trimStart < K) { kk = 0; newLen = 0; nul1 = 0; }
      }
      s_k = kk; s_sum = sum; s_nul0 = nul0; s_nul1 = nul1; s_newLen = newLen; s_med = s_v[0];
    }
    __syncthreads();
    const int kk = s_k;
    if (kk > 0) {   // median = the element of rank kk / 2 of std::sort(c, c + kk)
      for (int j = laneId(); j < kk; j += 64) {
        const int v = s_v[j];
        int rank = 0;
        for (int x = 0; x < kk; ++x) { const int w = s_v[x]; rank += (w < v || (w == v && x < j)) ? 1 : 0; }
        if (rank == kk / 2) s_med = v;
      }
    }
    __syncthreads();
    if (laneId() == 0) {
      if (kk < 0) { minOut[r] = -len; medOut[r] = -len; avgOut[r] = (float)-len; lenOut[r] = s_newLen; }
      else {
        int mn = s_v[0];
        for (int x = 1; x < kk; ++x) if (s_v[x] < mn) mn = s_v[x];
        for (int i = 0; i < len; ++i)   // the reference scans its old buffer, in which the trim wrote NUL at nul0 / nul1
          if (i != s_nul0 && i != s_nul1 && s_seg[i] == 'N') { if (mn >= 0) mn = 0; else if (mn <= 0) --mn; }
        minOut[r] = mn; medOut[r] = s_med; avgOut[r] = (float)((double)s_sum / (double)kk); lenOut[r] = s_newLen;
      }
    }
    __syncthreads();
  }
}

// t4_hits: GetHitsFromRead + SortHits. pass 0 counts the hits per read, pass 1 writes them.
__global__ __launch_bounds__(64) void hitsKernel(T4IndexView ix, T4BatchView bv, int strandArg, int allowTotalSkip, int pass,
                                                long long *offsets, T4HitOut *out, unsigned long long *gKeys, int gCap, int *status) {
  __shared__ unsigned s_posStart[T4_MAXPOS + 8];
  __shared__ unsigned s_posPref[T4_MAXPOS + 8];
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ int s_red[16];
  WaveMem wm;
  wm.seg = s_seg; wm.rc = s_rc;
  wm.keys = gKeys + (size_t)blockIdx.x * gCap;
  const int lane = laneId();
  for (long long r = blockIdx.x; r < bv.n; r += gridDim.x) {
    int len = bv.len[r];
    int barcode = bv.barcode ? bv.barcode[r] : -1;
    int Hv = 0;
    if (len >= 1 && len >= ix.k) {
      loadSegment(bv, r, 0, len, wm);
      int nk = len - ix.k + 1;
      int H = seedPositions(ix, wm, len, strandArg, barcode, allowTotalSkip != 0, s_posStart, s_posPref, s_red);
      if (H > gCap) { if (lane == 0) status[r] = 2; H = 0; }
      // keys ordered as _hit::operator< : (strand, idx, readOffset, offset)
      int dropped = 0;
      for (int s = lane; s < H; s += 64) {
        int lo = 0, hi = 2 * nk - 1;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (s_posPref[mid] <= (unsigned)s) lo = mid; else hi = mid - 1; }
        int q = lo;
        int2 po = ix.post[s_posStart[q] + ((unsigned)s - s_posPref[q])];
        int st = q >= nk, a = st ? q - nk : q;
        unsigned long long key = ~0ull;
        if (barcode != -1 && ix.seqs[po.x].barcode != barcode) ++dropped;
        else key = ((st ? 0ull : 1ull) << 63) | ((unsigned long long)po.x << 32) | ((unsigned long long)a << 20) | (unsigned long long)po.y;
        wm.keys[s] = key;
      }
      dropped = blockSum(dropped, s_red);
      Hv = H - dropped;
      if (pass == 1 && H > 0) {
        __syncthreads();
        if (H > 1) bitonicSort(wm.keys, H);
        __syncthreads();
        long long base = offsets[r];
        for (int s = lane; s < Hv; s += 64) {
          unsigned long long key = wm.keys[s];
          T4HitOut h;
          h.idx = (int)((key >> 32) & 0x3FFFFFull);
          h.readOffset = (int)((key >> 20) & 0xFFF);
          h.offset = (int)(key & 0xFFFFF);
          h.strand = (key >> 63) ? 1 : -1;
          int q = (h.strand == 1 ? 0 : nk) + h.readOffset;
          h.repeats = barcode != -1 ? 1 : (int)(s_posPref[q + 1] - s_posPref[q]);
          out[base + s] = h;
        }
      }
    }
    if (pass == 0 && lane == 0) offsets[r + 1] = Hv;
    __syncthreads();
  }
}


// AlignAlgo::IsMateOverlap (AlignAlgo.hpp:1027-1096) for a batch of pairs, one pair per wavefront: does a suffix of `fr`
// match a prefix of `sr` at exactly one offset? Offset j passes iff the mismatches over the compared stretch stay within
// (flen - j) - int((flen - j) * thr(flen - j)) -- the reference's running test `matchCnt + rest < need` is monotone in the
// mismatch count, so it equals the test on the total. Lanes take the offsets; the reference keeps the LAST passing offset.
// out: 3 ints per pair (return value: overlap size or -1, offset, bestMatchCnt; the latter two are the reference's outputs
// whenever at least one offset passed, else -1).
// (one wavefront; fr / sr in LDS; every lane returns the same values)
__device__ int mateOverlapWave(const char *s_f, int flen, const char *s_s, int slen, int mo, bool checkTandem, int &offset, int &bestMatch) {
  const int lane = laneId();
  int cnt = 0, lastJ = -1, lastMatch = -1, lastSize = -1;
  for (int j = lane; j < flen - mo; j += 64) {
    const int rem = flen - j;
    double thr = 0.95;
    if (rem >= 100) thr = 0.85; else if (rem >= 50) thr = 0.85 + (rem - 50) / 50.0 * 0.1;
    const int need = (int)(rem * thr), kEnd = rem < slen ? rem : slen;
    int match = 0;
    for (int k = 0; k < kEnd; ++k) match += s_f[j + k] == s_s[k] ? 1 : 0;
    // the reference stops at the first k with match(0..k) + (rem - k - 1) < need, i.e. as soon as mismatches exceed rem - need
    if (kEnd - match <= rem - need) { ++cnt; lastJ = j; lastMatch = match; lastSize = kEnd; }
  }
  const int total = waveSum(cnt);
  // the last passing offset = the largest j: lanes hold increasing j's in their own stride, so take the maximum
  int best = lastJ;
  for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(best, o); if (v > best) best = v; }
  const int src = __ffsll((long long)__ballot(lastJ == best && best >= 0)) - 1;
  const int bMatch = __shfl(lastMatch, src < 0 ? 0 : src), bSize = __shfl(lastSize, src < 0 ? 0 : src);
  int ret = -1;
  if (lane == 0 && total == 1) {
    ret = bSize;
    if (checkTandem && bSize <= mo * 2) {
      for (int i = 1; i <= bSize / 2 && ret >= 0; ++i) {
        bool tandem = true;
        for (int j = i; j + i - 1 < bSize && tandem; j += i)
          for (int k = j; k <= j + i - 1; ++k) if (s_s[k - j] != s_s[k]) { tandem = false; break; }
        if (tandem) ret = -1;
      }
    }
  }
  offset = total > 0 ? best : -1;
  bestMatch = total > 0 ? bMatch : -1;
  return __shfl(ret, 0);
}
__global__ __launch_bounds__(64) void mateOverlapKernel(int n, const long long *fOff, const char *fChars, const long long *sOff,
                                                      const char *sChars, const int *minOverlap, int checkTandem, int *out) {
  __shared__ char s_f[T4_MAXL + 8], s_s[T4_MAXL + 8];
  const int lane = laneId();
  for (int p = blockIdx.x; p < n; p += gridDim.x) {
    const int flen = (int)(fOff[p + 1] - fOff[p]), slen = (int)(sOff[p + 1] - sOff[p]), mo = minOverlap[p];
    if (flen > T4_MAXL || slen > T4_MAXL) { if (lane == 0) { out[3 * p] = -2; out[3 * p + 1] = -1; out[3 * p + 2] = -1; } continue; }
    for (int i = lane; i < flen; i += 64) s_f[i] = fChars[fOff[p] + i];
    for (int i = lane; i < slen; i += 64) s_s[i] = sChars[sOff[p] + i];
    __syncthreads();
    int offset, bestMatch;
    const int ret = mateOverlapWave(s_f, flen, s_s, slen, mo, checkTandem != 0, offset, bestMatch);
    if (lane == 0) { out[3 * p] = ret; out[3 * p + 1] = offset; out[3 * p + 2] = bestMatch; }
    __syncthreads();
  }
}

// ProcessRead (main.cpp:224-449) of a batch of mate pairs, one pair per wavefront: read 2 is reverse-complemented; if it runs
// through read 1 (IsMateOverlap of rc(read 2) against read 1) read 1 is cut to the overlap and takes the better-quality bases;
// else if the mates overlap at their ends (IsMateOverlap of read 1 against rc(read 2), tandem repeats refused) they are merged
// into one read of weight 2 when nearly all overlapped bases agree, or the mate of better quality stands for both; else both stay.
// IsLowComplexity (main.cpp:183-205) of what is left. r1 / r2: chars as read (any letter), q1 / q2: their qualities (used when
// the pair's bit in hasQual says so: bit 0 read 1, bit 1 read 2). Per pair, meta = {kind, length of the new read 1, flags, 0}:
// kind 0 both mates stay as they are, 1 read-through, 2 merged, 3 one mate stands for both; flags: 1 read 1 kept (not of low
// complexity), 2 read 2 kept, 4 weight 2 (the driver lists the merged read twice), 8 read 1 has qualities, 16 read 1 changed: its
// new bases / qualities are in outR / outQ at outOff[p] (room for len1 + len2 + 1).
__device__ __forceinline__ char rcChar(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
__device__ bool lowComplexityWave(const char *s, int n) {
  int cA = 0, cC = 0, cG = 0, cT = 0, cN = 0;
  for (int i = laneId(); i < n; i += 64) { const char ch = s[i]; if (ch == 'N') ++cN; else if (ch == 'C') ++cC; else if (ch == 'G') ++cG; else if (ch == 'T') ++cT; else ++cA; }   // other letters count as base 0
  cA = waveSum(cA); cC = waveSum(cC); cG = waveSum(cG); cT = waveSum(cT); cN = waveSum(cN);
  if (cA >= n / 2 || cC >= n / 2 || cG >= n / 2 || cT >= n / 2 || cN >= n / 10) return true;
  return ((cA <= 2) + (cC <= 2) + (cG <= 2) + (cT <= 2)) >= 2;
}
__global__ __launch_bounds__(64) void processPairKernel(int n, const long long *off1, const char *r1, const char *q1, const long long *off2, const char *r2,
                                                      const char *q2, const unsigned char *hasQual, const long long *outOff, char *outR, char *outQ, int4 *meta) {
  __shared__ char s_r1[T4_MAXL + 8], s_q1[T4_MAXL + 8], s_f[T4_MAXL + 8], s_fq[T4_MAXL + 8];
  __shared__ char s_m[2 * T4_MAXL + 16], s_mq[2 * T4_MAXL + 16];
  const int lane = laneId();
  for (int p = blockIdx.x; p < n; p += gridDim.x) {
    const int slen = (int)(off1[p + 1] - off1[p]), flen = (int)(off2[p + 1] - off2[p]);
    if (flen > T4_MAXL || slen > T4_MAXL) { if (lane == 0) meta[p] = make_int4(-2, 0, 0, 0); continue; }
    const bool hq1 = (hasQual[p] & 1) != 0, hq2 = (hasQual[p] & 2) != 0;
    for (int i = lane; i < slen; i += 64) { s_r1[i] = r1[off1[p] + i]; s_q1[i] = hq1 ? q1[off1[p] + i] : 0; }
    for (int i = lane; i < flen; i += 64) { s_f[i] = rcChar(r2[off2[p] + flen - 1 - i]); s_fq[i] = hq2 ? q2[off2[p] + flen - 1 - i] : 0; }
    __syncthreads();
    int mo = (flen + slen) / 10, mo2 = (flen + slen) / 20;
    if (mo > 31) mo = 31;
    if (mo2 > 31) mo2 = 31;
    int kind = 0, outLen = slen, flags = hq1 ? 8 : 0, offset, best;
    bool r2Alive = true;
    const char *fin = s_r1;   // where the final read 1 stands
    int ov = mateOverlapWave(s_f, flen, s_r1, slen, mo, false, offset, best);
    if (ov >= 0) {
      kind = 1; outLen = ov; r2Alive = false; flags |= 16;
      for (int j = lane; j < ov; j += 64) {
        char b = s_r1[j], q = s_q1[j];
        if (hq1 && (s_fq[j + offset] > q || b == 'N')) { b = s_f[j + offset]; q = s_fq[j + offset]; }
        s_m[j] = b; s_mq[j] = q;
      }
      fin = s_m;
    } else if ((ov = mateOverlapWave(s_r1, slen, s_f, flen, mo2, true, offset, best)) >= 0) {
      r2Alive = false;
      if ((double)best >= 0.95 * (double)ov) {
        kind = 2; flags |= 4 | 16;
        const int len = offset + flen;
        for (int j = lane; j < len; j += 64) {
          char b = j >= offset ? s_f[j - offset] : 0, q = j >= offset ? s_fq[j - offset] : 0;   // qualities of a mate without any read as 0
          if (j < slen && (j < offset || (int)s_q1[j] >= (int)q - 14 || b == 'N')) { b = s_r1[j]; q = s_q1[j]; }
          s_m[j] = b; s_mq[j] = q;
        }
        outLen = len; fin = s_m;
      } else {
        kind = 3;
        bool useFirst = true;
        if (hq1) {
          int a = 0, b = 0;
          for (int j = offset + lane; j < slen; j += 64) a += (int)s_q1[j] - 32;
          for (int j = flen - 1 - lane; j >= flen - ov; j -= 64) b += (int)s_fq[j] - 32;
          a = waveSum(a); b = waveSum(b);
          double da = (double)a, db = (double)b;
          da /= ov; db /= ov;
          if (da + 10 < db) useFirst = false;
        }
        if (!useFirst) {   // read 2 as it was read, with the qualities in the order of its reverse complement (as the reference leaves them)
          flags = (flags & ~8) | (hq2 ? 8 : 0) | 16;
          for (int j = lane; j < flen; j += 64) { s_m[j] = rcChar(s_f[flen - 1 - j]); s_mq[j] = s_fq[j]; }
          outLen = flen; fin = s_m;
        }
      }
    }
    __syncthreads();
    if (!lowComplexityWave(fin, outLen)) flags |= 1;
    if (r2Alive && !lowComplexityWave(s_f, flen)) flags |= 2;
    if (flags & 16) for (int j = lane; j < outLen; j += 64) { outR[outOff[p] + j] = s_m[j]; outQ[outOff[p] + j] = s_mq[j]; }
    if (lane == 0) meta[p] = make_int4(kind, outLen, flags, 0);
    __syncthreads();
  }
}

// t4_gap_dp: a batch of independent gap alignments, one per lane. kind 0: AlignAlgo::GlobalAlignment on
// chars, kind 1: GlobalAlignment_PosWeight on weights. impl 0: forward LDS version with scratch fallback
// (what overlap scoring uses), impl 1: scratch + traceback version only. out: 3 ints per problem.
__global__ __launch_bounds__(64) void gapDpKernel(int kind, int impl, int n, const long long *tOff, const long long *pOff,
                                                 const char *tChars, const T4PW *tW, const char *pChars, int *out,
                                                 int *dpRows, unsigned char *dpDir, signed char *alignOut, int alignStride) {
  __shared__ int s_slots[4 * T4_DPW * 64];   // 32 KiB
  __shared__ char s_p[64][T4_MAXGAP + 8];
  const int lane = laneId();
  DPScratch sc;
  sc.rows = dpRows + (size_t)blockIdx.x * (6 * T4_ROWW * 64);
  sc.dir = dpDir + ((size_t)blockIdx.x * 64 + lane) * T4_DIR_BYTES;
  if (impl == 3) {   // eight alignments per wavefront, one per group of 8 lanes; bands wider than 16 columns report status 2
    const int grp = lane >> 3;
    for (int i0 = blockIdx.x * 8; i0 < n; i0 += gridDim.x * 8) {
      const int i = i0 + grp;
      const bool has = i < n;
      int lent = 0, lenp = 0;
      if (has) { lent = (int)(tOff[i + 1] - tOff[i]); lenp = (int)(pOff[i + 1] - pOff[i]); }
      const bool fits = has && lenp <= T4_MAXGAP && lent <= T4_MAXGAP;
      if (fits) for (int j = lane & 7; j < lenp; j += 8) s_p[grp][j] = pChars[pOff[i] + j];
      __syncthreads();
      unsigned c = kind == 0 ? dpOct<false, true>(fits, tChars + (has ? tOff[i] : 0), (const T4PW *)0, lent, s_p[grp], lenp, s_p[8 + grp], T4_MAXGAP)
                             : dpOct<true, true>(fits, (const char *)0, tW + (has ? tOff[i] : 0), lent, s_p[grp], lenp, s_p[8 + grp], T4_MAXGAP);
      if (!fits) c = DP_FAIL;
      __syncthreads();
      if (has && (lane & 7) == 0) {
        if (c == DP_FAIL) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = 0; out[4 * i + 3] = 2; }
        else { out[4 * i] = (int)(c & 1023u); out[4 * i + 1] = (int)((c >> 10) & 1023u); out[4 * i + 2] = (int)(c >> 20); out[4 * i + 3] = 0; }
      }
    }
    return;
  }
  if (impl == 2) {   // the wave-cooperative formulation: one alignment per wavefront
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
      int lent = (int)(tOff[i + 1] - tOff[i]), lenp = (int)(pOff[i + 1] - pOff[i]);
      unsigned c = DP_FAIL;
      if (lenp <= T4_MAXGAP) {
        for (int j = lane; j < lenp; j += 64) s_p[0][j] = pChars[pOff[i] + j];
        __syncthreads();
        c = kind == 0 ? dpWave<false>(tChars + tOff[i], (const T4PW *)0, lent, s_p[0], lenp, s_p[1])
                      : dpWave<true>((const char *)0, tW + tOff[i], lent, s_p[0], lenp, s_p[1]);
        __syncthreads();
      }
      if (lane == 0) {
        if (c == DP_FAIL) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = 0; out[4 * i + 3] = 2; }   // 2: band wider than a wavefront
        else { out[4 * i] = (int)(c & 1023u); out[4 * i + 1] = (int)((c >> 10) & 1023u); out[4 * i + 2] = (int)(c >> 20); out[4 * i + 3] = 0; }
      }
    }
    return;
  }
  for (int i0 = blockIdx.x * 64; i0 < n; i0 += gridDim.x * 64) {
    int i = i0 + lane;
    if (i < n) {
      int lent = (int)(tOff[i + 1] - tOff[i]), lenp = (int)(pOff[i + 1] - pOff[i]);
      int c0 = 0, c1 = 0, c2 = 0;
      bool ok = lenp <= T4_MAXGAP;
      if (ok) {
        for (int j = 0; j < lenp; ++j) s_p[lane][j] = pChars[pOff[i] + j];   // the read side lives in LDS in the real path
        bool done = false;
        if (impl == 0) done = kind == 0 ? dpAffineFwd(tChars + tOff[i], lent, s_p[lane], lenp, s_slots + lane, 64, c0, c1, c2)
                                        : dpPosWeightFwd(tW + tOff[i], lent, s_p[lane], lenp, s_slots + lane, 64, c0, c1, c2);
        // impl 4 (posWeight aligner only): the traceback's edit string itself (AlignAlgo.hpp:160-205; what ExtendOverlap reads) goes out too
        signed char *al = (impl == 4 && kind == 1 && alignOut && lent + lenp + 2 <= alignStride) ? alignOut + (size_t)i * alignStride : (signed char *)0;
        if (!done) ok = kind == 0 ? dpAffine(tChars + tOff[i], lent, s_p[lane], lenp, sc, lane, c0, c1, c2)
                                  : dpPosWeight(tW + tOff[i], lent, s_p[lane], lenp, sc, lane, c0, c1, c2, al);
      }
      out[4 * i] = c0; out[4 * i + 1] = c1; out[4 * i + 2] = c2; out[4 * i + 3] = ok ? 0 : 1;
    }
  }
}

}  // namespace t4k

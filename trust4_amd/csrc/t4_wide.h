// trust4_amd/csrc/t4_wide.h -- the wide AddRead query: one read spread over the chip (included by t4_kernels.h, namespace t4k).
//
// A read that lies inside a gene segment which thousands of contigs share (a constant-region read of a deep sample) meets every one
// of them: 10^5 - 10^7 k-mer hits, 10^3 - 10^5 candidate overlaps -- the reference walks them in one thread (SeqSet.hpp:763-1063,
// 1508-2124), one workgroup per read left 255 of 256 CUs idle while it sorted them. Here the (strand, contig) groups of
// GetOverlapsFromHits are what is distributed: they are independent up to the steps that look across contigs.
//
//   queryKernel (8192-hit LDS tier)  a read whose seed stage finds more hits than the tier holds is DEFERRED: wideDeferRead leaves its
//                                    seed table (posting range of every emitted k-mer) and a partition plan, nothing else
//   wideScatterKernel   all reads    postings -> 64-bit hit keys, appended to the partition of their contig (contig ranges of equal width)
//   wideSortKernel      a partition  LDS sort of its keys; sizes of its (strand, contig) groups
//   wideStatsKernel     a read       the group statistics of SeqSet.hpp:784-823 over the groups of all partitions in the reference's
//                                    order (incl. the `i = j; ++i` stepping), novelMinHitRequired, removeOnlyRepeats, the head of the
//                                    hit array that the run-relative `hits[k].repeats` test reads (934-940)
//   wideChainKernel     a partition  runs -> chains -> overlap records -> gap alignments (the same device functions the LDS tiers
//                                    run); its groups' dependency records (hits, hull of the diagonals with three or more hits)
//   wideMergeKernel     a read       std::sort of all records (1597), strand of the best (1601-1634), the order-dependent pre-filters
//                                    (1705-1794), the similarity cut (2105-2119), result records into the pool
//   extendKernel        (existing)   ExtendOverlap of the result records, eight per wavefront, all over the chip
//
// Every kernel is a persistent grid that reads its work from counters the kernels before it left in device memory (the host does
// not know which reads are heavy when it launches the round); capacity overflows raise a flag, the later kernels do nothing, and
// the host repeats the call with larger pools / finer partitions.
#pragma once

#define T4_WIDE_SEEDS (2 * T4_MAXL + 2)
#define T4_WIDE_OVBITS 9           // overlap records of one partition: 512

// ws->red[13..15] are free for this (the scans use the first eight words)
__device__ T4_NI void wideDeferRead(const T4IndexView &ix, WaveMem &wm, WaveState *ws, const T4Wide &wd, int len, int strandArg, long long r,
                                    unsigned long long &hitTotal) {
  const int lane = tid(), NT = nthr();
  unsigned *posStart = (unsigned *)wm.ov, *posPref = wm.pairs;
  const int nk = len - ix.k + 1;
  __syncthreads();
  const int H = seedPositionsNovel(ix, wm, len, strandArg, -1, false, posStart, posPref, ws->red, wm.keys, (WaveState *)0);
  hitTotal += (unsigned long long)H;
  int huge = 0;
  for (int q = lane; q < 2 * nk; q += NT) if (posPref[q + 1] - posPref[q] > 10000u) huge = 1;
  huge = blockSum(huge, ws->red) != 0;
  if (lane == 0) {
    int slot = atomicAdd(&wd.ctl[0], 1), pBase = 0, P = 0, Wd = 1;
    if (slot >= wd.maxReads) { atomicOr(&wd.ctl[2], 1); slot = -1; }
    else {
      const long long per = 16LL * wd.pcap;
      long long want = ((long long)H * wd.safetyNum + per - 1) / per;
      const int nseq = ix.nseq > 0 ? ix.nseq : 1;
      if (want < 1) want = 1;
      if (want > nseq) want = nseq;
      P = (int)want;
      if (P > wd.maxPartPerRead) { atomicOr(&wd.ctl[2], 2); P = 0; }
      else {
        pBase = atomicAdd(&wd.ctl[1], P);
        if (pBase + P > wd.maxPart) { atomicOr(&wd.ctl[2], 2); P = 0; }
      }
      T4WidePlan pl;
      pl.pBase = pBase; pl.P = P; pl.Wd = Wd; pl.nk = nk; pl.H = (unsigned)H; pl.huge = huge; pl.read = (int)r; pl.grpBase = 0;
      wd.plan[slot] = pl;
    }
    ws->red[15] = slot; ws->red[14] = pBase; ws->red[13] = P;
  }
  __syncthreads();
  const int slot = ws->red[15], pBase = ws->red[14], P = ws->red[13];
  if (slot >= 0 && P > 0) {
    // Partition boundaries from the hits themselves: contigs that carry one gene segment cluster in the id order (the clones of a
    // family are seeded one after the other), so contig ranges of equal width fill unevenly -- 38 % of the rounds of C3's first
    // 500 k pairs had to be repeated with finer partitions (profiles/r04l_*). A sample of the hits' contigs (every H / n-th posting in
    // list order), sorted; the partitions are its quantiles. A (strand, contig) group still lies in one partition.
    int *bd = wd.bounds + (size_t)slot * (T4_WIDE_MAXP + 1);
    unsigned *smp = (unsigned *)wm.keys;   // (the key array is free: the seed stage's code buffer died with it)
    int nS = 2 * wm.cap < 4096 ? 2 * wm.cap : 4096;
    if (wd.samplePerPart > 0) {   // (a read that plans four partitions does not need 4 096 samples to cut them: gathers and sort of the sample are on the round's critical path)
      int want = wd.samplePerPart * P;
      if (want < 512) want = 512;
      if (want < nS) nS = want;
    }
    if (nS > H) nS = H;
    if (P > 1 && nS > 0) {
      for (int j = lane; j < nS; j += NT) {
        const unsigned s = (unsigned)((long long)j * H / nS);
        int lo = 0, hi = 2 * nk - 1;   // last q with posPref[q] <= s
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (posPref[mid] <= s) lo = mid; else hi = mid - 1; }
        smp[j] = (unsigned)ix.post[posStart[lo] + (s - posPref[lo])].x;
      }
      __syncthreads();
      if (nS > 1) bitonicSort32(smp, nS);
      __syncthreads();
      for (int p = lane; p <= P; p += NT) bd[p] = p == 0 ? 0 : p == P ? 0x7FFFFFFF : (int)smp[(long long)p * nS / P];
    } else if (lane == 0) { bd[0] = 0; for (int p = 1; p <= P; ++p) bd[p] = 0x7FFFFFFF; }
    __syncthreads();
  }
  if (slot >= 0) {
    uint2 *sd = wd.seed + (size_t)slot * T4_WIDE_SEEDS;
    for (int q = lane; q <= 2 * nk; q += NT) { uint2 v; v.x = q < 2 * nk ? posStart[q] : 0u; v.y = posPref[q]; sd[q] = v; }   // (start, exclusive prefix); entry 2 nk holds H
    for (int p = lane; p < P; p += NT) { wd.pCnt[pBase + p] = 0u; wd.pRead[pBase + p] = slot; wd.pRecCnt[pBase + p] = 0; }
  }
  __syncthreads();
}

// Reads known to be heavy (the window entry's last query was served wide) skip the LDS tier: their seed stage runs here, on a
// second stream, and the wide kernels behind it run BESIDE the round's query kernel instead of after it.
__global__ __launch_bounds__(512) void wideSeedKernel(T4IndexView ixArg, T4BatchView bvArg, T4Work wk, T4QueryArgs qa, T4Wide wd, const int *list, int nList) {
  __shared__ unsigned long long s_code[2048];   // k-mer codes of the seed stage, then the sample of the hits' contigs (4096 x u32)
  __shared__ unsigned s_pref[2 * T4_MAXL + 2];
  __shared__ unsigned s_start[2 * T4_MAXL + 2];
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ WaveState s_ws;
  __shared__ T4IndexView s_ix;
  __shared__ T4BatchView s_bv;
  __shared__ WaveMem s_wm;
  __shared__ T4Wide s_wd;   // (by reference into wideDeferRead: as a by-value kernel parameter it was copied into EVERY lane's private stack -- 224 B
                            // per lane, 60 GB written and 31 GB fetched per config C2 run by the PMC counters, profiles/r06m_c2_pmc_*)
  if (threadIdx.x == 0) {
    s_ix = ixArg; s_bv = bvArg; s_wd = wd;
    WaveMem &m = s_wm;
    m.keys = s_code; m.pairs = s_pref; m.cand = s_pref; m.ov = (OvRec *)s_start; m.fin = (OvRec *)s_start; m.ord = (unsigned short *)s_start;
    m.cap = 2048; m.maxOv = 0; m.maxFin = 0; m.candCap = 0; m.ldsArrays = 1; m.hitLimit = 0;
    m.ldsSort = nullptr; m.ldsSortCap = 0; m.dirBuf = (unsigned char *)s_code; m.dirBytes = 0;
    m.seg = s_seg; m.rc = s_rc;
  }
  __syncthreads();
  for (int w = blockIdx.x; w < nList; w += gridDim.x) {
    const long long r = list[w];
    const int len = s_bv.len[r];
    __syncthreads();
    if (len < s_ix.k) { if (threadIdx.x == 0) qa.counts[r] = -1; continue; }   // (GetOverlapsFromRead's own answer: no k-mer)
    loadSegment(s_bv, r, 0, len, s_wm);
    unsigned long long hitTotal = 0;
    wideDeferRead(s_ix, s_wm, &s_ws, s_wd, len, qa.strandPerRead[r], r, hitTotal);
    if (threadIdx.x == 0) atomicAdd(wk.hitCounter, hitTotal);
  }
}

__device__ __forceinline__ int wideReads(const T4Wide &wd) { const int n = wd.ctl[0]; return n < wd.maxReads ? n : wd.maxReads; }
__device__ __forceinline__ int wideParts(const T4Wide &wd) { const int n = wd.ctl[1]; return n < wd.maxPart ? n : wd.maxPart; }

// postings -> keys -> partitions. A work item is a chunk of 1024 postings of one deferred read; the items of all reads form one list
// that the blocks stride over (a round's latency is what counts: every thread issues its four gathers before the first of the
// dependent atomics, and no block walks one read's chunks one after the other).
#define T4_WIDE_CH 1024u
__global__ __launch_bounds__(256) void wideScatterKernel(T4IndexView ix, T4Wide wd) {
  __shared__ unsigned s_pref[T4_WIDE_SEEDS];
  __shared__ unsigned s_start[T4_WIDE_SEEDS];
  __shared__ int s_bound[T4_WIDE_MAXP + 1];
  __shared__ int s_item[2];
  if (wd.ctl[2]) return;
  const int nR = wideReads(wd), lane = threadIdx.x, NT = blockDim.x;
  int loaded = -1;
  for (unsigned item = blockIdx.x;; item += gridDim.x) {
    __syncthreads();
    if (lane == 0) {   // which read, which of its chunks (a few dozen reads at most: a walk)
      unsigned left = item;
      int w = 0;
      for (; w < nR; ++w) {
        const T4WidePlan pw = wd.plan[w];
        const unsigned nCh = pw.P > 0 ? (pw.H + T4_WIDE_CH - 1u) / T4_WIDE_CH : 0u;
        if (left < nCh) break;
        left -= nCh;
      }
      s_item[0] = w < nR ? w : -1; s_item[1] = (int)left;
    }
    __syncthreads();
    const int w = s_item[0];
    if (w < 0) break;
    const unsigned chunk = (unsigned)s_item[1];
    const T4WidePlan pl = wd.plan[w];
    const int nq = 2 * pl.nk;
    if (w != loaded) {
      const uint2 *sd = wd.seed + (size_t)w * T4_WIDE_SEEDS;
      for (int q = lane; q <= nq; q += NT) { const uint2 v = sd[q]; s_start[q] = v.x; s_pref[q] = v.y; }
      const int *bd = wd.bounds + (size_t)w * (T4_WIDE_MAXP + 1);
      for (int p = lane; p <= pl.P; p += NT) s_bound[p] = bd[p];
      loaded = w;
      __syncthreads();
    }
    const unsigned s0 = chunk * T4_WIDE_CH, end = s0 + T4_WIDE_CH < pl.H ? s0 + T4_WIDE_CH : pl.H;
    unsigned long long key[4];
    int part[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned s = s0 + (unsigned)(t * 256 + lane);
      part[t] = -1; key[t] = 0;
      if (s < end) {
        int lo = 0, hi = nq - 1;   // last q with pref[q] <= s
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= s) lo = mid; else hi = mid - 1; }
        const int q = lo;
        const int2 po = ix.post[s_start[q] + (s - s_pref[q])];
        const int st = q >= pl.nk, a = st ? q - pl.nk : q;
        key[t] = ((st ? 0ull : 1ull) << 63) | ((unsigned long long)po.x << (T4_C_BITS + T4_B_BITS)) |
                 ((unsigned long long)(a - po.y + T4_C_BIAS) << T4_B_BITS) | (unsigned long long)po.y;
        int plo = 0, phi = pl.P - 1;   // last partition whose first contig is <= this one
        while (plo < phi) { const int mid = (plo + phi + 1) >> 1; if (s_bound[mid] <= po.x) plo = mid; else phi = mid - 1; }
        part[t] = plo;
      }
    }
    // one atomic per (wavefront, partition) instead of one per posting: consecutive postings of a list belong to neighbouring
    // contigs, i.e. mostly to one partition
    const int wl = lane & 63;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool has = part[t] >= 0;
      unsigned long long todo = __ballot(has);
      while (todo) {
        const int leader = __ffsll(todo) - 1;
        const int p0 = __shfl(part[t], leader);
        const unsigned long long same = __ballot(has && part[t] == p0);
        unsigned base = 0;
        if (wl == leader) base = atomicAdd(&wd.pCnt[pl.pBase + p0], (unsigned)__popcll(same));
        base = __shfl(base, leader);
        if (has && part[t] == p0) {
          const unsigned at = base + (unsigned)__popcll(same & ((1ull << wl) - 1ull));
          if (at < (unsigned)wd.pcap) wd.pKeys[(size_t)(pl.pBase + p0) * wd.pcap + at] = key[t];
          else atomicOr(&wd.ctl[2], 4);
        }
        todo &= ~same;
      }
    }
  }
}

// a partition: sort, group sizes, (huge reads) per group the hits whose list holds at most 10000 postings
template <int PCAP>
__global__ __launch_bounds__(512) void wideSortKernel(T4IndexView ix, T4Wide wd) {
  __shared__ unsigned long long s_keys[PCAP];
  __shared__ unsigned s_gs[PCAP + 1];
  __shared__ unsigned s_cnt[T4_WIDE_SEEDS];
  __shared__ int s_red[16];
  if (wd.ctl[2]) return;
  const int nPart = wideParts(wd), lane = threadIdx.x, NT = blockDim.x;
  for (int pg = blockIdx.x; pg < nPart; pg += gridDim.x) {
    const int w = wd.pRead[pg];
    const T4WidePlan pl = wd.plan[w];
    const int n = (int)wd.pCnt[pg];
    __syncthreads();
    unsigned long long *gk = wd.pKeys + (size_t)pg * wd.pcap;
    for (int i = lane; i < n; i += NT) s_keys[i] = gk[i];
    __syncthreads();
    if (n > 1) bitonicSortRegLds<unsigned long long>(s_keys, n);
    __syncthreads();
    for (int i = lane; i < n; i += NT) gk[i] = s_keys[i];
    // groups
    int nG = 0;
    for (int i0 = 0; i0 < n; i0 += NT) {
      const int i = i0 + lane;
      const bool st = i < n && (i == 0 || KEY_G(s_keys[i]) != KEY_G(s_keys[i - 1]));
      int tot;
      const int inc = blockInclScan(st ? 1 : 0, s_red, tot);
      if (st) s_gs[nG + inc - 1] = (unsigned)i;
      nG += tot;
    }
    if (lane == 0) s_gs[nG] = (unsigned)n;
    __syncthreads();
    // the keys of the minus strand sort first: their number, and that of their groups, by bisection
    int nMinusKeys, nMinusGroups;
    { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (KEY_PLUS(s_keys[mid])) hi = mid; else lo = mid + 1; } nMinusKeys = lo; }
    { int lo = 0, hi = nG; while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)s_gs[mid] >= nMinusKeys) hi = mid; else lo = mid + 1; } nMinusGroups = lo; }
    unsigned short *gz = wd.gSize + (size_t)pg * wd.pcap;
    for (int g = lane; g < nG; g += NT) gz[g] = (unsigned short)(s_gs[g + 1] - s_gs[g]);
    if (lane == 0) { int *gc = wd.gCount + (size_t)pg * 4; gc[0] = nMinusGroups; gc[1] = nG - nMinusGroups; gc[2] = nMinusKeys; gc[3] = 0; }
    if (pl.huge) {
      const uint2 *sd = wd.seed + (size_t)w * T4_WIDE_SEEDS;
      for (int q = lane; q < 2 * pl.nk; q += NT) s_cnt[q] = sd[q + 1].y - sd[q].y;
      __syncthreads();
      unsigned char *gi = wd.gInfo + (size_t)pg * wd.pcap;
      for (int g = lane; g < nG; g += NT) {
        int small = 0, minA = 0x7FFFFFFF, minSmall = 0;
        for (unsigned i = s_gs[g]; i < s_gs[g + 1]; ++i) {
          const unsigned long long kt = s_keys[i];
          const int a = KEY_C(kt) - T4_C_BIAS + KEY_B(kt);
          const int isSmall = s_cnt[KEY_PLUS(kt) ? a : pl.nk + a] <= 10000u;
          small += isSmall;
          if (a < minA) { minA = a; minSmall = isSmall; }
        }
        gi[g] = (unsigned char)((small > 4 ? 4 : small) | (minSmall ? 8 : 0));
      }
    }
  }
}

// ints of a read's statistics record
#define WS_NOVELMIN 0    // [2]
#define WS_ROR 2         // [2] removeOnlyRepeats
#define WS_STABLE 4
#define WS_M 5           // entries of uniqPref that mean something (0: not a huge read)
#define WS_GROUPS 6
#define WS_N4 7          // groups of four or more hits, both strands
#define WS_STATS8 8      // [8] per strand: groups of >= 4 hits, of >= 5 hits, the largest group, novelMinHitRequired (T4QueryArgs::stats8)

// SeqSet.hpp:784-823 over all groups of one read, in the reference's order: every minus-strand group by contig, then every
// plus-strand group (a partition holds a contig range of both).
__global__ __launch_bounds__(512) void wideStatsKernel(T4IndexView ix, T4Wide wd) {
  __shared__ int s_off[2 * T4_WIDE_MAXP + 1];
  __shared__ int s_red[16];
  __shared__ int s_acc[16];
  __shared__ unsigned s_cum[8192 + 1];
  __shared__ unsigned long long s_blk[4096];
  __shared__ unsigned s_cnt[T4_WIDE_SEEDS];
  if (wd.ctl[2]) return;
  const int nR = wideReads(wd), lane = threadIdx.x, NT = blockDim.x;
  for (int w = blockIdx.x; w < nR; w += gridDim.x) {
    T4WidePlan pl = wd.plan[w];
    const int P = pl.P;
    if (P <= 0) continue;
    __syncthreads();
    // segment s = strand * P + p, strand 0 = minus: exclusive prefix of the group counts
    int carry = 0;
    for (int s0 = 0; s0 < 2 * P; s0 += NT) {
      const int s = s0 + lane;
      const int v = s < 2 * P ? wd.gCount[(size_t)(pl.pBase + s % P) * 4 + s / P] : 0;
      int tot;
      const int inc = blockInclScan(v, s_red, tot);
      if (s < 2 * P) { s_off[s] = carry + inc - v; wd.gOff[(size_t)(pl.pBase + s % P) * 2 + s / P] = carry + inc - v; }
      carry += tot;
    }
    const int G = carry;
    if (lane == 0) {
      s_off[2 * P] = G;
      for (int t = 0; t < 16; ++t) s_acc[t] = 0;
      const int base = atomicAdd(&wd.ctl[3], G);
      if (base + G > wd.grpCap) atomicOr(&wd.ctl[2], 16);
      wd.plan[w].grpBase = base;
    }
    __syncthreads();
    // true size (and small-hit info) of group t of the read
    auto groupOf = [&](int t, int &pg, int &gi) {
      int lo = 0, hi = 2 * P - 1;   // last segment with off <= t (empty segments share an offset with their successor: take the last)
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= t) lo = mid; else hi = mid - 1; }
      pg = pl.pBase + lo % P;
      gi = t - s_off[lo] + (lo >= P ? wd.gCount[(size_t)pg * 4] : 0);
    };
    int rCarry = -1;
    for (int t0 = 0; t0 < G; t0 += NT) {
      const int t = t0 + lane;
      int u = -1, nT = 0, info = 0, plus = 0;
      if (t < G) {
        int pg, gi;
        groupOf(t, pg, gi);
        nT = wd.gSize[(size_t)pg * wd.pcap + gi];
        info = pl.huge ? wd.gInfo[(size_t)pg * wd.pcap + gi] : 0;
        plus = t >= s_off[P];
        if (t >= 1) {
          int pg1, gi1;
          groupOf(t - 1, pg1, gi1);
          const int nPrev = wd.gSize[(size_t)pg1 * wd.pcap + gi1];
          if (t - 1 == 0 || nPrev >= 2) u = t - 1;
        }
      }
      int tot;
      int r = blockInclMaxScan(u, s_red, tot);
      if (rCarry > r) r = rCarry;
      if (tot > rCarry) rCarry = tot;
      if (t < G) {
        const bool skip = t >= 1 && (((t - r - 1) & 1) == 0);
        const int m = nT - (skip ? 1 : 0);
        if (m > 0) {
          if (m > 3) atomicAdd(&s_acc[plus], 1);             // possibleOverlapCnt
          atomicMax(&s_acc[2 + plus], m);                    // longestHits
          // removeOnlyRepeats: a visited group with three hits of at most 10000 postings in its measured part
          const int small = pl.huge ? ((info & 7) - ((skip && (info & 8)) ? 1 : 0)) : m;
          if (small >= 3) s_acc[10 + plus] = 1;
        }
        if (nT >= 4) { atomicAdd(&s_acc[4 + plus], 1); if (nT >= 5) atomicAdd(&s_acc[6 + plus], 1); }
        atomicMax(&s_acc[8 + plus], nT);
      }
    }
    __syncthreads();
    int *st = wd.stat + (size_t)w * T4_WIDE_STAT;
    if (lane == 0) {
      int stable = pl.huge ? 0 : 1;
      for (int t = 0; t <= 1; ++t) {
        const int possible = s_acc[t], longest = s_acc[2 + t];
        int nm = 3;
        if (possible > 100000) nm = (int)(longest * 0.75);
        else if (possible > 10000) nm = longest / 2;
        else if (possible > 1000) nm = longest / 3;
        else if (possible > 100) nm = longest / 4;
        st[WS_NOVELMIN + t] = nm;
        st[WS_ROR + t] = s_acc[10 + t];
        // the same certificate as overlapsFromKeys gives: the threshold cannot move while every group of three or more hits stays
        const int lo = s_acc[6 + t], hi = s_acc[4 + t], big = s_acc[8 + t];
        const int cLo = lo > 100000 ? 4 : lo > 10000 ? 3 : lo > 1000 ? 2 : lo > 100 ? 1 : 0;
        const int cHi = hi > 100000 ? 4 : hi > 10000 ? 3 : hi > 1000 ? 2 : hi > 100 ? 1 : 0;
        bool ok = cLo == cHi;
        if (ok && cLo > 0) {
          const int a = big - 1 > 0 ? big - 1 : 0;
          const int fa = cLo == 4 ? (int)(a * 0.75) : cLo == 3 ? a / 2 : cLo == 2 ? a / 3 : a / 4;
          const int fb = cLo == 4 ? (int)(big * 0.75) : cLo == 3 ? big / 2 : cLo == 2 ? big / 3 : big / 4;
          ok = fa == fb && fa >= 3;
        }
        if (!ok) stable = 0;
      }
      st[WS_STABLE] = stable;
      st[WS_M] = 0;
      st[WS_GROUPS] = G;
      st[WS_N4] = s_acc[4] + s_acc[5];
      for (int t = 0; t <= 1; ++t) { st[WS_STATS8 + t] = s_acc[4 + t]; st[WS_STATS8 + 2 + t] = s_acc[6 + t]; st[WS_STATS8 + 4 + t] = s_acc[8 + t]; st[WS_STATS8 + 6 + t] = st[WS_NOVELMIN + t]; }
    }
    __syncthreads();
    if (!pl.huge || !(s_acc[10] | s_acc[11])) continue;
    // The run test of SeqSet.hpp:934-940 reads hits[k].repeats with k RELATIVE to its group (a quirk): the entries it reads are the
    // first e <= (largest group) entries of the whole hit array in the reference's order -- (strand, contig, read offset). Collect
    // whole groups from the head of that order until they cover the largest group, order every one by read offset, and leave the
    // prefix counts of entries with at most 10000 postings.
    const int M = s_acc[8] > s_acc[9] ? s_acc[8] : s_acc[9];
    {
      const uint2 *sd = wd.seed + (size_t)w * T4_WIDE_SEEDS;
      for (int q = lane; q < 2 * pl.nk; q += NT) s_cnt[q] = sd[q + 1].y - sd[q].y;
    }
    int T = 0, cum = 0;   // groups collected, their hits
    for (int t0 = 0; t0 < G; t0 += NT) {
      const int t = t0 + lane;
      int nT = 0;
      if (t < G) { int pg, gi; groupOf(t, pg, gi); nT = wd.gSize[(size_t)pg * wd.pcap + gi]; }
      int tot;
      const int inc = blockInclScan(nT, s_red, tot);
      if (t < G && t < 8192) s_cum[t] = (unsigned)(cum + inc - nT);
      if (lane == 0) s_red[8] = 0x7FFFFFFF;
      __syncthreads();
      if (t < G && cum + inc >= M) atomicMin(&s_red[8], lane);   // first group whose inclusive sum covers the largest group
      __syncthreads();
      const int f = s_red[8];
      if (lane == f) s_red[9] = cum + inc;
      __syncthreads();
      if (f != 0x7FFFFFFF) { T = t0 + f + 1; cum = s_red[9]; break; }
      T = G - t0 < NT ? G : t0 + NT;
      cum += tot;
    }
    __syncthreads();
    if (T > 8192 || cum > 2 * wd.pcap) { if (lane == 0) atomicOr(&wd.ctl[2], 32); continue; }   // (cannot happen: cum < M + pcap, T <= cum)
    unsigned long long *tmp = wd.sortTmp + (size_t)w * 2 * wd.pcap;
    for (int t = lane; t < T; t += NT) {
      int pg, gi;
      groupOf(t, pg, gi);
      const unsigned short *gz = wd.gSize + (size_t)pg * wd.pcap;
      unsigned at = 0;
      for (int g = 0; g < gi; ++g) at += gz[g];
      const unsigned long long *gk = wd.pKeys + (size_t)pg * wd.pcap + at;
      const int nT = gz[gi];
      for (int j = 0; j < nT; ++j) {
        const unsigned long long kt = gk[j];
        const int a = KEY_C(kt) - T4_C_BIAS + KEY_B(kt);
        const unsigned small = s_cnt[KEY_PLUS(kt) ? a : pl.nk + a] <= 10000u ? 1u : 0u;
        tmp[s_cum[t] + j] = ((unsigned long long)t << 32) | ((unsigned long long)a << 8) | small;
      }
    }
    __syncthreads();
    if (cum > 1) bitonicSortBlocked<unsigned long long>(tmp, cum, s_blk, 4096);
    __syncthreads();
    unsigned short *up = wd.uniqPref + (size_t)w * (wd.pcap + 1);
    int run = 0;
    const int lim = M < cum ? M : cum;
    for (int k0 = 0; k0 < lim; k0 += NT) {
      const int k = k0 + lane;
      const int v = k < lim ? (int)(tmp[k] & 1ull) : 0;
      int tot;
      const int inc = blockInclScan(v, s_red, tot);
      if (k < lim) up[k + 1] = (unsigned short)(run + inc);
      run += tot;
    }
    if (lane == 0) { up[0] = 0; st[WS_M] = lim; }
    __syncthreads();
  }
}

// a partition: dependency records of its groups, runs -> chains -> scored overlap records
template <int PCAP, int MAXOV>
__global__ __launch_bounds__(512) void wideChainKernel(T4IndexView ixArg, T4BatchView bvArg, T4Work wkArg, T4QueryArgs qaArg, T4Wide wd) {
  __shared__ unsigned long long s_keys[PCAP];
  __shared__ unsigned s_pairs[PCAP + PCAP / 3 + 2];
  __shared__ OvRec s_ov[MAXOV];
  __shared__ unsigned short s_ord[MAXOV];
  __shared__ char s_seg[T4_MAXL + 8];
  __shared__ char s_rc[T4_MAXL + 8];
  __shared__ WaveState s_ws;
  __shared__ T4IndexView s_ix;
  __shared__ T4BatchView s_bv;
  __shared__ T4Work s_wk;
  __shared__ T4QueryArgs s_qa;
  __shared__ WaveMem s_wm;
  if (wd.ctl[2]) return;
  if (threadIdx.x == 0) {
    s_ix = ixArg; s_bv = bvArg; s_wk = wkArg; s_qa = qaArg;
    WaveMem &m = s_wm;
    m.keys = s_keys; m.pairs = s_pairs; m.cand = s_pairs + PCAP; m.ov = s_ov; m.fin = s_ov; m.ord = s_ord;
    m.cap = PCAP; m.maxOv = MAXOV; m.maxFin = MAXOV; m.candCap = PCAP / 3 + 2; m.ldsArrays = 1; m.hitLimit = PCAP;
    m.ldsSort = nullptr; m.ldsSortCap = 0; m.dirBuf = (unsigned char *)s_keys; m.dirBytes = PCAP * 8;
    m.seg = s_seg; m.rc = s_rc;
  }
  __syncthreads();
  const T4IndexView &ix = s_ix;
  const T4BatchView &bv = s_bv;
  const T4Work &wk = s_wk;
  WaveMem &wm = s_wm;
  WaveState *ws = &s_ws;
  DPScratch sc;
  sc.rows = wk.dpRows + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (6 * T4_ROWW * 64);
  sc.dir = wk.dpDir + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * T4_DIR_BYTES;
  const int nPart = wideParts(wd), lane = threadIdx.x, NT = blockDim.x, K = ix.k;
  for (int pg = blockIdx.x; pg < nPart; pg += gridDim.x) {
    const int w = wd.pRead[pg];
    const T4WidePlan pl = wd.plan[w];
    const int n = (int)wd.pCnt[pg];
    __syncthreads();
    if (lane == 0) { ws->overflow = 0; ws->unsupported = 0; ws->finCount = 0; ws->ovCount = 0; ws->candCount = 0; ws->jobCount = 0; ws->statsStable = 1; ws->wideWant = 0; }
    if (n == 0) continue;   // (pRecCnt was zeroed when the partition was planned; an empty partition has no group)
    const long long r = pl.read;
    const int len = bv.len[r];
    loadSegment(bv, r, 0, len, wm);
    const unsigned long long *gk = wd.pKeys + (size_t)pg * wd.pcap;
    for (int i = lane; i < n; i += NT) s_keys[i] = gk[i];
    const int *st = wd.stat + (size_t)w * T4_WIDE_STAT;
    if (lane == 0) { ws->novelMin[0] = st[WS_NOVELMIN]; ws->novelMin[1] = st[WS_NOVELMIN + 1]; }
    const int ror0 = st[WS_ROR], ror1 = st[WS_ROR + 1], M = st[WS_M];
    __syncthreads();
    // (1) dependency records: per group its hits and the hull of its diagonals with three or more hits
    unsigned *gs = wm.pairs;
    int nG = 0;
    for (int i0 = 0; i0 < n; i0 += NT) {
      const int i = i0 + lane;
      const bool stt = i < n && (i == 0 || KEY_G(s_keys[i]) != KEY_G(s_keys[i - 1]));
      int tot;
      const int inc = blockInclScan(stt ? 1 : 0, ws->red, tot);
      if (stt) gs[nG + inc - 1] = (unsigned)i;
      nG += tot;
    }
    __syncthreads();
    {
      const int *gc = wd.gCount + (size_t)pg * 4;
      const int nMinus = gc[0];
      const int off0 = wd.gOff[(size_t)pg * 2], off1 = wd.gOff[(size_t)pg * 2 + 1];
      T4Grp *out = wd.grpPool + pl.grpBase;
      for (int g = lane; g < nG; g += NT) {
        const int s0 = (int)gs[g], e0 = g + 1 < nG ? (int)gs[g + 1] : n;
        int lo = 0x7FFFFFFF, hi = -0x7FFFFFFF;
        for (int i = s0; i < e0;) {
          const int c = KEY_C(s_keys[i]);
          int j = i + 1;
          while (j < e0 && KEY_C(s_keys[j]) == c) ++j;
          if (j - i >= 3) {
            const int at = -(c - T4_C_BIAS);   // start of the read on the contig along this diagonal
            if (at < lo) lo = at;
            if (at + len - 1 > hi) hi = at + len - 1;
          }
          i = j;
        }
        const unsigned long long k0 = s_keys[s0];
        T4Grp rec;
        rec.key = (unsigned)KEY_IDX(k0) * 2u + (unsigned)KEY_PLUS(k0); rec.cnt = (unsigned)(e0 - s0); rec.lo = lo; rec.hi = hi;
        // (a read with lists beyond 10000 postings: the group's hits of shorter lists, capped at 4, and whether its first hit is one --
        // what the statistics loop reads of it for removeOnlyRepeats, SeqSet.hpp:796-806 -- ride in bits 24-27 of the count: T4_GRP_INFO_SHIFT)
        if (pl.huge) rec.cnt |= (unsigned)wd.gInfo[(size_t)pg * wd.pcap + g] << 24;
        out[g < nMinus ? off0 + g : off1 + (g - nMinus)] = rec;
      }
    }
    __syncthreads();
    // (2) candidate runs (SeqSet.hpp:828-940 for novel contigs: a run is a stretch of one diagonal)
    unsigned *starts = wm.pairs;
    int nRuns = 0;
    for (int i0 = 0; i0 < n; i0 += NT) {
      const int i = i0 + lane;
      bool runStart = false;
      if (i < n) {
        runStart = true;
        if (i > 0) { const unsigned long long kp = s_keys[i - 1], ki = s_keys[i]; if (KEY_G(kp) == KEY_G(ki) && KEY_C(ki) == KEY_C(kp)) runStart = false; }
      }
      int tot;
      const int inc = blockInclScan(runStart ? 1 : 0, ws->red, tot);
      if (runStart) starts[nRuns + inc - 1] = (unsigned)i;
      nRuns += tot;
    }
    __syncthreads();
    const uint2 *sd = wd.seed + (size_t)w * T4_WIDE_SEEDS;
    const unsigned short *up = wd.uniqPref + (size_t)w * (wd.pcap + 1);
    for (int rr = lane; rr < nRuns; rr += NT) {
      const int i = (int)starts[rr], e = rr + 1 < nRuns ? (int)starts[rr + 1] : n;
      const int nn = e - i;
      const unsigned long long ki = s_keys[i];
      const int plus = KEY_PLUS(ki);
      const int minHit = ws->novelMin[plus];
      if (!(nn >= minHit && nn * K >= ix.hitLenRequired)) continue;
      if (pl.huge && (plus ? ror1 : ror0)) {
        // the group must hold a hit of at most 10000 postings (SeqSet.hpp:876), and so must the entries [s, e) of the read's hit
        // array that the run-relative test reads (934-940)
        int g0 = i;
        while (g0 > 0 && KEY_G(s_keys[g0 - 1]) == KEY_G(ki)) --g0;
        int g1 = e;
        while (g1 < n && KEY_G(s_keys[g1]) == KEY_G(ki)) ++g1;
        bool uniq = false;
        for (int j = g0; j < g1 && !uniq; ++j) {
          const unsigned long long kt = s_keys[j];
          const int a = KEY_C(kt) - T4_C_BIAS + KEY_B(kt);
          const int q = KEY_PLUS(kt) ? a : pl.nk + a;
          if (sd[q + 1].y - sd[q].y <= 10000u) uniq = true;
        }
        if (!uniq) continue;
        const int rs = i - g0, re = e - g0;
        if (re > M) { ws->unsupported = 1; continue; }   // (cannot happen: M covers the largest group)
        if (up[re] == up[rs]) continue;
      }
      const int slot = atomicAdd(&ws->candCount, 1);
      if (nn >= (1 << (32 - CAND_START_BITS))) ws->unsupported = 1;
      else if (slot < wm.candCap) wm.cand[slot] = (unsigned)i | ((unsigned)nn << CAND_START_BITS);
      else ws->overflow = 1;
    }
    __syncthreads();
    const int nCand = ws->candCount < wm.candCap ? ws->candCount : wm.candCap;
    chainRunsRows(ix, wm, ws, nCand, ix.hitLenRequired);
    __syncthreads();
    int overlapCnt = ws->ovCount;
    if (ws->overflow || overlapCnt > MAXOV) { if (lane == 0) atomicOr(&wd.ctl[2], 8); continue; }
    for (int i = lane; i < overlapCnt; i += NT) s_ord[i] = (unsigned short)i;
    __syncthreads();
    // (3) gap alignments of every overlap of both strands (the strand of the read's best overlap is known to the merge only)
    if (!scoreOverlaps<true>(ix, wm, ws, overlapCnt, sc)) { if (lane == 0) atomicOr(&wd.ctl[2], 8); continue; }
    __syncthreads();
    OvRec *dst = (OvRec *)wd.pRec + (size_t)pg * wd.maxOvPart;
    for (int i = lane; i < overlapCnt; i += NT) dst[i] = s_ov[i];
    if (lane == 0) {
      wd.pRecCnt[pg] = overlapCnt;
      if (ws->unsupported) wk.status[r] = 1;
    }
  }
}

__device__ __forceinline__ OvRec *wideRec(const T4Wide &wd, const T4WidePlan &pl, int idx) {
  return (OvRec *)wd.pRec + (size_t)(pl.pBase + (idx >> T4_WIDE_OVBITS)) * wd.maxOvPart + (idx & ((1 << T4_WIDE_OVBITS) - 1));
}

// a read: the steps of GetOverlapsFromRead that look across contigs, over the records of all its partitions
__global__ __launch_bounds__(512) void wideMergeKernel(T4IndexView ix, T4BatchView bv, T4Work wk, T4QueryArgs qa, T4Wide wd) {
  __shared__ unsigned long long s_keys[8192];
  __shared__ int s_off[T4_WIDE_MAXP + 1];
  __shared__ int s_red[16];
  __shared__ int s_best;
  if (wd.ctl[2]) return;
  const int nR = wideReads(wd), lane = threadIdx.x, NT = blockDim.x;
  for (int w = blockIdx.x; w < nR; w += gridDim.x) {
    const T4WidePlan pl = wd.plan[w];
    const int P = pl.P;
    if (P <= 0) continue;
    const long long r = pl.read;
    const int len = bv.len[r];
    __syncthreads();
    int carry = 0;
    for (int p0 = 0; p0 < P; p0 += NT) {
      const int p = p0 + lane;
      const int v = p < P ? wd.pRecCnt[pl.pBase + p] : 0;
      int tot;
      const int inc = blockInclScan(v, s_red, tot);
      if (p < P) s_off[p] = carry + inc - v;
      carry += tot;
    }
    const int N = carry;
    if (lane == 0) s_off[P] = N;
    __syncthreads();
    const int *st = wd.stat + (size_t)w * T4_WIDE_STAT;
    if (lane == 0 && qa.statsStable) qa.statsStable[r] = st[WS_STABLE];
    if (lane == 0 && qa.n4) qa.n4[r] = st[WS_N4];
    const T4CandArgs *cs = qa.cs;
    if (cs && lane < 8) cs->stats8[T4_QSTATS * r + lane] = st[WS_STATS8 + lane];
    if (lane == 0 && cs && cs->candCnt) cs->candCnt[r] = 0;
    if (N == 0) { if (lane == 0) { qa.counts[r] = 0; qa.outBase[r] = 0; if (qa.aux) qa.aux[r] = 0; } continue; }
    // std::sort(overlaps) (SeqSet.hpp:1597) on the records as GetOverlapsFromHits left them: matchCnt (kept in chainLen), read span,
    // contig, strand in one key with the record's index; ties on all four are settled by the rest of operator<
    const bool inLds = N <= 8192;
    unsigned long long *mk = inLds ? s_keys : wd.mKeys + (size_t)pl.pBase * wd.maxOvPart;
    int *ord = wd.mOrd + (size_t)pl.pBase * wd.maxOvPart;
    for (int j = lane; j < N; j += NT) {
      int lo = 0, hi = P - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= j) lo = mid; else hi = mid - 1; }
      const int idx = (lo << T4_WIDE_OVBITS) | (j - s_off[lo]);
      const OvRec o = *wideRec(wd, pl, idx);
      const int span = o.re - o.rs, m0 = o.chainLen;
      mk[j] = ((unsigned long long)(2047 - (m0 & 2047)) << 53) | ((unsigned long long)(1023 - (span & 1023)) << 43) |
              ((unsigned long long)(unsigned)o.seqIdx << 21) | ((unsigned long long)((o.flags & OV_PLUS) ? 1 : 0) << 20) | (unsigned long long)idx;
    }
    __syncthreads();
    if (N > 1) { if (inLds) bitonicSortRegLds<unsigned long long>(s_keys, N); else bitonicSortBlocked<unsigned long long>(mk, N, s_keys, 8192); }
    __syncthreads();
    for (int p = lane; p < N; p += NT) {
      const unsigned long long key = mk[p], pre = key >> 20;
      const int idx = (int)(key & 0xFFFFFull);
      int gsx = p, ge = p + 1;
      while (gsx > 0 && (mk[gsx - 1] >> 20) == pre) --gsx;
      while (ge < N && (mk[ge] >> 20) == pre) ++ge;
      int rank = 0;
      if (ge - gsx > 1) {
        const OvRec me = *wideRec(wd, pl, idx);
        for (int q = gsx; q < ge; ++q) {
          const int j = (int)(mk[q] & 0xFFFFFull);
          if (j == idx) continue;
          const OvRec ot = *wideRec(wd, pl, j);
          int cm = 0;   // the rest of operator<: readStart, readEnd, seqStart, seqEnd
          if (ot.rs != me.rs) cm = ot.rs < me.rs ? -1 : 1;
          else if (ot.re != me.re) cm = ot.re < me.re ? -1 : 1;
          else if (ot.ss != me.ss) cm = ot.ss < me.ss ? -1 : 1;
          else if (ot.se != me.se) cm = ot.se < me.se ? -1 : 1;
          if (cm < 0 || (cm == 0 && j < idx)) ++rank;
        }
      }
      ord[gsx + rank] = idx;
    }
    __syncthreads();
    // the strand of the best overlap (SeqSet.hpp:1601-1616), order preserved
    const int strand0 = wideRec(wd, pl, ord[0])->flags & OV_PLUS;
    int kept = 0;
    for (int i0 = 0; i0 < N; i0 += NT) {
      const int i = i0 + lane;
      const int o = i < N ? ord[i] : 0;
      const bool keep = i < N && ((wideRec(wd, pl, o)->flags & OV_PLUS) == strand0);
      int tot;
      const int inc = blockInclScan(keep ? 1 : 0, s_red, tot);
      if (keep) ord[kept + inc - 1] = o;
      kept += tot;
      __syncthreads();
    }
    const int cnt = kept;
    if (lane == 0 && qa.aux) qa.aux[r] = (cnt > 32767 ? 32767 : cnt) | ((N - cnt > 32767 ? 32767 : N - cnt) << 15) | ((strand0 ? 1 : 0) << 30);   // see T4QueryArgs::aux
    // the pre-filters against the best novel overlap so far (SeqSet.hpp:1705-1794), replayed as prefilterNovel does
    if (cnt > 50) {
      int best = -1;
      OvRec bn = *wideRec(wd, pl, ord[0]);
      for (int i0 = 0; i0 < cnt; i0 += NT) {
        const int i = i0 + lane;
        const bool has = i < cnt;
        OvRec *op = wideRec(wd, pl, has ? ord[i] : ord[0]);
        OvRec o = *op;
        int from = i0;
        for (;;) {
          bool cut = false, cand = false;
          if (has && i >= from) {
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
              if (!cut && cnt > 1000 && m0 < 0.9 * bn.matchCnt) cut = true;
            }
            if (!cut && !(o.flags & OV_SIMZERO) && ovSim(o) > 0 && (best == -1 || ovLess(o, bn, true))) cand = true;
          }
          if (lane == 0) s_best = 0x7FFFFFFF;
          __syncthreads();
          if (cand) atomicMin(&s_best, i);
          __syncthreads();
          const int f = s_best;
          if (cut && i < f) {
            ovCutKeepScored(o);
            *op = o;
            from = 0x7FFFFFFF;
          }
          __syncthreads();
          if (f == 0x7FFFFFFF) break;
          best = f;
          bn = *wideRec(wd, pl, ord[f]);
          from = from == 0x7FFFFFFF ? from : f + 1;
        }
      }
      __syncthreads();
    }
    if (cs && cs->candOut && cnt > 0) {   // the candidate store (T4QueryArgs::cs): every overlap on the strand of the best one, in scan order
      if (lane == 0) {
        const unsigned cb = atomicAdd(cs->candCursor, (unsigned)cnt);
        s_red[13] = (cb + (unsigned)cnt > (unsigned)cs->candCap) ? -1 : (int)cb;
        if (s_red[13] < 0) atomicOr(cs->candOverflow, 1);
      }
      __syncthreads();
      const int cb = s_red[13];
      if (cb >= 0) {
        T4Cand *out = cs->candOut;
        for (int i = lane; i < cnt; i += NT) out[cb + i] = ovToCand(*wideRec(wd, pl, ord[i]));
        if (lane == 0) { cs->candBase[r] = cb; cs->candCnt[r] = cnt; }
      }
      __syncthreads();
    }
    // the similarity cut (SeqSet.hpp:2105-2119): count, take room in the result pool, write
    int outCnt = 0;
    for (int i0 = 0; i0 < cnt; i0 += NT) {
      const int i = i0 + lane;
      bool keep = false;
      if (i < cnt) keep = !(ovSim(*wideRec(wd, pl, ord[i])) < ix.novelSim);
      outCnt += blockSum(keep ? 1 : 0, s_red);
    }
    if (lane == 0) {
      const int base = outCnt > 0 ? (int)atomicAdd(qa.poolCursor, (unsigned)outCnt) : 0;
      s_red[15] = (base + outCnt > qa.poolCap) ? -1 : base;
      s_red[14] = base;
    }
    __syncthreads();
    const int base = s_red[15], reserved = s_red[14];
    __syncthreads();
    if (base < 0) {   // the pool is full: the host grows it and repeats the call (see processRead)
      if (qa.recRead) for (int i = reserved + lane; i < qa.poolCap && i < reserved + outCnt; i += NT) if (i >= 0) qa.recRead[i] = -1;
      if (lane == 0) { wk.status[r] = 3; qa.counts[r] = 0; }
      continue;
    }
    int done = 0;
    for (int i0 = 0; i0 < cnt; i0 += NT) {
      const int i = i0 + lane;
      bool keep = false;
      OvRec o;
      if (i < cnt) { o = *wideRec(wd, pl, ord[i]); keep = !(ovSim(o) < ix.novelSim); }
      int tot;
      const int inc = blockInclScan(keep ? 1 : 0, s_red, tot);
      if (keep) {
        const int at = base + done + inc - 1;
        o.chainLen = 0;
        storeOverlap(qa.out + at, o);
        storeOverlap(qa.outDev + at, o);
        qa.recRead[at] = (int)r;
      }
      done += tot;
    }
    if (lane == 0) { qa.outBase[r] = base; qa.counts[r] = outCnt; if (qa.readTicks) qa.readTicks[r] = 0; }
  }
}

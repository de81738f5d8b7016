// trust4_amd/host/process_read.h -- the host side of the input phase that has arithmetic of its own: AlignAlgo::IsMateOverlap
// (AlignAlgo.hpp:1027-1096), IsLowComplexity (main.cpp:183-205) and ProcessRead (main.cpp:224-449) as trust4-hip's host threads run
// them, the read record they work on, and the numbering of barcode / UMI strings (main.cpp:812-842). A header of its own so that tests/host_probe.cpp can put exactly this code next to the
// oracle's restatement (tests/test_host_logic.py); trust4_main.cpp is its only product user.
#pragma once
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <string.h>

#include <algorithm>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/trust4_hip.h"

namespace t4host {

inline int nucNum(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
const char NUM2NUC[4] = {'A', 'C', 'G', 'T'};

inline void revCompInPlace(std::string &s) {
  std::reverse(s.begin(), s.end());
  for (char &c : s) if (c != 'N') { int n = nucNum(c); c = n >= 0 ? NUM2NUC[3 - n] : 'N'; }
}

// ---- AlignAlgo::IsMateOverlap (AlignAlgo.hpp:1027-1096) ---------------------------------------------
inline int isMateOverlap(const std::string &fr, const std::string &sr, int minOverlap, int &offset, int &bestMatchCnt, bool checkTandem) {
  const int flen = (int)fr.size(), slen = (int)sr.size();
  int offsetCnt = 0, overlapSize = -1;
  bestMatchCnt = -1;
  for (int j = 0; j < flen - minOverlap; ++j) {
    int matchCnt = 0, k;
    bool ok = true;
    double thr = 0.95;
    if (flen - j >= 100) thr = 0.85;
    else if (flen - j >= 50) thr = 0.85 + (flen - j - 50) / 50.0 * 0.1;
    const int need = int((flen - j) * thr);
    // The reference's test after every base, `matchCnt + (flen - (j + k) - 1) < need`, says "more than (flen - j) - need mismatches so
    // far"; mismatches only grow, so an offset fails iff the mismatches of its whole range exceed that allowance -- counted 16 bases at
    // a time (SSE2 is part of x86-64), leaving as soon as the allowance is spent. A surviving offset has k = the range, matchCnt = its matches.
    const int range = flen - j < slen ? flen - j : slen, allowed = (flen - j) - need;
    int mism = 0;
    const char *a = fr.data() + j, *b = sr.data();
    k = 0;
#if defined(__SSE2__)
    for (; k + 16 <= range; k += 16) {
      const __m128i va = _mm_loadu_si128((const __m128i *)(a + k)), vb = _mm_loadu_si128((const __m128i *)(b + k));
      mism += 16 - __builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(va, vb)));
      if (mism > allowed) { ok = false; break; }
    }
#else   // (a host without SSE2: eight bases at a time)
    for (; k + 8 <= range; k += 8) {
      for (int t = 0; t < 8; ++t) mism += a[k + t] != b[k + t];
      if (mism > allowed) { ok = false; break; }
    }
#endif
    if (ok) {
      for (; k < range; ++k) if (a[k] != b[k]) ++mism;
      if (mism > allowed) ok = false;
      matchCnt = range - mism; k = range;
    }
    if (ok) { offset = j; ++offsetCnt; overlapSize = k; bestMatchCnt = matchCnt; }
  }
  if (offsetCnt != 1) return -1;
  if (checkTandem && overlapSize <= minOverlap * 2) {
    for (int i = 1; i <= overlapSize / 2; ++i) {
      bool tandem = true;
      for (int j = i; j + i - 1 < overlapSize; j += i) {
        int k;
        for (k = j; k <= j + i - 1; ++k) if (sr[k - j] != sr[k]) break;
        if (k <= j + i - 1) { tandem = false; break; }
      }
      if (tandem) return -1;
    }
  }
  return overlapSize;
}

struct SortRead {
  std::string id, read, qual;
  bool hasQual = false, dead = false;
  int minCnt = 0, medianCnt = 0;
  float avgCnt = 0;
  int len = 0, strand = 0, mateIdx = -1, info = -1;
  int barcode = -1, umi = -1, barcodeMinCnt = 0, barcodeMedianCnt = 0;
  float barcodeAvgCnt = 0;
  t4_overlap g[4];
  bool operator<(const SortRead &b) const {   // main.cpp:103-125
    if (minCnt != b.minCnt) return minCnt > b.minCnt;
    if (medianCnt != b.medianCnt) return medianCnt > b.medianCnt;
    if (avgCnt != b.avgCnt) return avgCnt > b.avgCnt;
    if (len != b.len) return len > b.len;
    int t = strcmp(read.c_str(), b.read.c_str());
    if (t != 0) return t < 0;
    return strcmp(id.c_str(), b.id.c_str()) < 0;
  }
};

inline bool isLowComplexity(const std::string &s) {   // main.cpp:183-205
  int cnt[5] = {0, 0, 0, 0, 0};
  const int n = (int)s.size();
  for (char ch : s) { if (ch == 'N') ++cnt[4]; else ++cnt[nucNum(ch) < 0 ? 0 : nucNum(ch)]; }
  if (cnt[0] >= n / 2 || cnt[1] >= n / 2 || cnt[2] >= n / 2 || cnt[3] >= n / 2 || cnt[4] >= n / 10) return true;
  int low = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}

// ProcessRead (main.cpp:224-449): read-through clipping, mate merging, low-complexity filter; the 21-mer counting of the
// surviving reads (same multiset of AddCount calls) is done afterwards over the whole read list
// `pre` (optional): the two IsMateOverlap tests of this pair as t4_mate_overlap computed them for the whole block --
// {ret, offset, bestMatchCnt} of (rc(mate 2), mate 1, minOverlap, no tandem check) and of (mate 1, rc(mate 2), minOverlap2, tandem check)
inline void processRead(const SortRead &in1, const SortRead &in2, bool hasMate2, std::vector<SortRead> &out, const int32_t *pre = nullptr) {
  // Private copies: every string the read list ends up holding was allocated by the thread that ran this function, so the passes that
  // later shrink or free them on all threads (quality strings after the trimming, ...) meet as many allocator arenas as threads.
  // Measured (round 4, 1 M barcoded pairs, -t 32, profiles/r04u_*): taking the reader threads' strings over instead of copying them
  // sends every later free to the readers' two arenas -- 49 s of system time instead of 6, the run 13.6 s instead of 10.5.
  SortRead r1 = in1, r2 = in2;
  int rWeight = 1;
  bool r2Alive = hasMate2;
  if (hasMate2) {
    const int flen = (int)r2.read.size(), slen = (int)r1.read.size();
    revCompInPlace(r2.read);
    if (r2.hasQual) std::reverse(r2.qual.begin(), r2.qual.end());
    int minOverlap = (flen + slen) / 10, minOverlap2 = (flen + slen) / 20;
    if (minOverlap > 31) minOverlap = 31;
    if (minOverlap2 > 31) minOverlap2 = 31;
    int offset = -1, best = -1;
    int ov;
    if (pre) { ov = pre[0]; offset = pre[1]; best = pre[2]; } else ov = isMateOverlap(r2.read, r1.read, minOverlap, offset, best, false);
    if (ov >= 0) {   // read-through: keep the overlapped part of read 1
      r1.read.resize(ov);
      if (r1.hasQual) {
        r1.qual.resize(ov);
        for (int j = 0; j < ov; ++j)
          if (r2.qual[j + offset] > r1.qual[j] || r1.read[j] == 'N') { r1.read[j] = r2.read[j + offset]; r1.qual[j] = r2.qual[j + offset]; }
      }
      r2Alive = false;
    } else if ((ov = pre ? (offset = pre[4], best = pre[5], (int)pre[3]) : isMateOverlap(r1.read, r2.read, minOverlap2, offset, best, true)) >= 0) {
      if (best >= 0.95 * ov) {   // merge the mates
        std::string r(slen + flen + 1, '\0'), q(slen + flen + 1, '\0');
        for (int j = 0; j < flen; ++j) { r[offset + j] = r2.read[j]; q[offset + j] = r2.hasQual ? r2.qual[j] : 0; }
        const int len = offset + flen;
        for (int j = 0; j < slen && j < len; ++j)
          if (j < offset || (r1.hasQual ? r1.qual[j] : 0) >= q[j] - 14 || r[j] == 'N') { r[j] = r1.read[j]; q[j] = r1.hasQual ? r1.qual[j] : 0; }
        r.resize(len); q.resize(len);
        r1.read = r; r1.qual = q;
        r2Alive = false;
        ++rWeight;
      } else {
        bool useFirst = true;
        if (r1.hasQual) {
          double a = 0, b = 0;
          for (int j = offset; j < slen; ++j) a += r1.qual[j] - 32;
          for (int j = flen - 1; j >= flen - ov; --j) b += r2.qual[j] - 32;
          a /= ov; b /= ov;
          if (a + 10 < b) useFirst = false;
        }
        if (!useFirst) { r1.read = r2.read; revCompInPlace(r1.read); r1.qual = r2.qual; r1.hasQual = r2.hasQual; }
        r2Alive = false;
      }
    } else {
      revCompInPlace(r2.read);
      if (r2.hasQual) std::reverse(r2.qual.begin(), r2.qual.end());
    }
  }
  if (!isLowComplexity(r1.read)) {
    if (rWeight == 2) { SortRead w = r1; w.id += ".1"; out.push_back(std::move(r1)); out.push_back(std::move(w)); }
    else out.push_back(std::move(r1));
  }
  if (r2Alive && !isLowComplexity(r2.read)) out.push_back(std::move(r2));
}

// Strings -> dense ints in order of first appearance: the numbering of barcodes and UMIs (main.cpp:812-820, 831-842; the reference
// keeps a std::map<std::string, int> each). It is the serial work per record of the input loop, so strings of at most 20 letters
// over ACGTN -- every barcode and UMI of the 10x / Drop-seq kind -- are packed into 61 bits and numbered in a flat open-addressing
// table; anything else goes through a std::unordered_map. Both share one counter.
struct StrNumbering {
  std::vector<uint64_t> keys;
  std::vector<int> vals;
  size_t used = 0;
  std::unordered_map<std::string, int> other;
  int count = 0;
  static bool pack(const std::string &s, uint64_t &k) {
    if (s.size() > 20) return false;
    k = 1;   // (the leading 1 tells lengths apart; 0 is the empty slot)
    for (char c : s) {
      uint64_t v;
      switch (c) { case 'A': v = 1; break; case 'C': v = 2; break; case 'G': v = 3; break; case 'T': v = 4; break; case 'N': v = 5; break; default: return false; }
      k = (k << 3) | v;
    }
    return true;
  }
  static uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
  void grow() {
    std::vector<uint64_t> ok; std::vector<int> ov;
    ok.swap(keys); ov.swap(vals);
    keys.assign(ok.empty() ? (size_t)1 << 16 : ok.size() * 2, 0); vals.assign(keys.size(), 0);
    const size_t mask = keys.size() - 1;
    for (size_t i = 0; i < ok.size(); ++i) if (ok[i]) { size_t h = (size_t)mix(ok[i]) & mask; while (keys[h]) h = (h + 1) & mask; keys[h] = ok[i]; vals[h] = ov[i]; }
  }
  int number(const std::string &s, bool &isNew) {
    uint64_t k;
    isNew = false;
    if (!pack(s, k)) {
      auto it = other.find(s);
      if (it != other.end()) return it->second;
      isNew = true;
      other.emplace(s, count);
      return count++;
    }
    if ((used + 1) * 2 > keys.size()) grow();
    const size_t mask = keys.size() - 1;
    size_t h = (size_t)mix(k) & mask;
    while (keys[h]) { if (keys[h] == k) return vals[h]; h = (h + 1) & mask; }
    keys[h] = k; vals[h] = count; ++used; isNew = true;
    return count++;
  }
};

}  // namespace t4host

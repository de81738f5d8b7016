// tests/host_probe.cpp -- test infrastructure: the driver's host-side ProcessRead / IsMateOverlap / IsLowComplexity
// (trust4_amd/host/process_read.h, exactly the code trust4-hip's host threads run) next to the oracle's restatement of the
// reference (oracle/t4_oracle.c: t4o_process_read, t4o_is_mate_overlap, themselves pinned against the reference's own functions) on
// seeded pairs: unrelated mates, mates that overlap (read-through, merge, one-mate-for-both), tandem repeats, Ns, low-complexity
// reads, with and without qualities. Usage: host_probe PAIRS SEED -> "ok ..." or the first difference (exit 1).
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../trust4_amd/host/process_read.h"
#include "../trust4_amd/host/seq_reader.h"
extern "C" {
#include "../oracle/t4_oracle.h"
}

using namespace t4host;

static std::string revComp(const std::string &s) { std::string r = s; revCompInPlace(r); return r; }

// `host_probe reader A B ...`: every file through SeqReader (plain files by the raw path, .gz by zlib) and through ThreadedSeqReader in
// blocks, every consumed block handed back to the reader (the recycling of the driver's input loop); all must give the same records.
static int readerMode(int argc, char **argv) {
  std::vector<std::string> want;
  for (int f = 2; f < argc; ++f) {
    std::vector<std::string> got;
    {
      SeqReader rd;
      rd.files.push_back(argv[f]);
      while (rd.next()) got.push_back(rd.id + "|" + rd.comment + "|" + rd.seq + "|" + (rd.hasQual ? rd.qual : std::string("-")));
    }
    std::vector<std::string> blocks;
    {
      ThreadedSeqReader tr;
      tr.files.push_back(argv[f]);
      ThreadedSeqReader::Block b, held;
      while (tr.nextBlock(b)) {
        for (auto &r : b) blocks.push_back(r.id + "|" + r.comment + "|" + r.seq + "|" + (r.hasQual ? r.qual : std::string("-")));
        for (auto &r : b) { r.seq.swap(r.id); r.qual += "stale"; }   // what a consumer leaves behind must not show in later records
        tr.recycle(held);   // one block is held back for a round, as the driver's units are
        held.swap(b);
      }
    }
    if (got != blocks) { printf("reader: blocks of %s differ from its records (%zu vs %zu)\n", argv[f], blocks.size(), got.size()); return 1; }
    if (f == 2) want = got;
    else if (got != want) { printf("reader: %s gives other records than %s (%zu vs %zu)\n", argv[f], argv[2], got.size(), want.size()); return 1; }
  }
  unsigned long long h = 1469598103934665603ull;
  for (const std::string &r : want) for (char c : r) { h ^= (unsigned char)c; h *= 1099511628211ull; }
  printf("ok reader: %zu records, fnv %016llx\n", want.size(), h);
  return 0;
}

// `host_probe numbering N SEED`: StrNumbering (packed 61-bit keys in a flat table, anything else through a map, one counter) against
// a std::map numbered in order of first appearance, as the reference numbers barcodes and UMIs (main.cpp:812-820, 831-842).
#include <map>
static int numberingMode(long n, unsigned seed) {
  std::mt19937 rng(seed);
  StrNumbering sn;
  std::map<std::string, int> ref;
  long packed = 0, other = 0, fresh = 0;
  for (long i = 0; i < n; ++i) {
    std::string s;
    const int kind = (int)(rng() % 10);
    const int len = kind == 0 ? 21 + (int)(rng() % 12) : kind == 1 ? 0 : 1 + (int)(rng() % 20);     // longer than 20 letters; empty; packable lengths
    const int alphabet = (int)(rng() % 3) == 0 ? 2 : 4;                                              // (small alphabets: many repeats of earlier strings)
    for (int j = 0; j < len; ++j) s += "ACGTN"[rng() % (size_t)alphabet];
    if (kind == 2) s[rng() % s.size()] = "acgtX-1"[rng() % 7];                                       // a letter outside ACGTN
    if (kind == 3) s = "missing_barcode";
    uint64_t k;
    (StrNumbering::pack(s, k) ? packed : other)++;
    bool isNew = false;
    const int got = sn.number(s, isNew);
    auto it = ref.find(s);
    const bool wantNew = it == ref.end();
    const int want = wantNew ? (int)ref.size() : it->second;
    if (wantNew) { ref[s] = want; ++fresh; }
    if (got != want || isNew != wantNew) { printf("numbering differs at %ld: '%s' -> %d (new %d), expected %d (new %d)\n", i, s.c_str(), got, (int)isNew, want, (int)wantNew); return 1; }
  }
  printf("ok numbering: %ld strings (%ld packed, %ld through the map), %ld distinct\n", n, packed, other, fresh);
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 2 && !strcmp(argv[1], "reader")) return readerMode(argc, argv);
  if (argc > 3 && !strcmp(argv[1], "numbering")) return numberingMode(atol(argv[2]), (unsigned)atol(argv[3]));
  const long pairs = argc > 1 ? atol(argv[1]) : 20000;
  std::mt19937 rng(argc > 2 ? (unsigned)atol(argv[2]) : 1u);
  const char *ac = "ACGT";
  long kinds[4] = {0, 0, 0, 0}, listed[4] = {0, 0, 0, 0}, overlapCalls = 0, overlapHits = 0;
  std::vector<char> outR, outQ;
  for (long it = 0; it < pairs; ++it) {
    // a fragment and its two mates (mate 2 is the reverse complement of the fragment's end, as a sequencer gives it)
    const int l1 = 40 + (int)(rng() % 111), l2 = 40 + (int)(rng() % 111);
    const int mode = (int)(rng() % 8);
    int frag = mode < 3 ? l1 + l2 + (int)(rng() % 200)                                   // no overlap
             : mode < 5 ? std::max(l1, l2) - (int)(rng() % 30)                           // read-through (fragment shorter than a read)
                        : std::max(l1, l2) + (int)(rng() % std::max(1, std::min(l1, l2)));   // the mates overlap
    if (frag < 25) frag = 25;
    std::string f((size_t)frag, 'A');
    if (mode == 7) { const int p = 1 + (int)(rng() % 5); std::string unit; for (int i = 0; i < p; ++i) unit += ac[rng() % 4]; for (int i = 0; i < frag; ++i) f[(size_t)i] = unit[(size_t)(i % p)]; }
    else for (char &c : f) c = ac[rng() % 4];
    if (rng() % 40 == 0) for (int i = 0; i < frag; ++i) f[(size_t)i] = ac[(rng() % 10) ? 0 : rng() % 4];   // low complexity
    std::string r1 = f.substr(0, (size_t)std::min(l1, frag)), r2 = revComp(f.substr((size_t)std::max(0, frag - l2)));
    const int errs = (int)(rng() % 4) == 0 ? (int)(rng() % 12) : (int)(rng() % 2);
    for (int e = 0; e < errs; ++e) { std::string &r = (rng() & 1) ? r1 : r2; r[rng() % r.size()] = (rng() % 6) ? ac[rng() % 4] : 'N'; }
    const bool withQ = (rng() % 3) != 0;
    std::string q1, q2;
    if (withQ) { q1.resize(r1.size()); q2.resize(r2.size()); for (char &c : q1) c = (char)(35 + rng() % 40); for (char &c : q2) c = (char)(35 + rng() % 40); }
    // ---- the two IsMateOverlap tests by themselves
    for (int t = 0; t < 2; ++t) {
      const std::string a = t ? r1 : revComp(r2), b = t ? revComp(r2) : r1;
      int mo = (int)((r1.size() + r2.size()) / (t ? 20 : 10)); if (mo > 31) mo = 31;
      int o1 = -9, b1 = -9, o2 = -9, b2 = -9;
      const int v1 = isMateOverlap(a, b, mo, o1, b1, t != 0);
      const int v2 = t4o_is_mate_overlap(a.c_str(), (int)a.size(), b.c_str(), (int)b.size(), mo, &o2, &b2, t);
      ++overlapCalls; if (v1 >= 0) ++overlapHits;
      if (v1 != v2 || (v1 >= 0 && (o1 != o2 || b1 != b2))) { printf("IsMateOverlap differs at pair %ld test %d: %d/%d/%d vs %d/%d/%d\n", it, t, v1, o1, b1, v2, o2, b2); return 1; }
    }
    // ---- ProcessRead
    SortRead a, b;
    a.id = b.id = "r" + std::to_string(it); a.read = r1; b.read = r2; a.qual = q1; b.qual = q2; a.hasQual = b.hasQual = withQ;
    std::vector<SortRead> out;
    processRead(a, b, true, out);
    outR.assign(r1.size() + r2.size() + 2, 0); outQ.assign(r1.size() + r2.size() + 2, 0);
    int flags = 0;
    const int kind = t4o_process_read(r1.c_str(), withQ ? q1.c_str() : nullptr, r2.c_str(), withQ ? q2.c_str() : nullptr, outR.data(), outQ.data(), &flags);
    ++kinds[kind & 3];
    size_t want = 0;
    if (flags & 1) want += (flags & 4) ? 2 : 1;
    if (flags & 2) ++want;
    bool ok = out.size() == want;
    size_t at = 0;
    if (ok && (flags & 1)) {
      const std::string oR(outR.data());
      ok = out[at].read == oR && out[at].hasQual == ((flags & 8) != 0) && out[at].id == a.id;
      if (ok && (flags & 8)) ok = out[at].qual.size() == oR.size() && memcmp(out[at].qual.data(), outQ.data(), oR.size()) == 0;
      ++at;
      if (ok && (flags & 4)) { ok = out[at].read == oR && out[at].id == a.id + ".1" && out[at].hasQual == ((flags & 8) != 0); ++at; }
      ++listed[kind & 3];
    }
    if (ok && (flags & 2)) ok = out[at].read == r2 && out[at].id == b.id && out[at].qual == q2;
    if (!ok) { printf("ProcessRead differs at pair %ld (kind %d, flags %d, %zu records for %zu)\nr1 %s\nr2 %s\n", it, kind, flags, out.size(), want, r1.c_str(), r2.c_str()); return 1; }
  }
  printf("ok pairs %ld: stay %ld read-through %ld merged %ld one-mate %ld; read 1 listed %ld %ld %ld %ld; IsMateOverlap %ld calls, %ld overlaps\n", pairs, kinds[0], kinds[1], kinds[2], kinds[3],
         listed[0], listed[1], listed[2], listed[3], overlapCalls, overlapHits);
  return 0;
}

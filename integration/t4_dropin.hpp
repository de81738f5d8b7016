// integration/t4_dropin.hpp -- the reference-side binding of libt4hip.so (include/trust4_hip.h).
//
// This is the glue a TRUST4 maintainer adds to the reference's own main.cpp so that its three hot loops run on an MI355X through
// the C ABI while everything else -- option parsing, ProcessRead, k-mer counts, sorting, trimming, the loop bodies around
// AddRead, and the whole mate-pair extension tail (ExtendSeqFromReads, RemoveRedundantSeq, main.cpp:2153-2290) -- stays the
// reference's code, untouched:
//   main.cpp:1084-1120  refSet.AnnotateRead(read, 0, ...) of every distinct read   -> t4bind::RoughAnnotate  (t4_annotate_rough)
//   main.cpp:1583-1940  seqSet.AddRead / RepeatAddRead / InputNovelRead / ...      -> t4bind::SeqSetProxy    (t4_assembler_*)
//   main.cpp:2075-2118  extendedSeq.AssignRead of every assembled read,
//                       extendedSeq.RecomputePosWeight                              -> t4bind::AssignReads, t4bind::RecomputePosWeight
//                                                                                      (t4_assign_strands, t4_posweight_recompute)
// integration/make_dropin.py applies the seven one-line edits (listed there) to a COPY of /root/reference/main.cpp at build time
// and compiles it against this header; the result (oracle/_ref/trust4-dropin) is integration-test infrastructure: it proves the
// boundary on the reference's own driver, `_final.out` included (tests/test_run_trust4_dropin.py). No reference source is stored
// in this repository. Included after SeqSet.hpp, inside `#define private public` (the binding reads SeqSet::seqs, as the
// oracle's probe does); functions that touch main.cpp's own types (_sortRead, _assignRead) are templates because those are
// declared after the includes.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "trust4_hip.h"

namespace t4bind {

inline t4_ctx *Ctx() {
  static t4_ctx *c = NULL;
  if (!c) {
    const int dev = getenv("T4_DEVICE") ? atoi(getenv("T4_DEVICE")) : 0;
    if (t4_init(dev, &c) != T4_OK) {   // no CPU fallback behind this binding
      fprintf(stderr, "trust4 (MI355X binding): no usable GPU (t4_init failed)\n");
      exit(EXIT_FAILURE);
    }
  }
  return c;
}
inline void Check(int rc, const char *what) {
  if (rc == T4_OK) return;
  fprintf(stderr, "trust4 (MI355X binding): %s failed (%d): %s\n", what, rc, t4_last_error(Ctx()));   // the reference exit(1)s on errors too (main.cpp:887-891)
  exit(EXIT_FAILURE);
}
inline void FromT4(struct _overlap &g, const t4_overlap &o) {
  g.seqIdx = o.seqIdx; g.readStart = o.readStart; g.readEnd = o.readEnd; g.seqStart = o.seqStart; g.seqEnd = o.seqEnd;
  g.strand = o.strand; g.matchCnt = o.matchCnt; g.indelCnt = o.indelCnt; g.similarity = o.similarity;
}

// ---- main.cpp:1084-1120: rough V/J/C annotation of every distinct read --------------------------------------------------------
template <class Reads>
bool RoughAnnotate(SeqSet &refSet, Reads &sortedReads, int readCnt) {
  t4_index *ref = NULL;
  Check(t4_index_create(Ctx(), refSet.kmerLength, 0, &ref), "t4_index_create");
  Check(t4_index_set_params(ref, refSet.hitLenRequired, refSet.radius, 0.9), "t4_index_set_params");
  for (int i = 0; i < (int)refSet.seqs.size(); ++i) {   // the gene set as InputRefFa left it (filtered, de-duplicated, names merged)
    int id = -1;
    Check(t4_index_add_ref_record(ref, refSet.seqs[i].name, refSet.seqs[i].consensus, &id), "t4_index_add_ref_record");
    if (id != i) { fprintf(stderr, "trust4 (MI355X binding): gene %d (%s) became sequence %d of the device set\n", i, refSet.seqs[i].name, id); exit(EXIT_FAILURE); }
  }
  Check(t4_index_commit(ref), "t4_index_commit");
  const int CHUNK = 4000000;
  std::vector<t4_overlap> out;
  for (int begin = 0; begin < readCnt;) {
    std::string bases;
    std::vector<int64_t> off(1, 0);
    std::vector<int> firstOf;
    int end = begin;
    for (; end < readCnt && (int)firstOf.size() < CHUNK; ++end)
      if (end == 0 || strcmp(sortedReads[end].read, sortedReads[end - 1].read)) { bases += sortedReads[end].read; off.push_back((int64_t)bases.size()); firstOf.push_back(end); }
    while (end < readCnt && !strcmp(sortedReads[end].read, sortedReads[end - 1].read)) ++end;   // the copies of the chunk's last read
    const int n = (int)firstOf.size();
    if (n > 0) {
      t4_batch *batch = NULL;
      if (bases.empty()) bases.push_back('A');
      Check(t4_reads_upload(Ctx(), bases.data(), off.data(), NULL, n, &batch), "t4_reads_upload");
      out.resize(4 * (size_t)n);
      Check(t4_annotate_rough(ref, batch, out.data()), "t4_annotate_rough");
      t4_batch_destroy(batch);
    }
    // a read that equals its predecessor takes the predecessor's result (main.cpp:1088-1092); the first read of the chunk is distinct
    // from the last of the previous chunk by construction
    int k = -1;
    for (int i = begin; i < end; ++i) {
      if (k + 1 < n && firstOf[k + 1] == i) ++k;
      for (int j = 0; j < 4; ++j) FromT4(sortedReads[i].geneOverlap[j], out[4 * (size_t)k + j]);
    }
    begin = end;
  }
  t4_index_destroy(ref);
  return true;
}

// ---- main.cpp:642, 1528-1971: `SeqSet seqSet` (the novel contigs) behind the members main.cpp calls on it -------------------------
class SeqSetProxy {
 public:
  explicit SeqSetProxy(int kl) : k_(kl), hitLen_(31), considerBarcode_(false), a_(NULL), helper_(kl), mat_(NULL), cur_(-1), readCnt_(0), next_(NULL) {}
  ~SeqSetProxy() { if (mat_) delete mat_; if (a_) t4_assembler_destroy(a_); }
  operator SeqSet &() { return helper_; }   // ProcessRead takes a SeqSet& for ReverseComplementInPlace only (main.cpp:224-387)

  void SetHitLenRequired(int l) { hitLen_ = l; if (a_) t4_assembler_set_params(a_, hitLen_, 10, 0.9); }
  void SetConsiderBarcodeInIndexHash(bool s) { Fresh("SetConsiderBarcodeInIndexHash"); considerBarcode_ = s; }
  void SetIsLongSeqSet(bool in) { if (in) { fprintf(stderr, "trust4 (MI355X binding): long-read mode (first read > 200 bp) is not built\n"); exit(EXIT_FAILURE); } }
  void ChangeKmerLength(int kl) {
    k_ = kl;
    if (a_) Check(t4_assembler_change_kmer_length(a_, kl), "t4_assembler_change_kmer_length");
    Stale();
  }
  void InputNovelFa(char *filename) {   // SeqSet::InputNovelFa (SeqSet.hpp:2986-2993), --debug-ns
    ReadFiles fa;
    fa.AddReadFile(filename, false);
    while (fa.Next()) InputNovelRead(fa.id, fa.seq, 1, -1);
  }
  int Size() { return a_ ? t4_assembler_size(a_) : 0; }
  int GetSeqCnt() { return Size(); }
  int HasMotif(char *read, int strand) { return helper_.HasMotif(read, strand); }   // a pure function of the read (SeqSet.hpp:5029)

  int InputNovelRead(const char *id, char *read, int strand, int barcode) {
    const int r = t4_assembler_input_novel_read(A(), id, read, strand, barcode);
    if (r < -50) Check(r + 100, "t4_assembler_input_novel_read");
    Stale();
    return r;
  }
  int AddRead(char *read, char *geneName, int &strand, int barcode, int minKmerCount, bool repetitiveData, double similarityThreshold) {
    lastRead_ = read;
    if (next_ && cur_ >= 0 && !considerBarcode_ && !t4_assembler_window_valid(A())) next_(this, repetitiveData);   // speculation window over the upcoming reads
    const int r = t4_assembler_add_read(A(), read, geneName, &strand, barcode, minKmerCount, repetitiveData ? 1 : 0, similarityThreshold);
    if (r < -50) Check(r + 100, "t4_assembler_add_read");
    Stale();
    return r;
  }
  int RepeatAddRead(char *read) { Stale(); return t4_assembler_repeat_add_read(A(), read); }
  void UpdateAllConsensus() { if (a_) Check(t4_assembler_update_all_consensus(a_), "t4_assembler_update_all_consensus"); Stale(); }
  void ReleaseFinishedBarcodeSeq(std::map<int, int> barcodes, bool removeFromIndex, int contigMinCov, bool earlyStop) {
    (void)removeFromIndex; (void)earlyStop;   // main.cpp:1855 is the only caller: (true, true)
    for (std::map<int, int>::iterator it = barcodes.begin(); it != barcodes.end(); ++it) Check(t4_assembler_release_finished_barcode(A(), it->first, contigMinCov), "t4_assembler_release_finished_barcode");
    Stale();
  }
  void ReleaseShallowContigs(int minCov) { if (a_) Check(t4_assembler_release_shallow_contigs(a_, minCov), "t4_assembler_release_shallow_contigs"); Stale(); }
  void Output(FILE *fp, std::vector<std::string> *barcodeIntToStr = NULL) { Materialize().Output(fp, barcodeIntToStr); }

  // The engine's contigs as a reference SeqSet (every slot, released ones with a NULL consensus as SeqSet::ReleaseSeq leaves
  // them): what the host-side tail takes over (extendedSeq.InputSeqSet, main.cpp:2048) and what Output prints.
  SeqSet &Materialize() {
    if (mat_) return *mat_;
    mat_ = new SeqSet(k_);
    const int n = Size();
    for (int i = 0; i < n; ++i) {
      t4_contig_view v;
      Check(t4_assembler_contig(a_, i, &v), "t4_assembler_contig");
      struct _seqWrapper ns;
      ns.name = strdup(v.name);
      ns.consensus = v.consensus ? strdup(v.consensus) : NULL;
      ns.consensusLen = v.len;
      ns.isRef = false;
      ns.barcode = v.barcode; ns.numRead = v.num_read;
      ns.minLeftExtAnchor = v.min_left_ext_anchor; ns.minRightExtAnchor = v.min_right_ext_anchor;
      ns.index = v.in_index != 0; ns.posWeightCompressed = false;
      if (v.posweight) {
        ns.posWeight.ExpandTo(v.len);
        for (int j = 0; j < v.len; ++j) for (int c = 0; c < 4; ++c) ns.posWeight[j].count[c] = v.posweight[4 * j + c];
      }
      mat_->seqs.push_back(ns);
    }
    return *mat_;
  }

  // called at the top of every iteration of the assembly loop (main.cpp:1585): where the loop stands, so that AddRead can announce
  // the reads that follow. `next` restates, for upcoming reads, the two things of the loop body the announcement needs -- whether
  // the read is offered to AddRead at all and with which strand (main.cpp:1596-1674) -- from their rough annotations.
  void Step(int i, int readCnt, void (*next)(SeqSetProxy *, bool), int threadCnt) {
    cur_ = i; readCnt_ = readCnt; next_ = next;
    if (a_ && i == 0) t4_assembler_set_threads(a_, threadCnt);
    threads_ = threadCnt;
  }
  int Cur() const { return cur_; }
  const char *LastRead() const { return lastRead_; }
  int ReadCnt() const { return readCnt_; }
  void Prefetch(const std::vector<const char *> &reads, const std::vector<int> &strands, const std::vector<int> &barcodes, bool repetitive) {
    if (reads.empty()) return;
    Check(t4_assembler_prefetch(A(), (int)reads.size(), reads.data(), strands.data(), barcodes.data(), repetitive ? 1 : 0), "t4_assembler_prefetch");
  }

 private:
  t4_assembler *A() {
    if (!a_) {
      Check(t4_assembler_create(Ctx(), k_, considerBarcode_ ? 1 : 0, &a_), "t4_assembler_create");
      t4_assembler_set_params(a_, hitLen_, 10, 0.9);
      t4_assembler_set_threads(a_, threads_ > 0 ? threads_ : 1);
    }
    return a_;
  }
  void Fresh(const char *what) { if (a_) { fprintf(stderr, "trust4 (MI355X binding): %s after the set was used\n", what); exit(EXIT_FAILURE); } }
  void Stale() { if (mat_) { delete mat_; mat_ = NULL; } }
  int k_, hitLen_;
  bool considerBarcode_;
  t4_assembler *a_;
  SeqSet helper_;
  SeqSet *mat_;
  int cur_, readCnt_, threads_ = 1;
  const char *lastRead_ = NULL;
  void (*next_)(SeqSetProxy *, bool);
};

// The announcement of the assembly loop's next reads: every read from the loop's position on that the loop body will offer to
// AddRead (a new sequence / barcode, main.cpp:1596-1597, that its rough annotation does not filter, 1609-1651), with the strand
// argument the body derives (1662-1674). A wrong guess here could only cost a query: t4_assembler_add_read serves a window entry
// only when read, strand and barcode are the ones announced.
template <class Reads>
struct Announcer {
  static Reads *reads;
  static int constantGeneEnd, window;
  static void Next(SeqSetProxy *set, bool repetitive) {
    Reads &sr = *reads;
    std::vector<const char *> rs;
    std::vector<int> st, bc;
    const int n = set->ReadCnt();
    if (set->Cur() >= n || sr[set->Cur()].read != set->LastRead()) return;   // not the main pass (the rescue pass offers reads out of order, main.cpp:1904-1937)
    for (int j = set->Cur(); j < n && (int)rs.size() < window; ++j) {
      if (j > 0 && !strcmp(sr[j].read, sr[j - 1].read) && sr[j].barcode == sr[j - 1].barcode) continue;
      const struct _overlap *g = sr[j].geneOverlap;
      bool filter = false;
      for (int a = 0; a < 4 && !filter; ++a) {
        if (g[a].seqIdx == -1) continue;
        for (int b = a + 1; b < 4; ++b) { if (g[b].seqIdx == -1) continue; if (g[a].readEnd - 10 > g[b].readStart) { filter = true; break; } }
      }
      if (g[3].seqIdx != -1 && g[0].seqIdx == -1 && g[2].seqIdx == -1) {
        if (g[3].seqStart >= constantGeneEnd) filter = true;
        else if (constantGeneEnd <= 200 && g[3].seqStart >= 100 && (g[3].strand == 1 || g[3].readEnd - g[3].readStart + 1 < sr[j].len)) filter = true;
      }
      if (filter) continue;
      int strand = 0, ambiguous = 0;
      for (int a = 0; a < 4; ++a)
        if (g[a].seqIdx != -1) { if (strand != 0 && strand != g[a].strand) ambiguous = 1; strand = g[a].strand; }
      if (ambiguous) strand = 0;
      rs.push_back(sr[j].read); st.push_back(strand); bc.push_back(sr[j].barcode);
    }
    set->Prefetch(rs, st, bc, repetitive);
  }
};
template <class Reads> Reads *Announcer<Reads>::reads = NULL;
template <class Reads> int Announcer<Reads>::constantGeneEnd = 200;
template <class Reads> int Announcer<Reads>::window = 192;

template <class Reads>
void Step(SeqSetProxy &set, Reads &sortedReads, int i, int readCnt, int constantGeneEnd, int threadCnt) {
  Announcer<Reads>::reads = &sortedReads;
  Announcer<Reads>::constantGeneEnd = constantGeneEnd;
  if (getenv("T4_WINDOW")) Announcer<Reads>::window = atoi(getenv("T4_WINDOW"));
  set.Step(i, readCnt, Announcer<Reads>::window > 1 ? &Announcer<Reads>::Next : NULL, threadCnt);
}

// ---- main.cpp:2075-2118: AssignRead of every assembled read against extendedSeq, then RecomputePosWeight ---------------------------
struct TailState {
  t4_index *ix;
  t4_batch *batch;
  std::vector<int> firstOf, mult;       // distinct consecutive reads: index of the first copy, number of copies
  std::vector<t4_overlap> assign;       // AssignRead result per distinct read
  TailState() : ix(NULL), batch(NULL) {}
};
inline TailState &Tail() { static TailState t; return t; }

template <class AReads>
bool AssignReads(SeqSet &extendedSeq, AReads &assembledReads, int assembledReadCnt) {
  TailState &t = Tail();
  // device image of extendedSeq as InputSeqSet built it (SeqSet.hpp:3108-3140: every contig indexed by BuildIndexFromRead)
  Check(t4_index_create(Ctx(), extendedSeq.kmerLength, 0, &t.ix), "t4_index_create");
  Check(t4_index_set_params(t.ix, extendedSeq.hitLenRequired, extendedSeq.radius, extendedSeq.novelSeqSimilarity), "t4_index_set_params");
  for (int i = 0; i < (int)extendedSeq.seqs.size(); ++i) {
    struct _seqWrapper &s = extendedSeq.seqs[i];
    if (s.isRef || s.consensus == NULL || !s.index) { fprintf(stderr, "trust4 (MI355X binding): contig %d of the extension set is not an indexed novel contig\n", i); exit(EXIT_FAILURE); }
    int id = -1;
    Check(t4_index_add_contig(t.ix, s.name, s.consensus, s.barcode, (const int32_t *)s.posWeight.BeginAddress(), &id), "t4_index_add_contig");
    if (id != i) { fprintf(stderr, "trust4 (MI355X binding): contig %d became sequence %d of the device set\n", i, id); exit(EXIT_FAILURE); }
  }
  Check(t4_index_commit(t.ix), "t4_index_commit");
  // one AssignRead per run of identical consecutive reads, with the strand of the run's first read (main.cpp:2079-2082)
  std::string bases;
  std::vector<int64_t> off(1, 0);
  std::vector<int32_t> strands;
  for (int i = 0; i < assembledReadCnt; ++i) {
    if (i == 0 || strcmp(assembledReads[i].read, assembledReads[i - 1].read)) {
      bases += assembledReads[i].read; off.push_back((int64_t)bases.size());
      t.firstOf.push_back(i); t.mult.push_back(1); strands.push_back(assembledReads[i].overlap.strand);
      if (assembledReads[i].barcode != -1) { fprintf(stderr, "trust4 (MI355X binding): the extension tail is bulk mode only\n"); exit(EXIT_FAILURE); }
    } else ++t.mult.back();
  }
  const int n = (int)t.firstOf.size();
  if (bases.empty()) bases.push_back('A');
  Check(t4_reads_upload(Ctx(), bases.data(), off.data(), NULL, n, &t.batch), "t4_reads_upload");
  t.assign.resize(n > 0 ? n : 1);
  std::vector<int32_t> ret(n > 0 ? n : 1);
  if (n > 0) Check(t4_assign_strands(t.ix, t.batch, strands.data(), ret.data(), t.assign.data()), "t4_assign_strands");
  // `assign` is one variable across the loop (main.cpp:2050): a failed AssignRead only sets seqIdx = -1 (SeqSet.hpp:4641) and leaves
  // the other fields of the last success behind
  struct _overlap assign;
  int d = -1;
  for (int i = 0; i < assembledReadCnt; ++i) {
    if (d + 1 < n && t.firstOf[d + 1] == i) {
      ++d;
      if (ret[d] != -1) FromT4(assign, t.assign[d]); else { assign.seqIdx = -1; t.assign[d].seqIdx = -1; }
    }
    assembledReads[i].overlap = assign;
  }
  return true;
}

template <class AReads>
void RecomputePosWeight(SeqSet &extendedSeq, AReads &assembledReads) {
  (void)assembledReads;   // their assignments are the ones AssignReads left in Tail()
  TailState &t = Tail();
  int64_t bases = 0;
  for (int i = 0; i < (int)extendedSeq.seqs.size(); ++i) bases += extendedSeq.seqs[i].consensusLen;
  std::vector<int32_t> pw(4 * (size_t)(bases > 0 ? bases : 1));
  Check(t4_posweight_recompute(t.ix, t.batch, t.assign.data(), t.mult.data(), pw.data(), (int64_t)pw.size()), "t4_posweight_recompute");
  size_t at = 0;
  for (int i = 0; i < (int)extendedSeq.seqs.size(); ++i) {
    struct _seqWrapper &s = extendedSeq.seqs[i];
    for (int j = 0; j < s.consensusLen; ++j, ++at) for (int c = 0; c < 4; ++c) s.posWeight[j].count[c] = pw[4 * at + c];
  }
  t4_batch_destroy(t.batch); t.batch = NULL;
  t4_index_destroy(t.ix); t.ix = NULL;
}

}  // namespace t4bind

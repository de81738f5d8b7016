// oracle/ref_probe.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" shim around the UNMODIFIED reference headers as they lie in
// /root/reference (SeqSet.hpp / KmerIndex.hpp / AlignAlgo.hpp). It is compiled by
// oracle/Makefile into oracle/_ref/libt4ref.so and is used to
//   (1) pin the C restatement in oracle/t4_oracle.c against the real reference, and
//   (2) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py).
// No reference source is copied: the reference headers are #included from where they are.
//
// The private members of SeqSet are reached with the `#define private public`
// technique described in SURVEY.md section 8(c).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define private public
#include "SeqSet.hpp"
#include "KmerCount.hpp"
#undef private

// The two globals every reference TU must define (main.cpp:39-44).
int nucToNum[26] = {0, -1, 1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1,
                    0, -1, -1, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1};
char numToNuc[26] = {'A', 'C', 'G', 'T'};

extern "C" {

typedef struct {
  int seqIdx, readStart, readEnd, seqStart, seqEnd, strand, matchCnt, indelCnt;
  double similarity;
} ref_overlap_t;

void *ref_seqset_new(int k) { return new SeqSet(k); }
void ref_seqset_free(void *h) { delete (SeqSet *)h; }

void ref_input_ref_fa(void *h, const char *path) {
  ((SeqSet *)h)->InputRefFa((char *)path);
}
void ref_set_hit_len_required(void *h, int l) { ((SeqSet *)h)->SetHitLenRequired(l); }
void ref_set_radius(void *h, int r) { ((SeqSet *)h)->SetRadius(r); }
int ref_size(void *h) { return ((SeqSet *)h)->Size(); }
int ref_seq_len(void *h, int i) { return ((SeqSet *)h)->seqs[i].consensusLen; }
const char *ref_seq_name(void *h, int i) { return ((SeqSet *)h)->seqs[i].name; }
const char *ref_seq_consensus(void *h, int i) { return ((SeqSet *)h)->seqs[i].consensus; }
int ref_nomatch_gap_limit(void *h) { return ((SeqSet *)h)->nomatchGapLimit; }

// Seed a novel contig (SeqSet::InputNovelRead, SeqSet.hpp:3028).
int ref_input_novel_read(void *h, const char *id, const char *read, int strand, int barcode) {
  char *r = strdup(read);
  int ret = ((SeqSet *)h)->InputNovelRead(id, r, strand, barcode);
  free(r);
  return ret;
}

// posWeight of one contig, 4 ints per base.
void ref_seq_posweight(void *h, int i, int *out) {
  SeqSet *s = (SeqSet *)h;
  for (int j = 0; j < s->seqs[i].consensusLen; ++j)
    for (int c = 0; c < 4; ++c) out[4 * j + c] = s->seqs[i].posWeight[j].count[c];
}
// overwrite posWeight (to build test contigs with arbitrary weights)
void ref_seq_set_posweight(void *h, int i, const int *in) {
  SeqSet *s = (SeqSet *)h;
  for (int j = 0; j < s->seqs[i].consensusLen; ++j)
    for (int c = 0; c < 4; ++c) s->seqs[i].posWeight[j].count[c] = in[4 * j + c];
}

// GetHitsFromRead + SortHits (SeqSet.hpp:1341, 1306). Each hit -> 5 ints:
// idx, offset, readOffset, strand, repeats. Returns number of hits; fills at most cap.
int ref_hits(void *h, const char *read, int strand, int barcode, int allowTotalSkip, int doSort,
             int *out, int cap) {
  SeqSet *s = (SeqSet *)h;
  int len = strlen(read);
  char *r = strdup(read);
  char *rc = new char[len + 1];
  SimpleVector<struct _hit> hits;
  s->GetHitsFromRead(r, rc, strand, barcode, allowTotalSkip != 0, hits, NULL);
  if (doSort) s->SortHits(hits, true);
  int n = hits.Size();
  for (int i = 0; i < n && i < cap; ++i) {
    out[5 * i + 0] = hits[i].indexHit.idx;
    out[5 * i + 1] = hits[i].indexHit.offset;
    out[5 * i + 2] = hits[i].readOffset;
    out[5 * i + 3] = hits[i].strand;
    out[5 * i + 4] = hits[i].repeats;
  }
  delete[] rc;
  free(r);
  return n;
}

// GetHitsFromRead + SortHits + GetOverlapsFromHits (SeqSet.hpp:763). Overlaps are returned in
// emission order with their chains: chain k of overlap i lives in
// coords[2*(chainOff[i]+k) .. +1] = (a=readOffset, b=seqOffset).
int ref_overlaps_from_hits(void *h, const char *read, int strand, int barcode, int allowTotalSkip,
                           int hitLenRequired, int filter, ref_overlap_t *out, int cap,
                           int *chainOff, int *coords, int coordCap) {
  SeqSet *s = (SeqSet *)h;
  int len = strlen(read);
  char *r = strdup(read);
  char *rc = new char[len + 1];
  SimpleVector<struct _hit> hits;
  s->GetHitsFromRead(r, rc, strand, barcode, allowTotalSkip != 0, hits, NULL);
  s->SortHits(hits, true);
  std::vector<struct _overlap> ov;
  s->GetOverlapsFromHits(hits, hitLenRequired, filter, false, ov);
  int n = ov.size(), c = 0;
  for (int i = 0; i < n; ++i) {
    if (i < cap) {
      out[i].seqIdx = ov[i].seqIdx; out[i].readStart = ov[i].readStart;
      out[i].readEnd = ov[i].readEnd; out[i].seqStart = ov[i].seqStart;
      out[i].seqEnd = ov[i].seqEnd; out[i].strand = ov[i].strand;
      out[i].matchCnt = ov[i].matchCnt; out[i].indelCnt = ov[i].indelCnt;
      out[i].similarity = ov[i].similarity;
      chainOff[i] = c;
    }
    int m = ov[i].hitCoords->Size();
    for (int k = 0; k < m; ++k) {
      if (c < coordCap) { coords[2 * c] = ov[i].hitCoords->Get(k).a; coords[2 * c + 1] = ov[i].hitCoords->Get(k).b; }
      ++c;
    }
    delete ov[i].hitCoords;
  }
  if (n < cap) chainOff[n] = c;
  delete[] rc;
  free(r);
  return n;
}

static void copy_overlap(ref_overlap_t *o, const struct _overlap &v) {
  o->seqIdx = v.seqIdx; o->readStart = v.readStart; o->readEnd = v.readEnd;
  o->seqStart = v.seqStart; o->seqEnd = v.seqEnd; o->strand = v.strand;
  o->matchCnt = v.matchCnt; o->indelCnt = v.indelCnt; o->similarity = v.similarity;
}

// GetOverlapsFromRead (SeqSet.hpp:1508). Returns the function's return value (-1 / count).
int ref_overlaps_from_read(void *h, const char *read, int strand, int barcode, int readType,
                           int skipRepeats, ref_overlap_t *out, int cap) {
  SeqSet *s = (SeqSet *)h;
  char *r = strdup(read);
  std::vector<struct _overlap> ov;
  int ret = s->GetOverlapsFromRead(r, strand, barcode, readType, skipRepeats != 0, ov);
  for (int i = 0; i < (int)ov.size() && i < cap; ++i) copy_overlap(out + i, ov[i]);
  free(r);
  return ret;
}

// AnnotateRead(read, 0, geneOverlap, NULL, NULL) (SeqSet.hpp:6016), as called by main.cpp:1089.
int ref_annotate_read0(void *h, const char *read, ref_overlap_t out[4]) {
  SeqSet *s = (SeqSet *)h;
  char *r = strdup(read);
  struct _overlap g[4];
  int ret = s->AnnotateRead(r, 0, g, NULL, NULL);
  for (int i = 0; i < 4; ++i) copy_overlap(out + i, g[i]);
  free(r);
  return ret;
}

// ExtendOverlap (SeqSet.hpp:1165). `in`/`out` are single overlaps. Returns its return value.
int ref_extend_overlap(void *h, const char *read, double mmFactor, const ref_overlap_t *in,
                       ref_overlap_t *out) {
  SeqSet *s = (SeqSet *)h;
  int len = strlen(read);
  char *r = strdup(read);
  signed char *align = new signed char[2 * len + 4 + 2 * s->seqs[in->seqIdx].consensusLen];
  struct _overlap a, b;
  a.seqIdx = in->seqIdx; a.readStart = in->readStart; a.readEnd = in->readEnd;
  a.seqStart = in->seqStart; a.seqEnd = in->seqEnd; a.strand = in->strand;
  a.matchCnt = in->matchCnt; a.indelCnt = in->indelCnt; a.similarity = in->similarity;
  int ret = s->ExtendOverlap(r, len, s->seqs[in->seqIdx], mmFactor, align, a, b);
  copy_overlap(out, b);
  delete[] align;
  free(r);
  return ret;
}

// AssignRead (SeqSet.hpp:4632).
int ref_assign_read(void *h, const char *read, int strand, int barcode, ref_overlap_t *out) {
  SeqSet *s = (SeqSet *)h;
  char *r = strdup(read);
  struct _overlap a;
  int ret = s->AssignRead(r, strand, barcode, a);
  copy_overlap(out, a);
  free(r);
  return ret;
}

// RecomputePosWeight (SeqSet.hpp:4705-4738) from assigned reads (main.cpp:2118).
void ref_recompute_posweight(void *h, int n, const char *const *reads, const ref_overlap_t *assign) {
  SeqSet *s = (SeqSet *)h;
  std::vector<struct _assignRead> v((size_t)n);
  for (int i = 0; i < n; ++i) {
    v[i].id = NULL; v[i].read = strdup(reads[i]); v[i].barcode = -1; v[i].umi = -1; v[i].info = i;
    v[i].overlap.seqIdx = assign[i].seqIdx; v[i].overlap.strand = assign[i].strand; v[i].overlap.seqStart = assign[i].seqStart;
    v[i].overlap.seqEnd = assign[i].seqEnd; v[i].overlap.readStart = assign[i].readStart; v[i].overlap.readEnd = assign[i].readEnd;
  }
  s->RecomputePosWeight(v);
  for (int i = 0; i < n; ++i) free(v[i].read);
}
void ref_set_novel_seq_similarity(void *h, double v) { ((SeqSet *)h)->SetNovelSeqSimilarity(v); }

// AddRead (SeqSet.hpp:3426) -- mutates the set.
int ref_add_read(void *h, const char *read, const char *geneName, int *strand, int barcode,
                 int minKmerCount, int repetitiveData, double similarityThreshold) {
  SeqSet *s = (SeqSet *)h;
  char *r = strdup(read);
  char *g = strdup(geneName);
  int st = *strand;
  int ret = s->AddRead(r, g, st, barcode, minKmerCount, repetitiveData != 0, similarityThreshold);
  *strand = st;
  free(r);
  free(g);
  return ret;
}
int ref_repeat_add_read(void *h, const char *read) {
  SeqSet *s = (SeqSet *)h;
  char *r = strdup(read);
  int ret = s->RepeatAddRead(r);
  free(r);
  return ret;
}
// SeqSet::HasHitInSet (SeqSet.hpp:3144-3327): the stage-0 candidate test of FastqExtractor.cpp:129-134
int ref_has_hit_in_set(void *h, const char *read, int mode) {
  char *r = strdup(read);
  int ret = ((SeqSet *)h)->HasHitInSet(r, mode);
  free(r);
  return ret;
}
void ref_update_all_consensus(void *h) { ((SeqSet *)h)->UpdateAllConsensus(); }
void ref_change_kmer_length(void *h, int k) { ((SeqSet *)h)->ChangeKmerLength(k); }   // SeqSet.hpp:4624-4629 (main.cpp:1874-1879)
// barcode mode (main.cpp:1549-1559, 1846-1859, 1968-1969)
void ref_set_consider_barcode(void *h, int on) { ((SeqSet *)h)->SetConsiderBarcodeInIndexHash(on != 0); }
void ref_release_finished_barcode(void *h, int barcode, int total) {
  std::map<int, int> fin;
  fin[barcode] = total;
  ((SeqSet *)h)->ReleaseFinishedBarcodeSeq(fin, true, 0, true);
}
void ref_output_barcodes(void *h, const char *path, const char *const *names, int n) {
  std::vector<std::string> v;
  for (int i = 0; i < n; ++i) v.push_back(names[i]);
  FILE *fp = fopen(path, "w");
  ((SeqSet *)h)->Output(fp, &v);
  fclose(fp);
}
void ref_output(void *h, const char *path) {
  FILE *fp = fopen(path, "w");
  ((SeqSet *)h)->Output(fp);
  fclose(fp);
}

// AlignAlgo::GlobalAlignment (AlignAlgo.hpp:218). align must hold lent+lenp+2 entries.
int ref_global_alignment(const char *t, int lent, const char *p, int lenp, signed char *align) {
  return AlignAlgo::GlobalAlignment((char *)t, lent, (char *)p, lenp, align);
}
// AlignAlgo::GlobalAlignment_PosWeight (AlignAlgo.hpp:57). w = 4 ints per target base.
int ref_global_alignment_posweight(const int *w, int lent, const char *p, int lenp,
                                   signed char *align) {
  struct _posWeight *pw = new struct _posWeight[lent + 1];
  for (int i = 0; i < lent; ++i)
    for (int c = 0; c < 4; ++c) pw[i].count[c] = w[4 * i + c];
  int ret = (int)AlignAlgo::GlobalAlignment_PosWeight(pw, lent, (char *)p, lenp, align);
  delete[] pw;
  return ret;
}
// AlignAlgo::IsMateOverlap (AlignAlgo.hpp:1027).
int ref_is_mate_overlap(const char *fr, int flen, const char *sr, int slen, int minOverlap,
                        int *offset, int *bestMatchCnt, int checkTandem) {
  int off = -1, bm = -1;
  int ret = AlignAlgo::IsMateOverlap((char *)fr, flen, (char *)sr, slen, minOverlap, off, bm,
                                     checkTandem != 0);
  *offset = off;
  *bestMatchCnt = bm;
  return ret;
}

// LongestIncreasingSubsequence (SeqSet.hpp:342). pairs in: (a,b) sorted by b. Returns LIS size.
int ref_lis(void *h, const int *pairs, int n, int *out) {
  SeqSet *s = (SeqSet *)h;
  SimpleVector<struct _pair> in, lis;
  for (int i = 0; i < n; ++i) { struct _pair p; p.a = pairs[2 * i]; p.b = pairs[2 * i + 1]; in.PushBack(p); }
  int r = s->LongestIncreasingSubsequence(in, lis);
  for (int i = 0; i < r; ++i) { out[2 * i] = lis[i].a; out[2 * i + 1] = lis[i].b; }
  return r;
}

}  // extern "C"

// Batch form of AnnotateRead(level 0) for bench.py's cpu_baseline leg ("reference" kind): reads are
// fixed-stride NUL-terminated records; returns the number of reads annotated.
extern "C" long ref_annotate_batch(void *h, const char *reads, int stride, long n, ref_overlap_t *out4) {
  SeqSet *s = (SeqSet *)h;
  struct _overlap g[4];
  for (long i = 0; i < n; ++i) {
    s->AnnotateRead((char *)(reads + i * stride), 0, g, NULL, NULL);
    if (out4) for (int t = 0; t < 4; ++t) copy_overlap(out4 + 4 * i + t, g[t]);
  }
  return n;
}

// KmerCount (KmerCount.hpp): the canonical 21-mer counts of stage 1 (main.cpp:905-915) and GetCountStatsAndTrim (980-1061).
extern "C" void *ref_kc_new(int k) { return new KmerCount(k); }
extern "C" void ref_kc_free(void *h) { delete (KmerCount *)h; }
extern "C" int ref_kc_add(void *h, const char *read) { return ((KmerCount *)h)->AddCount((char *)read); }
// read / qual are modified in place as the reference does (NUL at the trim position); qual may be NULL
extern "C" int ref_kc_stats(void *h, char *read, char *qual, int *minCnt, int *medianCnt, float *avgCnt) {
  return ((KmerCount *)h)->GetCountStatsAndTrim(read, qual, *minCnt, *medianCnt, *avgCnt);
}

/* oracle/t4_oracle.c -- CPU ORACLE for the TRUST4 stage-1 seed -> chain -> extend path.
 *
 * TEST INFRASTRUCTURE, NOT THE PRODUCT. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (trust4_amd/libt4hip.so) never does.
 *
 * This is a plain-C restatement (written from the algorithm's behaviour, array based, no std::map)
 * of the reference functions listed below; every function cites the reference file:line it
 * follows. Parity is PINNED: tests/test_oracle_vs_ref.py checks every entry point of this file
 * against the unmodified reference compiled into oracle/_ref/libt4ref.so (oracle/ref_probe.cpp),
 * and tests/golden/ holds vectors generated from that reference build.
 *
 *   KmerCode.hpp:94-109     kc_append            KmerIndex.hpp:29-33,66-141  index_*
 *   SeqSet.hpp:1341-1501    get_hits             SeqSet.hpp:1306-1339        sort_hits
 *   SeqSet.hpp:342-499      lis                  SeqSet.hpp:763-1063         overlaps_from_hits
 *   SeqSet.hpp:1066-1161    vj_overlaps          SeqSet.hpp:1508-2124        overlaps_from_read
 *   SeqSet.hpp:1165-1277    extend_overlap       SeqSet.hpp:4632-4701        assign_read
 *   SeqSet.hpp:6016-6321    annotate_read0       SeqSet.hpp:2673-2865        add_ref_record
 *   AlignAlgo.hpp:57-216    ga_posweight         AlignAlgo.hpp:218-424       ga_affine
 *   AlignAlgo.hpp:1027-1096 is_mate_overlap
 * Long-read mode (isLongSeqSet) and readType 1 are outside the stage-1 150 bp scope and are not
 * restated.
 */
#include "t4_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define KINDEX_HASH_MAX 1000003
#define ABSI(x) ((x) < 0 ? -(x) : (x))
#define MINI(a, b) ((a) < (b) ? (a) : (b))
#define MAXI(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------ nucleotide tables */
/* main.cpp:39-44 */
static const int NUC2NUM[26] = {0, -1, 1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1,
                                0, -1, -1, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1};
static const char NUM2NUC[4] = {'A', 'C', 'G', 'T'};
static int nuc(char c) { return (c >= 'A' && c <= 'Z') ? NUC2NUM[c - 'A'] : -1; }

/* SeqSet.hpp:2628-2639 */
static void reverse_complement(char *rc, const char *seq, int len) {
  for (int i = 0; i < len; ++i) {
    char c = seq[len - 1 - i];
    rc[i] = (c != 'N') ? NUM2NUC[3 - nuc(c)] : 'N';
  }
  rc[len] = 0;
}

/* ------------------------------------------------------------------ KmerCode (KmerCode.hpp) */
typedef struct { int k, invalidPos; uint64_t code, mask; } kcode;
static void kc_init(kcode *c, int k) {
  c->k = k; c->code = 0; c->invalidPos = -1;
  c->mask = k < 32 ? ((1ull << (2 * k)) - 1ull) : (uint64_t)-1;
}
static void kc_restart(kcode *c) { c->code = 0; c->invalidPos = -1; }
/* KmerCode.hpp:94-109 */
static void kc_append(kcode *c, char ch) {
  if (c->invalidPos != -1) ++c->invalidPos;
  c->code = ((c->code << 2) & c->mask) | (uint64_t)(nuc(ch) & 3);
  if (ch == 'N') c->invalidPos = 0;
  if (c->invalidPos >= c->k) c->invalidPos = -1;
}
static int kc_valid(const kcode *c) { return c->invalidPos == -1; }

/* ------------------------------------------------------------------ k-mer index (KmerIndex.hpp) */
typedef struct { int idx, offset; } post_t;
typedef struct { uint64_t code; int h; post_t *p; int n, cap; int used; } slot_t;
typedef struct { slot_t *s; uint64_t nslots, nused; int considerBarcode; } kindex;

static void index_init(kindex *ix) {
  ix->nslots = 1 << 16; ix->nused = 0; ix->considerBarcode = 0;
  ix->s = (slot_t *)calloc(ix->nslots, sizeof(slot_t));
}
static void index_free(kindex *ix) {
  for (uint64_t i = 0; i < ix->nslots; ++i) free(ix->s[i].p);
  free(ix->s);
}
/* KmerIndex.hpp:29-33 */
static int index_hash(const kindex *ix, uint64_t code, int barcode) {
  return (int)((code + (uint64_t)(int64_t)(ix->considerBarcode ? (barcode + 1) : 0)) % KINDEX_HASH_MAX);
}
static uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static slot_t *index_find(const kindex *ix, uint64_t code, int h, int create) {
  uint64_t m = ix->nslots - 1, i = mix64(code * 1000003ull + (uint64_t)h) & m;
  while (ix->s[i].used) {
    if (ix->s[i].code == code && ix->s[i].h == h) return &ix->s[i];
    i = (i + 1) & m;
  }
  if (!create) return NULL;
  ix->s[i].used = 1; ix->s[i].code = code; ix->s[i].h = h;
  return &ix->s[i];
}
static void index_grow(kindex *ix) {
  slot_t *old = ix->s; uint64_t on = ix->nslots;
  ix->nslots *= 2; ix->s = (slot_t *)calloc(ix->nslots, sizeof(slot_t));
  for (uint64_t i = 0; i < on; ++i)
    if (old[i].used) { slot_t *d = index_find(ix, old[i].code, old[i].h, 1); *d = old[i]; }
  free(old);
}
/* KmerIndex.hpp:66-79 */
static void index_insert(kindex *ix, const kcode *kc, int idx, int offset, int barcode) {
  if (!kc_valid(kc)) return;
  if ((ix->nused + 1) * 2 > ix->nslots) index_grow(ix);
  int h = index_hash(ix, kc->code, barcode);
  slot_t *s = index_find(ix, kc->code, h, 0);
  if (!s) { s = index_find(ix, kc->code, h, 1); ++ix->nused; }
  if (s->n == s->cap) { s->cap = s->cap ? 2 * s->cap : 4; s->p = (post_t *)realloc(s->p, sizeof(post_t) * s->cap); }
  s->p[s->n].idx = idx; s->p[s->n].offset = offset; ++s->n;
}
/* KmerIndex.hpp:104-116 */
static const slot_t *index_search(const kindex *ix, const kcode *kc, int barcode) {
  if (!kc_valid(kc)) return NULL;
  return index_find(ix, kc->code, index_hash(ix, kc->code, barcode), 0);
}
/* KmerIndex.hpp:118-141 (note the `i == kl` quirk and the all-zero initial prev code) */
static void index_build_from_seq(kindex *ix, int k, const char *s, int len, int id, int barcode, int shift) {
  if (len < k) return;
  kcode kc, prev; kc_init(&kc, k); kc_init(&prev, k);
  int i;
  for (i = 0; i < k - 1; ++i) kc_append(&kc, s[i]);
  for (; i < len; ++i) {
    kc_append(&kc, s[i]);
    if (kc_valid(&kc) && (i == k || kc.code != prev.code)) index_insert(ix, &kc, id, i - k + 1 + shift, barcode);
    prev = kc;
  }
}

/* ------------------------------------------------------------------ the sequence set */
typedef struct {
  char *name, *cons; int len, isRef, barcode; int *pw; /* 4 ints per base, novel seqs only */
} seq_t;

struct t4o_set {
  int k, radius, hitLenRequired, gapN, nomatchGapLimit;
  double novelSim, refSim, repeatSim;
  seq_t *seqs; int nseq, capseq;
  kindex ix;
};

/* SeqSet.hpp:2476-2482, 2557-2576 */
t4o_set *t4o_new(int k) {
  t4o_set *s = (t4o_set *)calloc(1, sizeof(t4o_set));
  s->k = k; s->radius = 10; s->hitLenRequired = 31; s->gapN = 7;
  s->novelSim = 0.9; s->refSim = 0.75; s->repeatSim = 0.95;
  double kmerHitProb = pow(0.8, k);
  s->nomatchGapLimit = (int)(k * (log(0.01) / log(1 - kmerHitProb))) + 1;
  index_init(&s->ix);
  return s;
}
void t4o_free(t4o_set *s) {
  for (int i = 0; i < s->nseq; ++i) { free(s->seqs[i].name); free(s->seqs[i].cons); free(s->seqs[i].pw); }
  free(s->seqs); index_free(&s->ix); free(s);
}
void t4o_set_hit_len_required(t4o_set *s, int l) { s->hitLenRequired = l; }
void t4o_set_radius(t4o_set *s, int r) { s->radius = r; }
void t4o_set_novel_similarity(t4o_set *s, double v) { s->novelSim = v; }
void t4o_set_consider_barcode(t4o_set *s, int v) { s->ix.considerBarcode = v; }
int t4o_size(const t4o_set *s) { return s->nseq; }
int t4o_seq_len(const t4o_set *s, int i) { return s->seqs[i].len; }
const char *t4o_seq_name(const t4o_set *s, int i) { return s->seqs[i].name; }
const char *t4o_seq_consensus(const t4o_set *s, int i) { return s->seqs[i].cons; }
int t4o_nomatch_gap_limit(const t4o_set *s) { return s->nomatchGapLimit; }

static int set_push(t4o_set *s) {
  if (s->nseq == s->capseq) { s->capseq = s->capseq ? 2 * s->capseq : 64; s->seqs = (seq_t *)realloc(s->seqs, sizeof(seq_t) * s->capseq); }
  memset(&s->seqs[s->nseq], 0, sizeof(seq_t));
  return s->nseq++;
}

/* SeqSet.hpp:5132-5155 */
static int chain_type(const char *name) {
  if (name[0] == 'I') { if (name[2] == 'H') return 0; if (name[2] == 'K') return 1; if (name[2] == 'L') return 2; }
  else if (name[0] == 'T') { if (name[2] == 'A') return 3; if (name[2] == 'B') return 4; if (name[2] == 'G') return 5; if (name[2] == 'D') return 6; }
  return 8;
}
/* SeqSet.hpp:5076-5100: 0 V, 1 D, 2 J, 3 C, -1 other */
static int gene_type(const char *name) {
  if (name[0] == 'N' && name[1] == 'o') return -1;
  switch (name[3]) {
    case 'V': return 0;
    case 'D': return (name[4] >= '0' && name[4] <= '9') ? 1 : 3;
    case 'J': return 2;
    case 'L': if (chain_type(name) == 2) return -1; return 3;
    default: return 3;
  }
}

/* One record of InputRefFa (SeqSet.hpp:2691-2865, non-IMGT branch). Returns the seq id or -1 when
 * the record was filtered or merged into an existing identical sequence. */
int t4o_add_ref_record(t4o_set *s, const char *id, const char *seq) {
  int i, k;
  if (gene_type(id) != 1) {
    for (i = 0; id[i]; ++i)
      if (id[i] == '/' && id[i + 1] == 'O' && id[i + 2] == 'R') break;
    if (id[i] == '/') return -1;
  }
  int seqLen = (int)strlen(seq);
  char *cons = (char *)malloc(seqLen + 1);
  k = 0;
  for (i = 0; i < seqLen; ++i) {
    if (seq[i] == '.') continue;
    int c = (signed char)seq[i];
    if (c >= 'a' && c <= 'z') c = (signed char)(c - ('a' + 'A')); /* the reference's arithmetic: lower case ends up as N */
    if (c >= 'A' && c <= 'Z') { if (NUC2NUM[c - 'A'] == -1 && c != 'N') c = 'N'; }
    else c = 'N';
    cons[k++] = (char)c;
  }
  cons[k] = 0;
  for (i = 0; i < s->nseq; ++i) /* dedup: first identical earlier sequence */
    if (s->seqs[i].len == k && !strcmp(s->seqs[i].cons, cons)) break;
  if (i < s->nseq) {
    if (strstr(s->seqs[i].name, id) == NULL) {
      size_t li = strlen(s->seqs[i].name), lc = strlen(id);
      char *t = (char *)malloc(li + lc + 2);
      strcpy(t, s->seqs[i].name); t[li] = '|'; strcpy(t + li + 1, id);
      free(s->seqs[i].name); s->seqs[i].name = t;
    }
    free(cons);
    return -1;
  }
  int sid = set_push(s);
  seq_t *q = &s->seqs[sid];
  q->name = strdup(id); q->cons = cons; q->len = k; q->isRef = 1; q->barcode = -1; q->pw = NULL;
  index_build_from_seq(&s->ix, s->k, cons, k, sid, -1, 0);
  return sid;
}

int t4o_load_ref_fasta(t4o_set *s, const char *path) {
  gzFile fp = gzopen(path, "rb");
  if (!fp) return -1;
  size_t cap = 1 << 16, len = 0; char *seq = (char *)malloc(cap); char id[4096]; int have = 0;
  char *line = (char *)malloc(1 << 20);
  seq[0] = 0;
  while (gzgets(fp, line, 1 << 20)) {
    size_t l = strlen(line);
    while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
    if (line[0] == '>') {
      if (have) t4o_add_ref_record(s, id, seq);
      size_t i = 0;
      while (line[1 + i] && line[1 + i] != ' ' && line[1 + i] != '\t' && i < sizeof(id) - 1) { id[i] = line[1 + i]; ++i; }
      id[i] = 0;
      /* ReadFiles.hpp:180-185: a trailing /1 or /2 is stripped from ids */
      if (i >= 2 && (id[i - 1] == '1' || id[i - 1] == '2') && id[i - 2] == '/') id[i - 2] = 0;
      have = 1; len = 0; seq[0] = 0;
    } else if (have) {
      if (len + l + 1 > cap) { while (len + l + 1 > cap) cap *= 2; seq = (char *)realloc(seq, cap); }
      memcpy(seq + len, line, l); len += l; seq[len] = 0;
    }
  }
  if (have) t4o_add_ref_record(s, id, seq);
  free(line); free(seq); gzclose(fp);
  return s->nseq;
}

/* SeqSet.hpp:3028-3073 */
int t4o_add_novel_seq(t4o_set *s, const char *name, const char *seq, int strand, int barcode, const int *posweight) {
  int sid = set_push(s);
  seq_t *q = &s->seqs[sid];
  int len = (int)strlen(seq);
  q->name = strdup(name); q->cons = (char *)malloc(len + 1);
  if (strand == -1) reverse_complement(q->cons, seq, len); else strcpy(q->cons, seq);
  q->len = len; q->isRef = 0; q->barcode = barcode;
  q->pw = (int *)calloc(4 * (size_t)len + 4, sizeof(int));
  if (posweight) memcpy(q->pw, posweight, sizeof(int) * 4 * (size_t)len);
  else for (int i = 0; i < len; ++i) if (q->cons[i] != 'N') q->pw[4 * i + nuc(q->cons[i])] = 1;
  index_build_from_seq(&s->ix, s->k, q->cons, len, sid, barcode, 0);
  return sid;
}

/* ------------------------------------------------------------------ hits */
typedef struct { int idx, offset, readOffset, strand, repeats; } hit_t;
typedef struct { hit_t *h; int n, cap; } hitvec;
static void hv_push(hitvec *v, hit_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->h = (hit_t *)realloc(v->h, sizeof(hit_t) * v->cap); }
  v->h[v->n++] = x;
}

/* One strand of SeqSet::GetHitsFromRead (1365-1426 / 1430-1499); prev persists across strands. */
static void hits_one_strand(t4o_set *s, const char *r, int len, int strandTag, int barcode, int allowTotalSkip,
                            kcode *kc, kcode *prev, int skipLimit, hitvec *hits) {
  int k = s->k, i, skipCnt = 0;
  for (i = 0; i < k - 1; ++i) kc_append(kc, r[i]);
  for (; i < len; ++i) {
    kc_append(kc, r[i]);
    if (i == k - 1 || prev->code != kc->code) {
      const slot_t *sl = index_search(&s->ix, kc, barcode);
      int size = sl ? sl->n : 0;
      if (size >= 100 && i != k - 1 && i != len - 1 && skipCnt < skipLimit) { ++skipCnt; continue; }
      if (size >= 100 && allowTotalSkip) continue;
      skipCnt = 0;
      int repeats = (barcode != -1) ? 1 : size;
      for (int j = 0; j < size; ++j) {
        if (barcode != -1 && s->seqs[sl->p[j].idx].barcode != barcode) continue;
        hit_t h; h.idx = sl->p[j].idx; h.offset = sl->p[j].offset; h.readOffset = i - k + 1; h.strand = strandTag; h.repeats = repeats;
        hv_push(hits, h);
      }
    }
    *prev = *kc;
  }
}

/* SeqSet.hpp:1341-1501 (puse == NULL, no long-read down-sampling) */
static int get_hits(t4o_set *s, const char *read, char *rc, int len, int strand, int barcode, int allowTotalSkip, hitvec *hits) {
  kcode kc, prev; kc_init(&kc, s->k); kc_init(&prev, s->k);
  int skipLimit = s->k / 2;
  if (s->nseq > 0 && s->seqs[0].isRef) skipLimit = 0;
  if (strand != -1) hits_one_strand(s, read, len, 1, barcode, allowTotalSkip, &kc, &prev, skipLimit, hits);
  reverse_complement(rc, read, len);
  if (strand != 1) { kc_restart(&kc); hits_one_strand(s, rc, len, -1, barcode, allowTotalSkip, &kc, &prev, skipLimit, hits); }
  return hits->n;
}

/* generic stable merge sort */
typedef int (*cmp_fn)(const void *, const void *);
static void msort(void *base, size_t n, size_t sz, cmp_fn lt) {
  if (n < 2) return;
  char *a = (char *)base, *tmp = (char *)malloc(n * sz);
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, o = lo;
      while (i < mid && j < hi) {
        if (lt(a + j * sz, a + i * sz)) { memcpy(tmp + o * sz, a + j * sz, sz); ++j; }
        else { memcpy(tmp + o * sz, a + i * sz, sz); ++i; }
        ++o;
      }
      while (i < mid) { memcpy(tmp + o * sz, a + i * sz, sz); ++i; ++o; }
      while (j < hi) { memcpy(tmp + o * sz, a + j * sz, sz); ++j; ++o; }
    }
    memcpy(a, tmp, n * sz);
  }
  free(tmp);
}

static int hit_lt_bucket(const void *x, const void *y) { /* SortHits bucket path: (strand==1, idx), stable */
  const hit_t *a = (const hit_t *)x, *b = (const hit_t *)y;
  int ta = a->strand == 1, tb = b->strand == 1;
  if (ta != tb) return ta < tb;
  return a->idx < b->idx;
}
static int hit_lt_full(const void *x, const void *y) { /* _hit::operator< (SeqSet.hpp:61-73) */
  const hit_t *a = (const hit_t *)x, *b = (const hit_t *)y;
  if (a->strand != b->strand) return a->strand < b->strand;
  if (a->idx != b->idx) return a->idx < b->idx;
  if (a->readOffset != b->readOffset) return a->readOffset < b->readOffset;
  if (a->offset != b->offset) return a->offset < b->offset;
  return 0;
}
/* SeqSet.hpp:1306-1339 */
static void sort_hits(t4o_set *s, hitvec *hits) {
  if ((unsigned)hits->n > 2u * (unsigned)s->nseq) msort(hits->h, hits->n, sizeof(hit_t), hit_lt_bucket);
  else msort(hits->h, hits->n, sizeof(hit_t), hit_lt_full);
}

/* ------------------------------------------------------------------ LIS (SeqSet.hpp:316-499) */
typedef struct { int a, b; } pair_t;
static int bsearch_lis(const int *top, int size, int valA, const pair_t *hits) {
  int l = 0, r = size - 1, m;
  while (l <= r) {
    m = (l + r) / 2;
    if (valA == hits[top[m]].a) return m;
    else if (valA < hits[top[m]].a) r = m - 1;
    else l = m + 1;
  }
  return l - 1;
}
static double dabs(double x) { return x < 0 ? -x : x; }
static int lis(const pair_t *hits, int size, pair_t *LIS) {
  int i, j, k, ret;
  int *top = (int *)malloc(sizeof(int) * size), *link = (int *)malloc(sizeof(int) * size);
  double avgDiff = 0;
  for (i = 1; i < size; ++i) avgDiff += (hits[i].a - hits[i].b);
  avgDiff /= size;
  top[0] = 0; link[0] = -1; ret = 1;
  for (i = 1; i < size; ++i) {
    int tag;
    if (hits[top[ret - 1]].a <= hits[i].a) tag = ret - 1;
    else tag = bsearch_lis(top, ret, hits[i].a, hits);
    if (tag == -1) { top[0] = i; link[i] = -1; }
    else if (hits[i].a > hits[top[tag]].a) {
      if (tag == ret - 1) { top[ret] = i; ++ret; link[i] = top[tag]; }
      else if (hits[i].a < hits[top[tag + 1]].a) { top[tag + 1] = i; link[i] = top[tag]; }
    } else if (hits[i].a == hits[top[tag]].a) {
      if (dabs(hits[i].a - hits[i].b - avgDiff) < dabs(hits[top[tag]].a - hits[top[tag]].b - avgDiff)) {
        top[tag] = i;
        link[i] = tag > 0 ? top[tag - 1] : -1;
      }
    }
  }
  k = top[ret - 1];
  for (i = ret - 1; i >= 0; --i) { LIS[i] = hits[k]; k = link[k]; }
  /* collapse equal-b runs, keep least divergence (first on ties) */
  k = 0;
  for (i = 0; i < ret;) {
    for (j = i + 1; j < ret; ++j) if (LIS[i].b != LIS[j].b) break;
    if (j == i + 1) LIS[k] = LIS[i];
    else {
      int l, mintag = i; double minDiff = dabs(LIS[i].a - LIS[i].b - avgDiff);
      for (l = i + 1; l < j; ++l)
        if (dabs(LIS[l].a - LIS[l].b - avgDiff) < minDiff) { minDiff = dabs(LIS[l].a - LIS[l].b - avgDiff); mintag = l; }
      LIS[k] = LIS[mintag];
    }
    i = j; ++k;
  }
  ret = k;
  /* replacement sweep */
  i = 0; j = 0;
  while (i < ret && j < size) {
    if (hits[j].b < LIS[i].b) ++j;
    else if (i + 1 < ret && LIS[i + 1].b <= hits[j].b) ++i;
    else if (LIS[i].a == hits[j].a && LIS[i].b == hits[j].b) ++j;
    else {
      if (LIS[i].a <= hits[j].a && (i == ret - 1 || hits[j].a < LIS[i + 1].a) &&
          dabs(hits[j].a - hits[j].b - avgDiff) < dabs(LIS[i].a - LIS[i].b - avgDiff))
        LIS[i] = hits[j];
      ++j;
    }
  }
  free(top); free(link);
  return ret;
}
int t4o_lis(const int *pairs, int n, int *out) {
  if (n <= 0) return 0;
  pair_t *lisv = (pair_t *)malloc(sizeof(pair_t) * n);
  int r = lis((const pair_t *)pairs, n, lisv);
  memcpy(out, lisv, sizeof(pair_t) * r);
  free(lisv);
  return r;
}

/* ------------------------------------------------------------------ overlaps */
typedef struct {
  int seqIdx, readStart, readEnd, seqStart, seqEnd, strand, matchCnt, indelCnt;
  double similarity;
  pair_t *coords; int ncoords; int infoFromHits;
} ov_t;
typedef struct { ov_t *o; int n, cap; } ovvec;
static void ov_push(ovvec *v, ov_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 32; v->o = (ov_t *)realloc(v->o, sizeof(ov_t) * v->cap); }
  v->o[v->n++] = x;
}
static void ov_clear(ovvec *v) { for (int i = 0; i < v->n; ++i) free(v->o[i].coords); v->n = 0; }

typedef struct { int a, b, c; } triple_t;
static int triple_lt(const void *x, const void *y) { /* CompSortHitCoordDiff (SeqSet.hpp:241-249) */
  const triple_t *a = (const triple_t *)x, *b = (const triple_t *)y;
  if (a->c != b->c) return a->c < b->c;
  if (a->b != b->b) return a->b < b->b;
  return a->a < b->a;
}
static int pair_lt_b(const void *x, const void *y) { /* CompSortPairBInc (208-214) */
  const pair_t *a = (const pair_t *)x, *b = (const pair_t *)y;
  if (a->b != b->b) return a->b < b->b;
  return a->a < b->a;
}

/* SeqSet.hpp:3330-3367 on a chain of (a = read offset, b = seq offset) */
static int total_hit_len(const pair_t *c, int n, int k, int onSeq) {
  int i, j, ret = 0;
  for (i = 0; i < n;) {
    for (j = i + 1; j < n; ++j) {
      int cur = onSeq ? c[j].b : c[j].a, prv = onSeq ? c[j - 1].b : c[j - 1].a;
      if (cur > prv + k - 1) break;
    }
    ret += (onSeq ? c[j - 1].b - c[i].b : c[j - 1].a - c[i].a) + k;
    i = j;
  }
  return ret;
}

/* SeqSet.hpp:763-1063 (conservativeChain == false, isLongSeqSet == false) */
static int overlaps_from_hits(t4o_set *s, const hit_t *hits, int hitSize, int hitLenRequired, int filter, ovvec *out) {
  int i, j, k, K = s->k;
  int novelMin[2] = {3, 3}, refMin[2] = {3, 3}, removeOnlyRepeats[2] = {0, 0}, possibleOverlapCnt[2] = {0, 0};
  triple_t *diff = (triple_t *)malloc(sizeof(triple_t) * (hitSize + 1));
  pair_t *conc = (pair_t *)malloc(sizeof(pair_t) * (hitSize + 1));
  pair_t *lisv = (pair_t *)malloc(sizeof(pair_t) * (hitSize + 1));
  if (filter == 1) {
    int longestHits[2] = {0, 0};
    for (i = 0; i < hitSize; ++i) { /* note: `i = j` followed by the loop's `++i` (784, 810) */
      int plus = (1 + hits[i].strand) / 2;
      for (j = i + 1; j < hitSize; ++j)
        if (hits[j].strand != hits[i].strand || hits[j].idx != hits[i].idx) break;
      if (!s->seqs[hits[i].idx].isRef) {
        if (j - i > novelMin[plus]) ++possibleOverlapCnt[plus];
        if (j - i > longestHits[plus]) longestHits[plus] = j - i;
      }
      if (!removeOnlyRepeats[plus]) {
        int cnt = 0;
        for (k = i; k < j; ++k) if (hits[k].repeats <= 10000) ++cnt;
        if (cnt >= novelMin[plus]) removeOnlyRepeats[plus] = 1;
      }
      i = j;
    }
    for (i = 0; i <= 1; ++i) {
      if (possibleOverlapCnt[i] > 100000) novelMin[i] = (int)(longestHits[i] * 0.75);
      else if (possibleOverlapCnt[i] > 10000) novelMin[i] = longestHits[i] / 2;
      else if (possibleOverlapCnt[i] > 1000) novelMin[i] = longestHits[i] / 3;
      else if (possibleOverlapCnt[i] > 100) novelMin[i] = longestHits[i] / 4;
    }
  }
  for (i = 0; i < hitSize;) {
    for (j = i + 1; j < hitSize; ++j)
      if (hits[j].strand != hits[i].strand || hits[j].idx != hits[i].idx) break;
    int plus = (1 + hits[i].strand) / 2, isRef = s->seqs[hits[i].idx].isRef;
    int minHit = isRef ? refMin[plus] : novelMin[plus];
    if (j - i < minHit) { i = j; continue; }
    if (removeOnlyRepeats[plus]) {
      int hasUnique = 0;
      for (k = i; k < j; ++k) if (hits[k].repeats <= 10000) { hasUnique = 1; break; }
      if (!hasUnique) { i = j; continue; }
    }
    int g = j - i;
    for (k = i; k < j; ++k) { diff[k - i].a = hits[k].readOffset; diff[k - i].b = hits[k].offset; diff[k - i].c = hits[k].readOffset - hits[k].offset; }
    msort(diff, g, sizeof(triple_t), triple_lt);
    int sIdx, e, adjustRadius = isRef ? s->radius : 0;
    for (sIdx = 0; sIdx < g;) {
      for (e = sIdx + 1; e < g; ++e) {
        int d = diff[e].c - diff[e - 1].c;
        if (d < 0) d = -d;
        if (d > adjustRadius) break;
      }
      if (e - sIdx < minHit || (e - sIdx) * K < hitLenRequired) { sIdx = e; continue; }
      if (removeOnlyRepeats[plus]) { /* quirk (934-940): indexes hits[] with run-relative k */
        int hasUnique = 0;
        for (k = sIdx; k < e; ++k) if (hits[k].repeats <= 10000) { hasUnique = 1; break; }
        if (!hasUnique) { sIdx = e; continue; }
      }
      int n = e - sIdx;
      for (k = sIdx; k < e; ++k) { conc[k - sIdx].a = diff[k].a; conc[k - sIdx].b = diff[k].b; }
      if (adjustRadius > 0) msort(conc, n, sizeof(pair_t), pair_lt_b);
      int lisSize = lis(conc, n, lisv);
      if (lisSize * K < hitLenRequired) { sIdx = e; continue; }
      int hitLen = total_hit_len(lisv, lisSize, K, 0);
      if (hitLen < hitLenRequired) { sIdx = e; continue; }
      if (total_hit_len(lisv, lisSize, K, 1) < hitLenRequired) { sIdx = e; continue; }
      ov_t no; memset(&no, 0, sizeof(no));
      no.seqIdx = hits[i].idx; no.readStart = lisv[0].a; no.readEnd = lisv[lisSize - 1].a + K - 1;
      no.strand = hits[i].strand; no.seqStart = lisv[0].b; no.seqEnd = lisv[lisSize - 1].b + K - 1;
      no.matchCnt = 2 * hitLen; no.similarity = 0; no.indelCnt = 0;
      if (!isRef && hitLen * 2 < no.seqEnd - no.seqStart + 1) { sIdx = e; continue; }
      no.coords = (pair_t *)malloc(sizeof(pair_t) * lisSize); no.ncoords = lisSize;
      memcpy(no.coords, lisv, sizeof(pair_t) * lisSize);
      ov_push(out, no);
      sIdx = e;
    }
    i = j;
  }
  free(diff); free(conc); free(lisv);
  return out->n;
}

/* _overlap::operator< (SeqSet.hpp:104-128) */
static int ov_lt(const void *x, const void *y) {
  const ov_t *a = (const ov_t *)x, *b = (const ov_t *)y;
  if (a->matchCnt != b->matchCnt) return a->matchCnt > b->matchCnt;
  if (a->similarity != b->similarity) return a->similarity > b->similarity;
  if (a->readEnd - a->readStart != b->readEnd - b->readStart) return a->readEnd - a->readStart > b->readEnd - b->readStart;
  if (a->seqIdx != b->seqIdx) return a->seqIdx < b->seqIdx;
  if (a->strand != b->strand) return a->strand < b->strand;
  if (a->readStart != b->readStart) return a->readStart < b->readStart;
  if (a->readEnd != b->readEnd) return a->readEnd < b->readEnd;
  if (a->seqStart != b->seqStart) return a->seqStart < b->seqStart;
  return a->seqEnd < b->seqEnd;
}
/* _sortOverlapOnRef (SeqSet.hpp:139-166) */
static int ov_lt_onref(const void *x, const void *y) {
  const ov_t *a = (const ov_t *)x, *b = (const ov_t *)y;
  if (a->matchCnt != b->matchCnt) return a->matchCnt > b->matchCnt;
  if (a->similarity != b->similarity) return a->similarity > b->similarity;
  if (a->readEnd - a->readStart != b->readEnd - b->readStart) return a->readEnd - a->readStart > b->readEnd - b->readStart;
  if (a->strand != b->strand) return a->strand < b->strand;
  if (a->seqStart != b->seqStart) return a->seqStart < b->seqStart;
  if (a->seqEnd != b->seqEnd) return a->seqEnd < b->seqEnd;
  if (a->readStart != b->readStart) return a->readStart < b->readStart;
  if (a->readEnd != b->readEnd) return a->readEnd < b->readEnd;
  return a->seqIdx < b->seqIdx;
}

/* SeqSet.hpp:1066-1161 */
static int vj_overlaps_from_hits(t4o_set *s, const hit_t *hits, int hitSize, ovvec *out) {
  int i, j;
  hitvec vj = {0, 0, 0};
  for (i = 0; i < hitSize; ++i) {
    const seq_t *q = &s->seqs[hits[i].idx];
    if (!q->isRef) continue;
    if (q->name[3] == 'V' && hits[i].offset >= q->len - 31) hv_push(&vj, hits[i]);
    else if (q->name[3] == 'J' && hits[i].offset < 31) hv_push(&vj, hits[i]);
  }
  overlaps_from_hits(s, vj.h, vj.n, 17, 0, out);
  free(vj.h);
  int cnt = out->n, maxMatch = 0, tagi = 0, tagj = 0;
  for (i = 0; i < cnt; ++i)
    for (j = i + 1; j < cnt; ++j) {
      const char *ni = s->seqs[out->o[i].seqIdx].name, *nj = s->seqs[out->o[j].seqIdx].name;
      if (ni[0] != nj[0] || ni[1] != nj[1] || ni[2] != nj[2] || ni[3] == nj[3]) continue;
      if (ni[3] == 'V') { if (out->o[i].readStart > out->o[j].readStart) continue; }
      else { if (out->o[i].readStart < out->o[j].readStart) continue; }
      if (out->o[i].matchCnt + out->o[j].matchCnt > maxMatch) { maxMatch = out->o[i].matchCnt + out->o[j].matchCnt; tagi = i; tagj = j; }
    }
  if (maxMatch == 0) { ov_clear(out); return 0; }
  ov_t a = out->o[tagi], b = out->o[tagj];
  for (i = 0; i < cnt; ++i) if (i != tagi && i != tagj) free(out->o[i].coords);
  out->o[0] = a; out->o[1] = b; out->n = 2;
  return 2;
}

/* ------------------------------------------------------------------ AlignAlgo */
enum { EDIT_MATCH = 0, EDIT_MISMATCH = 1, EDIT_INSERT = 2, EDIT_DELETE = 3 };
#define SCORE_MATCH 2
#define SCORE_MISMATCH (-2)
#define SCORE_GAPOPEN (-4)
#define SCORE_GAPEXTEND (-1)
#define SCORE_INDEL (-4)

/* AlignAlgo.hpp:49-55 */
static int base_equal_w(const int *w, char c) {
  int sum = w[0] + w[1] + w[2] + w[3];
  if (sum == 0 || c == 'N' || sum < 3 * w[nuc(c) & 3]) return 1;
  return 0;
}
static void reverse_align(signed char *align, int tag) {
  align[tag] = -1;
  for (int i = 0, j = tag - 1; i < j; ++i, --j) { signed char t = align[i]; align[i] = align[j]; align[j] = t; }
}

/* AlignAlgo.hpp:57-216 */
int t4o_global_alignment_posweight(const int *w, int lent, const char *p, int lenp, signed char *align) {
  if (lent == 0 || lenp == 0) { align[0] = -1; return 0; }
  if (lent == 1 && lenp == 1) {
    if (base_equal_w(w, p[0])) { align[0] = EDIT_MATCH; align[1] = -1; return SCORE_MATCH; }
    align[0] = EDIT_MISMATCH; align[1] = -1; return SCORE_MISMATCH;
  }
  int i, j;
  if (lent == lenp) {
    int score = 0;
    for (i = 0; i < lent; ++i) {
      if (base_equal_w(w + 4 * i, p[i])) { align[i] = EDIT_MATCH; score += SCORE_MATCH; }
      else { align[i] = EDIT_MISMATCH; score += SCORE_MISMATCH; }
    }
    align[i] = -1;
    if (score >= lent * SCORE_MATCH + 2 * SCORE_INDEL) return score;
  }
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  int negInf = (lent + 1) * (lenp + 1) * SCORE_INDEL, bmax = lent + 1;
  int *m = (int *)calloc((size_t)(lenp + 1) * (lent + 1), sizeof(int));
  m[0] = 0;
  for (i = 1; i <= lenp; ++i) m[i * bmax] = SCORE_INDEL + i * SCORE_INDEL;
  for (j = 1; j <= lent; ++j) m[j] = SCORE_INDEL + j * SCORE_INDEL;
  for (i = 1; i <= lenp; ++i) {
    int start = (i - leftBand < 1) ? 1 : (i - leftBand), end = (i + rightBand > lent) ? lent : (i + rightBand);
    if (start > 1) m[i * bmax + start - 1] = negInf;
    if (end < lent) m[i * bmax + end + 1] = negInf;
    for (j = start; j <= end; ++j) {
      int score = m[(i - 1) * bmax + j - 1] + (base_equal_w(w + 4 * (j - 1), p[i - 1]) ? SCORE_MATCH : SCORE_MISMATCH);
      score = MAXI(score, m[i * bmax + j - 1] + SCORE_INDEL);
      score = MAXI(score, m[(i - 1) * bmax + j] + SCORE_INDEL);
      m[i * bmax + j] = score;
    }
  }
  int ret = m[lenp * bmax + lent], tagi = lenp, tagj = lent, tag = 0;
  while (tagi > 0 || tagj > 0) {
    int max = m[tagi * bmax + tagj], a = 0;
    if (tagj > 0 && m[tagi * bmax + tagj - 1] + SCORE_INDEL == max) a = EDIT_DELETE;
    if (tagi > 0 && m[(tagi - 1) * bmax + tagj] + SCORE_INDEL == max) a = EDIT_INSERT;
    if (tagj > 0 && tagi > 0) {
      int d = base_equal_w(w + 4 * (tagj - 1), p[tagi - 1]) ? SCORE_MATCH : SCORE_MISMATCH;
      if (m[(tagi - 1) * bmax + tagj - 1] + d == max) a = (d == SCORE_MATCH) ? EDIT_MATCH : EDIT_MISMATCH;
    }
    align[tag++] = (signed char)a;
    if (a == EDIT_DELETE) --tagj; else if (a == EDIT_INSERT) --tagi; else { --tagi; --tagj; }
  }
  reverse_align(align, tag);
  free(m);
  return ret;
}

static int base_equal_c(char t, char p) { return t == p || t == 'N' || p == 'N'; }
/* AlignAlgo.hpp:218-424 (note the stale `i` in the e[0][j] border, line 271) */
int t4o_global_alignment(const char *t, int lent, const char *p, int lenp, signed char *align) {
  if (lent == 0 || lenp == 0) { align[0] = -1; return 0; }
  if (lent == 1 && lenp == 1) {
    if (base_equal_c(t[0], p[0])) { align[0] = EDIT_MATCH; align[1] = -1; return SCORE_MATCH; }
    align[0] = EDIT_MISMATCH; align[1] = -1; return SCORE_MISMATCH;
  }
  int leftBand = 5, rightBand = 5, i, j;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  int negInf = (lent + 1) * (lenp + 1) * SCORE_GAPOPEN, bmax = lent + 1;
  size_t cells = (size_t)(lenp + 1) * (lent + 1);
  int *m = (int *)calloc(cells, sizeof(int)), *e = (int *)calloc(cells, sizeof(int)), *f = (int *)calloc(cells, sizeof(int));
  m[0] = e[0] = f[0] = 0;
  for (i = 1; i <= lenp; ++i) {
    e[i * bmax] = SCORE_GAPOPEN + i * SCORE_GAPEXTEND;
    f[i * bmax] = SCORE_GAPOPEN + i * SCORE_GAPOPEN;
    m[i * bmax] = SCORE_GAPOPEN + i * SCORE_GAPOPEN;
  }
  for (j = 1; j <= lent; ++j) {
    f[j] = SCORE_GAPOPEN + j * SCORE_GAPEXTEND;
    e[j] = SCORE_GAPOPEN + i * SCORE_GAPOPEN; /* i == lenp + 1 here */
    m[j] = SCORE_GAPOPEN + j * SCORE_GAPOPEN;
  }
  for (i = 1; i <= lenp; ++i) {
    int start = (i - leftBand < 1) ? 1 : (i - leftBand), end = (i + rightBand > lent) ? lent : (i + rightBand);
    if (start > 1) { j = start - 1; e[i * bmax + j] = f[i * bmax + j] = m[i * bmax + j] = negInf; }
    if (end < lent) { j = end + 1; e[i * bmax + j] = f[i * bmax + j] = m[i * bmax + j] = negInf; }
    for (j = start; j <= end; ++j) {
      int score = e[(i - 1) * bmax + j] + SCORE_GAPEXTEND;
      score = MAXI(score, m[(i - 1) * bmax + j] + SCORE_GAPOPEN + SCORE_GAPEXTEND);
      e[i * bmax + j] = score;
      score = f[i * bmax + j - 1] + SCORE_GAPEXTEND;
      score = MAXI(score, m[i * bmax + j - 1] + SCORE_GAPOPEN + SCORE_GAPEXTEND);
      f[i * bmax + j] = score;
      score = m[(i - 1) * bmax + j - 1] + (base_equal_c(t[j - 1], p[i - 1]) ? SCORE_MATCH : SCORE_MISMATCH);
      score = MAXI(score, e[i * bmax + j]);
      score = MAXI(score, f[i * bmax + j]);
      m[i * bmax + j] = score;
    }
  }
  int ret = m[lenp * bmax + lent], tagi = lenp, tagj = lent, mat = 0, tag = 0;
  while (tagi > 0 || tagj > 0) {
    if (mat == 0) {
      int max = e[tagi * bmax + tagj], a = EDIT_INSERT;
      if (f[tagi * bmax + tagj] >= max) a = EDIT_DELETE;
      if (tagi > 0 && tagj > 0 &&
          m[(tagi - 1) * bmax + tagj - 1] + (base_equal_c(t[tagj - 1], p[tagi - 1]) ? SCORE_MATCH : SCORE_MISMATCH) == m[tagi * bmax + tagj])
        a = base_equal_c(t[tagj - 1], p[tagi - 1]) ? EDIT_MATCH : EDIT_MISMATCH;
      if (a == EDIT_MATCH || a == EDIT_MISMATCH) { align[tag++] = (signed char)a; --tagi; --tagj; }
      else if (a == EDIT_INSERT) mat = 1;
      else mat = 2;
    } else if (mat == 1) {
      align[tag++] = EDIT_INSERT;
      if (tagi > 0) {
        if (m[(tagi - 1) * bmax + tagj] + SCORE_GAPOPEN + SCORE_GAPEXTEND == e[tagi * bmax + tagj]) { --tagi; mat = 0; }
        else { --tagi; mat = 1; }
      } else mat = 2;
    } else {
      align[tag++] = EDIT_DELETE;
      if (tagj > 0) {
        if (m[tagi * bmax + tagj - 1] + SCORE_GAPOPEN + SCORE_GAPEXTEND == f[tagi * bmax + tagj]) { --tagj; mat = 0; }
        else { --tagj; mat = 2; }
      } else mat = 1;
    }
  }
  reverse_align(align, tag);
  free(m); free(e); free(f);
  return ret;
}

/* AlignAlgo.hpp:1027-1096 */
int t4o_is_mate_overlap(const char *fr, int flen, const char *sr, int slen, int minOverlap, int *offset, int *bestMatchCnt, int checkTandem) {
  int i, j, k, offsetCnt = 0, overlapSize = -1;
  *bestMatchCnt = -1;
  for (j = 0; j < flen - minOverlap; ++j) {
    int matchCnt = 0, flag = 1;
    double thr = 0.95;
    if (flen - j >= 100) thr = 0.85;
    else if (flen - j >= 50) thr = 0.85 + (flen - j - 50) / 50.0 * 0.1;
    for (k = 0; j + k < flen && k < slen; ++k) {
      if (fr[j + k] == sr[k]) ++matchCnt;
      if (matchCnt + (flen - (j + k) - 1) < (int)((flen - j) * thr)) { flag = 0; break; }
    }
    if (flag) { *offset = j; ++offsetCnt; overlapSize = k; *bestMatchCnt = matchCnt; }
  }
  if (offsetCnt != 1) return -1;
  if (checkTandem && overlapSize <= minOverlap * 2) {
    for (i = 1; i <= overlapSize / 2; ++i) {
      int tandem = 1;
      for (j = i; j + i - 1 < overlapSize; j += i) {
        for (k = j; k <= j + i - 1; ++k) if (sr[k - j] != sr[k]) break;
        if (k <= j + i - 1) { tandem = 0; break; }
      }
      if (tandem) return -1;
    }
  }
  return overlapSize;
}

/* IsLowComplexity (main.cpp:183-205) */
static int is_low_complexity(const char *seq) {
  int cnt[5] = {0, 0, 0, 0, 0}, i, low = 0;
  for (i = 0; seq[i]; ++i) { if (seq[i] == 'N') ++cnt[4]; else ++cnt[nuc(seq[i])]; }
  if (cnt[0] >= i / 2 || cnt[1] >= i / 2 || cnt[2] >= i / 2 || cnt[3] >= i / 2 || cnt[4] >= i / 10) return 1;
  for (i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}
/* ProcessRead (main.cpp:224-449) for one pair of ACGTN reads with qualities on both mates or on none (the reference reads the
 * qualities of both without asking, 282, 304, 309). r1 / q1 / r2 / q2: C strings (q1 = q2 = NULL: no qualities). outR / outQ (room
 * for len1 + len2 + 1 each) receive read 1 as ProcessRead leaves it; *flags: 1 read 1 is pushed to the read list (not of low
 * complexity), 2 read 2 is, 4 weight 2 (the merged read is listed twice), 8 read 1 has qualities. Returns the branch taken: 0 the
 * mates stay, 1 read-through (248-287), 2 merged (291-337), 3 one mate stands for both (338-385). The k-mer counting of the
 * surviving reads and the bookkeeping of ids are the caller's. PARITY: pinned against the function itself -- the reference's main.cpp
 * compiled as a library with its `main` renamed (oracle/ref_main_probe.cpp -> _ref/libt4refmain.so;
 * tests/test_oracle_vs_ref.py::test_process_read_vs_the_reference_main) -- and through the reference binary (whole stage 1). */
int t4o_process_read(const char *r1in, const char *q1in, const char *r2in, const char *q2in, char *outR, char *outQ, int *flags) {
  int slen = (int)strlen(r1in), flen = (int)strlen(r2in), j, k, kind = 0, rWeight = 1, r2Alive = 1;
  char *r1 = strdup(r1in), *q1 = q1in ? strdup(q1in) : NULL;
  char *r2 = (char *)malloc(flen + 1), *q2 = q2in ? strdup(q2in) : NULL;
  reverse_complement(r2, r2in, flen);                     /* 234 */
  if (q2) for (j = 0, k = flen - 1; j < k; ++j, --k) { char t = q2[j]; q2[j] = q2[k]; q2[k] = t; }
  int minOverlap = (flen + slen) / 10, minOverlap2 = (flen + slen) / 20, offset = -1, best = -1;
  if (minOverlap > 31) minOverlap = 31;
  if (minOverlap2 > 31) minOverlap2 = 31;
  int hasQ1 = q1 != NULL;
  int ov = t4o_is_mate_overlap(r2, flen, r1, slen, minOverlap, &offset, &best, 0);
  if (ov >= 0) {
    kind = 1;
    r1[ov] = 0;
    if (q1) {
      q1[ov] = 0;
      for (j = 0; j < ov; ++j) if (q2[j + offset] > q1[j] || r1[j] == 'N') { r1[j] = r2[j + offset]; q1[j] = q2[j + offset]; }
    }
    r2Alive = 0;
  } else if ((ov = t4o_is_mate_overlap(r1, slen, r2, flen, minOverlap2, &offset, &best, 1)) >= 0) {
    if (best >= 0.95 * ov) {
      kind = 2;
      char *r = (char *)calloc(slen + flen + 1, 1), *q = (char *)calloc(slen + flen + 1, 1);
      for (j = 0; j < flen; ++j) { r[offset + j] = r2[j]; q[offset + j] = q2 ? q2[j] : 0; }
      int len = offset + j;
      for (j = 0; j < slen && j < len; ++j)
        if (j < offset || (q1 ? q1[j] : 0) >= q[j] - 14 || r[j] == 'N') { r[j] = r1[j]; q[j] = q1 ? q1[j] : 0; }
      r[len] = q[len] = 0;
      free(r1); free(q1);
      r1 = r; q1 = q;                                      /* (q stands for "no qualities" when the input had none: hasQ1 says) */
      r2Alive = 0; ++rWeight;
    } else {
      kind = 3;
      int useFirst = 1;
      if (q1) {
        double a = 0, b = 0;
        for (j = offset; j < slen; ++j) a += q1[j] - 32;
        for (j = flen - 1; j >= flen - ov; --j) b += q2[j] - 32;
        a /= ov; b /= ov;
        if (a + 10 < b) useFirst = 0;
      }
      if (!useFirst) {
        free(r1); free(q1);
        r1 = (char *)malloc(flen + 1);
        reverse_complement(r1, r2, flen);                 /* 366-367; the qualities stay reversed (368-377) */
        q1 = q2 ? strdup(q2) : NULL;
        hasQ1 = q2 != NULL;
      }
      r2Alive = 0;
    }
  }
  int len1 = (int)strlen(r1);
  memcpy(outR, r1, len1 + 1);
  if (q1) memcpy(outQ, q1, len1 + 1); else memset(outQ, 0, len1 + 1);
  *flags = (hasQ1 ? 8 : 0) | (rWeight == 2 ? 4 : 0);
  if (!is_low_complexity(r1)) *flags |= 1;
  if (r2Alive && !is_low_complexity(r2in)) *flags |= 2;   /* read 2 is reverse-complemented back (388-399): the same letters */
  free(r1); free(q1); free(r2); free(q2);
  return kind;
}

/* SeqSet::HasHitInSet (SeqSet.hpp:3144-3327): hits bucketed by (strand, sequence) in emission order, the bucket with the most
 * distinct read offsets per strand, GetOverlapsFromHits (filter 1) on the chosen bucket(s). Returns -1 / 0 / 1. */
int t4o_has_hit_in_set(t4o_set *s, const char *read, int mode) {
  int len = (int)strlen(read), i, j, k;
  if (len < s->k) return 0;
  char *rc = (char *)malloc(len + 1);
  hitvec hits = {0, 0, 0};
  get_hits(s, read, rc, len, 0, -1, 0, &hits);
  free(rc);
  if (hits.n == 0) { free(hits.h); return 0; }
  const int seqCnt = s->nseq;
  /* bucket sort, stable: order[] lists the hits of bucket (tag, idx) consecutively */
  int *cnt = (int *)calloc((size_t)2 * seqCnt + 1, sizeof(int));
  for (i = 0; i < hits.n; ++i) ++cnt[(hits.h[i].strand == 1 ? seqCnt : 0) + hits.h[i].idx + 1];
  for (i = 0; i < 2 * seqCnt; ++i) cnt[i + 1] += cnt[i];
  hit_t *bk = (hit_t *)malloc(sizeof(hit_t) * hits.n);
  int *fill = (int *)malloc(sizeof(int) * 2 * seqCnt);
  memcpy(fill, cnt, sizeof(int) * 2 * seqCnt);
  for (i = 0; i < hits.n; ++i) bk[fill[(hits.h[i].strand == 1 ? seqCnt : 0) + hits.h[i].idx]++] = hits.h[i];
  int max[2] = {-1, -1}, maxSeqIdx[2] = {-1, -1}, maxTag = -1;
  for (k = 0; k <= 1; ++k)
    for (i = 0; i < seqCnt; ++i) {
      const hit_t *b = bk + cnt[k * seqCnt + i];
      int size = cnt[k * seqCnt + i + 1] - cnt[k * seqCnt + i], readHitCount = 1;
      for (j = 1; j < size; ++j) if (b[j].readOffset != b[j - 1].readOffset) ++readHitCount;
      if (size > 0 && readHitCount > max[k]) { maxSeqIdx[k] = i; max[k] = readHitCount; }
    }
  ovvec ov = {0, 0, 0};
  const int K = s->k, hlr = s->hitLenRequired;
  const int both = (max[0] + K - 1 >= hlr && max[1] + K - 1 >= hlr);
  if (mode == 1 && both) {
    maxTag = 1;
    int maxMatchCnt = 0;
    for (k = 0; k <= 1; ++k)
      for (i = 0; i < seqCnt; ++i) {
        const hit_t *b = bk + cnt[k * seqCnt + i];
        int size = cnt[k * seqCnt + i + 1] - cnt[k * seqCnt + i], readHitCount = 1, l;
        for (j = 1; j < size; ++j) if (b[j].readOffset != b[j - 1].readOffset) ++readHitCount;
        if (readHitCount + K - 1 < hlr) continue;
        ovvec tmp = {0, 0, 0};
        overlaps_from_hits(s, b, size, hlr, 1, &tmp);
        int taken = 0;
        for (l = 0; l < tmp.n; ++l)
          if (tmp.o[l].matchCnt > maxMatchCnt) {
            ov_clear(&ov); free(ov.o);
            ov = tmp; maxMatchCnt = ov.o[l].matchCnt; maxTag = ov.o[l].strand == 1 ? 1 : 0; taken = 1;
            break;
          }
        if (!taken) { ov_clear(&tmp); free(tmp.o); }
      }
  } else if (both) {
    ovvec t0 = {0, 0, 0}, t1 = {0, 0, 0};
    overlaps_from_hits(s, bk + cnt[maxSeqIdx[0]], cnt[maxSeqIdx[0] + 1] - cnt[maxSeqIdx[0]], hlr, 1, &t0);
    overlaps_from_hits(s, bk + cnt[seqCnt + maxSeqIdx[1]], cnt[seqCnt + maxSeqIdx[1] + 1] - cnt[seqCnt + maxSeqIdx[1]], hlr, 1, &t1);
    if (t0.n > 0 && t1.n > 0) maxTag = t0.o[0].matchCnt >= t1.o[0].matchCnt ? 0 : 1;
    else if (t0.n > 0) maxTag = 0;
    else maxTag = 1;
    if (maxTag == 0) { ov = t0; ov_clear(&t1); free(t1.o); } else { ov = t1; ov_clear(&t0); free(t0.o); }
  } else {
    maxTag = max[1] >= max[0] ? 1 : 0;
    int at = maxTag * seqCnt + maxSeqIdx[maxTag];
    overlaps_from_hits(s, bk + cnt[at], cnt[at + 1] - cnt[at], hlr, 1, &ov);
  }
  int n = ov.n;
  ov_clear(&ov); free(ov.o); free(cnt); free(fill); free(bk); free(hits.h);
  if (n == 0) return 0;
  return maxTag == 0 ? -1 : 1;
}

/* SeqSet.hpp:570-587 */
static void align_stats(const signed char *align, int update, int *m, int *mm, int *indel) {
  if (!update) *m = *mm = *indel = 0;
  for (int k = 0; align[k] != -1; ++k) {
    if (align[k] == EDIT_MATCH) ++*m; else if (align[k] == EDIT_MISMATCH) ++*mm; else ++*indel;
  }
}

/* SeqSet.hpp:590-617 */
static int overlap_low_complex(const char *r, const ov_t *o) {
  int cnt[4] = {0, 0, 0, 0}, i, lowCnt = 0, lowTotal = 0;
  for (i = o->readStart; i <= o->readEnd; ++i) { if (r[i] == 'N') continue; ++cnt[nuc(r[i]) & 3]; }
  for (i = 0; i < 4; ++i) if (cnt[i] <= 2) { ++lowCnt; lowTotal += cnt[i]; }
  if (lowTotal * 7 >= o->readEnd - o->readStart + 1) return 0;
  return lowCnt >= 2;
}

static int64_t g_hit_counter = 0; /* H_r accounting for t4o_annotate_batch */

/* SeqSet.hpp:1508-2124 (readType 0, puse NULL, isLongSeqSet false) */
static int overlaps_from_read(t4o_set *s, const char *read, int strand, int barcode, int skipRepeats, ovvec *ov) {
  int i, j, k, K = s->k, len = (int)strlen(read);
  if (len < K) return -1;
  int overlapCnt = 0;
  hitvec hits = {0, 0, 0};
  char *rc = (char *)malloc(len + 1);
  if (skipRepeats) {
    get_hits(s, read, rc, len, strand, barcode, 1, &hits);
    g_hit_counter += hits.n;
    sort_hits(s, &hits);
    overlapCnt = overlaps_from_hits(s, hits.h, hits.n, s->hitLenRequired, 0, ov);
    if (overlapCnt == 0) { hits.n = 0; ov_clear(ov); }
  }
  if (overlapCnt == 0) {
    get_hits(s, read, rc, len, strand, barcode, 0, &hits);
    g_hit_counter += hits.n;
    sort_hits(s, &hits);
    overlapCnt = overlaps_from_hits(s, hits.h, hits.n, s->hitLenRequired, 1, ov);
  }
  if (overlapCnt == 0) {
    overlapCnt = vj_overlaps_from_hits(s, hits.h, hits.n, ov);
    if (overlapCnt == 0) { free(hits.h); free(rc); return 0; }
  }
  free(hits.h);
  msort(ov->o, ov->n, sizeof(ov_t), ov_lt);
  k = 1;
  for (i = 1; i < overlapCnt; ++i) {
    if (ov->o[i].strand != ov->o[0].strand) { free(ov->o[i].coords); ov->o[i].coords = NULL; continue; }
    if (i != k) ov->o[k] = ov->o[i];
    ++k;
  }
  ov->n = k; overlapCnt = k;
  reverse_complement(rc, read, len);

  int bestNovel = -1;
  int *rep = (int *)malloc(sizeof(int) * (overlapCnt + 1)), nrep = 0;
  ov_t *o = ov->o;
  for (i = 0; i < overlapCnt; ++i) {
    const char *r = o[i].strand == 1 ? read : rc;
    const seq_t *sq = &s->seqs[o[i].seqIdx];
    o[i].infoFromHits = i;
    pair_t *hc = o[i].coords; int hitCnt = o[i].ncoords;
    int matchCnt = 0, mismatchCnt = 0, indelCnt = 0; double similarity = 1;
    if (sq->isRef) { /* firstRef bookkeeping only */ }
    else if (bestNovel != -1 && overlapCnt > 50) {
      const ov_t *bn = &o[bestNovel];
      if (bn->readStart == 0 && bn->readEnd == len - 1) {
        if (bn->similarity == 1) { o[i].similarity = 0; continue; }
        else if (bn->similarity > s->repeatSim && o[i].matchCnt < 0.9 * bn->matchCnt) { o[i].similarity = 0; continue; }
      }
      if (bn->readStart + len - 1 - bn->readEnd < s->radius) {
        if (bn->similarity == 1 && o[i].matchCnt < 0.9 * bn->matchCnt) { o[i].similarity = 0; continue; }
        else if (bn->similarity > s->repeatSim && o[i].matchCnt < 0.8 * bn->matchCnt) { o[i].similarity = 0; continue; }
      }
      if (o[i].seqStart - o[i].readStart >= s->radius && o[i].seqEnd + (len - 1 - o[i].readEnd) + s->radius < sq->len &&
          bn->matchCnt > 0.97 * (2 * len) && bn->similarity > s->repeatSim && o[i].matchCnt < 0.9 * bn->matchCnt) { o[i].similarity = 0; continue; }
      if (o[i].matchCnt < 0.4 * bn->matchCnt) { o[i].similarity = 0; continue; }
      if (overlapCnt > 1000 && o[i].matchCnt < 0.9 * bn->matchCnt) { o[i].similarity = 0; continue; }
    }
    matchCnt += 2 * K;
    signed char *align = (signed char *)malloc(o[i].readEnd - o[i].readStart + 1 + o[i].seqEnd - o[i].seqStart + 1 + 8);
    for (j = 1; j < hitCnt; ++j) {
      int pa = hc[j - 1].a, pb = hc[j - 1].b, qa = hc[j].a, qb = hc[j].b;
      int doDP = 0;
      if (pb - pa == qb - qa) {
        if (pa + K - 1 >= qa) matchCnt += 2 * (qa - pa);
        else { matchCnt += 2 * K; doDP = 1; }
      } else {
        if (s->radius == 0 || !sq->isRef) { similarity = 0; break; }
        if (pa + K - 1 >= qa && pb + K - 1 < qb) { matchCnt += 2 * (qa - pa); indelCnt += (qb - (pb + K) + (qa + K - pa)); }
        else if (pa + K - 1 < qa && pb + K - 1 >= qb) { matchCnt += 2 * (qb - pb); indelCnt += (qa - (pa + K) + (qb + K - pb)); }
        else if (pa + K - 1 >= qa && pb + K - 1 >= qb) { matchCnt += 2 * MINI(qa - pa, qb - pb); indelCnt += ABSI((qa - qb) - (pa - pb)); }
        else { matchCnt += 2 * K; doDP = 2; }
      }
      if (doDP) {
        if (qb - (pb + K) > s->nomatchGapLimit || qa - (pa + K) > s->nomatchGapLimit) { similarity = 0; break; }
        if (sq->isRef) t4o_global_alignment(sq->cons + pb + K, qb - (pb + K), r + pa + K, qa - (pa + K), align);
        else t4o_global_alignment_posweight(sq->pw + 4 * (pb + K), qb - (pb + K), r + pa + K, qa - (pa + K), align);
        int c0, c1, c2;
        align_stats(align, 0, &c0, &c1, &c2);
        matchCnt += 2 * c0; mismatchCnt += c1; indelCnt += c2;
        if (doDP == 1) { if ((s->radius == 0 || !sq->isRef) && indelCnt > 0) { similarity = 0; break; } }
        else { if (!sq->isRef && indelCnt > 0) { similarity = 0; break; } }
      }
    }
    free(align);
    (void)mismatchCnt;
    o[i].matchCnt = matchCnt; o[i].indelCnt = indelCnt;
    if (similarity == 1) o[i].similarity = (double)matchCnt / (o[i].seqEnd - o[i].seqStart + 1 + o[i].readEnd - o[i].readStart + 1);
    else o[i].similarity = 0;
    if (overlap_low_complex(r, &o[i])) o[i].similarity = 0;
    if (!sq->isRef && o[i].similarity > 0) { if (bestNovel == -1 || ov_lt(&o[i], &o[bestNovel])) bestNovel = i; }
    if (o[i].similarity > 0) {
      for (j = 0; j < nrep; ++j) { int kk = rep[j]; if (o[i].readStart >= o[kk].readStart && o[i].readEnd <= o[kk].readEnd) break; }
      if (j >= nrep) rep[nrep++] = i;
    }
  }
  free(rep); free(rc);
  for (i = 0; i < overlapCnt; ++i) { free(o[i].coords); o[i].coords = NULL; o[i].ncoords = 0; }
  k = 0;
  for (i = 0; i < overlapCnt; ++i) {
    int isRef = s->seqs[o[i].seqIdx].isRef;
    if (isRef && o[i].similarity < s->refSim) continue;
    if (!isRef && o[i].similarity < s->novelSim) continue;
    o[k++] = o[i];
  }
  ov->n = k;
  return k;
}

static void ov_export(t4o_overlap *d, const ov_t *v) {
  d->seqIdx = v->seqIdx; d->readStart = v->readStart; d->readEnd = v->readEnd; d->seqStart = v->seqStart;
  d->seqEnd = v->seqEnd; d->strand = v->strand; d->matchCnt = v->matchCnt; d->indelCnt = v->indelCnt; d->similarity = v->similarity;
}

int t4o_hits(t4o_set *s, const char *read, int strand, int barcode, int allowTotalSkip, int doSort, int *out5, int cap) {
  int len = (int)strlen(read);
  char *rc = (char *)malloc(len + 1);
  hitvec hits = {0, 0, 0};
  if (len >= 1) get_hits(s, read, rc, len, strand, barcode, allowTotalSkip, &hits);
  if (doSort) sort_hits(s, &hits);
  for (int i = 0; i < hits.n && i < cap; ++i) {
    out5[5 * i] = hits.h[i].idx; out5[5 * i + 1] = hits.h[i].offset; out5[5 * i + 2] = hits.h[i].readOffset;
    out5[5 * i + 3] = hits.h[i].strand; out5[5 * i + 4] = hits.h[i].repeats;
  }
  int n = hits.n;
  free(hits.h); free(rc);
  return n;
}

int t4o_overlaps_from_hits(t4o_set *s, const char *read, int strand, int barcode, int allowTotalSkip, int hitLenRequired,
                           int filter, t4o_overlap *out, int cap, int *chainOff, int *coords, int coordCap) {
  int len = (int)strlen(read), c = 0;
  char *rc = (char *)malloc(len + 1);
  hitvec hits = {0, 0, 0};
  ovvec ov = {0, 0, 0};
  get_hits(s, read, rc, len, strand, barcode, allowTotalSkip, &hits);
  sort_hits(s, &hits);
  overlaps_from_hits(s, hits.h, hits.n, hitLenRequired, filter, &ov);
  for (int i = 0; i < ov.n; ++i) {
    if (i < cap) { ov_export(out + i, &ov.o[i]); chainOff[i] = c; }
    for (int k = 0; k < ov.o[i].ncoords; ++k) {
      if (c < coordCap) { coords[2 * c] = ov.o[i].coords[k].a; coords[2 * c + 1] = ov.o[i].coords[k].b; }
      ++c;
    }
  }
  if (ov.n < cap) chainOff[ov.n] = c;
  int n = ov.n;
  ov_clear(&ov); free(ov.o); free(hits.h); free(rc);
  return n;
}

int t4o_overlaps_from_read(t4o_set *s, const char *read, int strand, int barcode, int readType, int skipRepeats, t4o_overlap *out, int cap) {
  if (readType != 0) return -2;
  ovvec ov = {0, 0, 0};
  int ret = overlaps_from_read(s, read, strand, barcode, skipRepeats, &ov);
  for (int i = 0; i < ov.n && i < cap; ++i) ov_export(out + i, &ov.o[i]);
  ov_clear(&ov); free(ov.o);
  return ret;
}

/* SeqSet.hpp:5289-5321 */
static int contig_intervals(const char *read, int gapN, pair_t *contigs) {
  int i, j, n = 0;
  for (i = 0; read[i];) {
    int NCnt = 0;
    for (j = i + 1; read[j]; ++j) {
      if (j >= i + gapN && read[j - gapN] == 'N') --NCnt;
      if (read[j] == 'N') ++NCnt;
      if (NCnt >= gapN) break;
    }
    contigs[n].a = i;
    contigs[n].b = read[j] ? j - gapN : j - 1;
    ++n;
    if (!read[j]) break;
    i = j + 1;
  }
  return n;
}

/* AnnotateRead with detailLevel 0 (SeqSet.hpp:6016-6066, 6167-6321) */
static int annotate_read0(t4o_set *s, const char *read, ov_t g[4]) {
  int i, j, k, len = (int)strlen(read);
  for (i = 0; i < 4; ++i) { memset(&g[i], 0, sizeof(ov_t)); g[i].seqIdx = -1; g[i].readStart = g[i].readEnd = g[i].seqStart = g[i].seqEnd = -1; g[i].strand = 1; }
  pair_t *contigs = (pair_t *)malloc(sizeof(pair_t) * (len + 2));
  int contigCnt = contig_intervals(read, s->gapN, contigs);
  char *buf = (char *)malloc(len + 1);
  ovvec all = {0, 0, 0};
  for (k = 0; k < contigCnt; ++k) {
    int cl = contigs[k].b - contigs[k].a + 1;
    memcpy(buf, read + contigs[k].a, cl); buf[cl] = 0;
    ovvec ov = {0, 0, 0};
    overlaps_from_read(s, buf, 0, -1, 0, &ov);
    for (i = 0; i < ov.n; ++i) { ov.o[i].readStart += contigs[k].a; ov.o[i].readEnd += contigs[k].a; }
    msort(ov.o, ov.n, sizeof(ov_t), ov_lt);
    for (i = 0; i < ov.n; ++i) ov_push(&all, ov.o[i]);
    free(ov.o);
  }
  free(buf); free(contigs);
  msort(all.o, all.n, sizeof(ov_t), ov_lt);
  int *seqUsed = (int *)malloc(sizeof(int) * (s->nseq + 1));
  for (i = 0; i < s->nseq; ++i) seqUsed[i] = -1;
  ov_t *o = all.o; int overlapCnt = all.n;
  const double geneSim = 0.8;
  k = 0;
  for (i = 0; i < overlapCnt; ++i) {
    int gt = gene_type(s->seqs[o[i].seqIdx].name);
    if (gt < 0 || gt == 1) continue;
    if (seqUsed[o[i].seqIdx] == -1 && o[i].similarity >= geneSim) { seqUsed[o[i].seqIdx] = k; o[k] = o[i]; ++k; }
    else if (seqUsed[o[i].seqIdx] != -1 && gt == 2) {
      ov_t *base = &o[seqUsed[o[i].seqIdx]];
      if (o[i].matchCnt == base->matchCnt && o[i].similarity == base->similarity) {
        for (j = 0; j < k; ++j) if (gene_type(s->seqs[o[j].seqIdx].name) == 3) break;
        if (j < k && o[i].readEnd <= o[j].readStart + 3) {
          if (base->readEnd > o[j].readStart + 3 || ABSI(o[i].readEnd - o[j].readStart) < ABSI(base->readEnd - o[j].readStart)) *base = o[i];
        }
      }
    }
  }
  overlapCnt = k;
  free(seqUsed);
  if (overlapCnt == 0) { free(all.o); return 0; }
  char BT = 0, chain = 0;
  for (i = 0; i < overlapCnt; ++i) {
    const char *name = s->seqs[o[i].seqIdx].name;
    if (BT && name[0] != BT) continue;
    BT = name[0];
    if (chain && !(name[2] == chain || (name[2] == 'D' && chain == 'A') || (name[2] == 'A' && chain == 'D'))) continue;
    chain = name[2];
    int gt = gene_type(name);
    if (gt >= 0 && g[gt].seqIdx == -1) g[gt] = o[i];
  }
  if (g[3].seqIdx != -1 && g[3].readEnd - g[3].readStart + 1 <= len / 2 && g[3].readEnd - g[3].readStart + 1 <= 50) {
    for (i = 0; i < 3; ++i)
      if (g[i].seqIdx >= 0 && (g[i].readEnd - 17 > g[3].readStart || g[3].readEnd < g[i].readEnd) && g[3].seqStart >= 100) { g[3].seqIdx = -1; break; }
  }
  free(all.o);
  return 1;
}

int t4o_annotate_read0(t4o_set *s, const char *read, t4o_overlap out[4]) {
  ov_t g[4];
  int r = annotate_read0(s, read, g);
  for (int i = 0; i < 4; ++i) ov_export(out + i, &g[i]);
  return r;
}

int64_t t4o_annotate_batch(t4o_set *s, const char *reads, int stride, int64_t n, t4o_overlap *out4, int64_t *hitsPerRead) {
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) {
    g_hit_counter = 0;
    t4o_annotate_read0(s, reads + i * stride, out4 + 4 * i);
    if (hitsPerRead) hitsPerRead[i] = g_hit_counter;
    total += g_hit_counter;
  }
  return total;
}

/* SeqSet.hpp:1165-1277 */
static int extend_overlap(t4o_set *s, const char *r, int len, const seq_t *seq, double mmFactor, signed char *align, const ov_t *ov, ov_t *ext) {
  int matchCnt, mismatchCnt, indelCnt, i, kk, ret = 1;
  int left = MINI(ov->readStart, ov->seqStart), goodLeft = 0;
  t4o_global_alignment_posweight(seq->pw + 4 * (ov->seqStart - left), left, r + ov->readStart - left, left, align);
  align_stats(align, 0, &matchCnt, &mismatchCnt, &indelCnt);
  if (indelCnt > 0) { left = 0; ret = 0; }
  for (i = 0; align[i] != -1; ++i) ;
  int tmpMatch = 0;
  for (i = i - 1, kk = 1; i >= 0; --i, ++kk) {
    if (align[i] == EDIT_MATCH) { ++tmpMatch; if (tmpMatch > 0.75 * kk) goodLeft = kk; }
    else if (align[i] != EDIT_MISMATCH) break;
  }
  int right = MINI(len - 1 - ov->readEnd, seq->len - 1 - ov->seqEnd), goodRight = 0;
  t4o_global_alignment_posweight(seq->pw + 4 * (ov->seqEnd + 1), right, r + ov->readEnd + 1, right, align);
  int oldIndel = indelCnt;
  align_stats(align, 1, &matchCnt, &mismatchCnt, &indelCnt);
  if (indelCnt > oldIndel) { right = 0; ret = 0; }
  tmpMatch = 0;
  for (i = 0; align[i] != -1; ++i) {
    if (align[i] == EDIT_MATCH) { ++tmpMatch; if (tmpMatch > 0.75 * (i + 1)) goodRight = i + 1; }
    else if (align[i] != EDIT_MISMATCH) break;
  }
  int mmThr = 2;
  if (left >= 2) ++mmThr;
  if (right >= 2) ++mmThr;
  double density = 1.5 / s->k;
  mmThr = (int)(mmThr * mmFactor);
  if (mismatchCnt > mmThr && (double)mismatchCnt / (left + right) > density) ret = 0;
  /* only these fields are assigned (1243-1251); everything else of *ext keeps its previous value */
  ext->seqIdx = ov->seqIdx;
  ext->readStart = ov->readStart - left; ext->readEnd = ov->readEnd + right;
  ext->seqStart = ov->seqStart - left; ext->seqEnd = ov->seqEnd + right;
  ext->strand = ov->strand;
  ext->matchCnt = 2 * matchCnt + ov->matchCnt;
  ext->similarity = (double)(2 * matchCnt + ov->matchCnt) / (ext->readEnd - ext->readStart + 1 + ext->seqEnd - ext->seqStart + 1);
  if ((seq->isRef && ext->similarity < s->refSim) || (!seq->isRef && ext->similarity < s->novelSim)) {
    *ext = *ov; ext->coords = NULL; ext->ncoords = 0; ret = 0;
  }
  if (ret == 0) {
    ext->readStart = ov->readStart - goodLeft; ext->readEnd = ov->readEnd + goodRight;
    ext->seqStart = ov->seqStart - goodLeft; ext->seqEnd = ov->seqEnd + goodRight;
  }
  return ret;
}

int t4o_extend_overlap(t4o_set *s, const char *read, double mmFactor, const t4o_overlap *in, t4o_overlap *out) {
  int len = (int)strlen(read);
  ov_t a, b; memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
  b.seqIdx = -1; b.readStart = b.readEnd = b.seqStart = b.seqEnd = -1; b.strand = 1;
  a.seqIdx = in->seqIdx; a.readStart = in->readStart; a.readEnd = in->readEnd; a.seqStart = in->seqStart; a.seqEnd = in->seqEnd;
  a.strand = in->strand; a.matchCnt = in->matchCnt; a.indelCnt = in->indelCnt; a.similarity = in->similarity;
  signed char *align = (signed char *)malloc(2 * len + 8 + 2 * s->seqs[in->seqIdx].len);
  int ret = extend_overlap(s, read, len, &s->seqs[in->seqIdx], mmFactor, align, &a, &b);
  ov_export(out, &b);
  free(align);
  return ret;
}

/* SeqSet.hpp:4632-4701 */
int t4o_assign_read(t4o_set *s, const char *read, int strand, int barcode, t4o_overlap *out) {
  ovvec ov = {0, 0, 0};
  ov_t ext; memset(&ext, 0, sizeof ext); ext.seqIdx = -1; ext.readStart = ext.readEnd = ext.seqStart = ext.seqEnd = -1; ext.strand = 1;
  int cnt = overlaps_from_read(s, read, strand, barcode, 0, &ov), i;
  ov_t none = ext;
  if (cnt <= 0 || s->nseq == 0) { ov_export(out, &none); free(ov.o); return -1; }
  msort(ov.o, ov.n, sizeof(ov_t), s->seqs[0].isRef ? ov_lt_onref : ov_lt);
  int len = (int)strlen(read);
  char *rc = (char *)malloc(len + 1);
  reverse_complement(rc, read, len);
  const char *r = ov.o[0].strand == -1 ? rc : read;
  signed char *align = (signed char *)malloc(2 * len + 8);
  for (i = 0; i < cnt; ++i) {
    if (extend_overlap(s, r, len, &s->seqs[ov.o[i].seqIdx], barcode == -1 ? 1.0 : 2.0, align, &ov.o[i], &ext) == 1)
      if (ext.readStart == 0 && ext.readEnd == len - 1) break;
  }
  free(rc); free(align);
  int ret = -1;
  if (i < cnt) { ov_export(out, &ext); ret = ext.seqIdx; }
  else ov_export(out, &none);
  free(ov.o);
  return ret;
}

/* SeqSet::RecomputePosWeight (SeqSet.hpp:4705-4738) with UpdatePosWeightFromRead (2466-2474): the weights of every contig
 * from the reads assigned to it. reads[i] is added on strand assign[i].strand from column assign[i].seqStart on; entries with
 * seqIdx == -1 are skipped; columns left at zero get count 1 on their consensus base unless it is 'N'. */
void t4o_recompute_posweight(t4o_set *s, int n, const char *const *reads, const t4o_overlap *assign) {
  int i, j;
  for (i = 0; i < s->nseq; ++i) if (s->seqs[i].pw) memset(s->seqs[i].pw, 0, sizeof(int) * 4 * (size_t)s->seqs[i].len);
  for (i = 0; i < n; ++i) {
    if (assign[i].seqIdx == -1) continue;
    seq_t *q = &s->seqs[assign[i].seqIdx];
    int len = (int)strlen(reads[i]);
    char *r = (char *)malloc(len + 1);
    if (assign[i].strand == 1) memcpy(r, reads[i], len + 1); else reverse_complement(r, reads[i], len);
    for (j = 0; j < len; ++j)
      if (r[j] != 'N') ++q->pw[4 * (j + assign[i].seqStart) + nuc(r[j])];
    free(r);
  }
  for (i = 0; i < s->nseq; ++i) {
    seq_t *q = &s->seqs[i];
    if (!q->pw) continue;
    for (j = 0; j < q->len; ++j)
      if (q->cons[j] != 'N' && q->pw[4 * j] + q->pw[4 * j + 1] + q->pw[4 * j + 2] + q->pw[4 * j + 3] == 0) q->pw[4 * j + nuc(q->cons[j])] = 1;
  }
}
/* SeqSet::UpdateConsensus (SeqSet.hpp:4537-4588) of every contig, consensus characters only (UpdateAllConsensus, 4525-4535): per
 * column the first base with the largest count replaces the consensus base when that one's count is strictly smaller; columns
 * without counts stay; 'N' reads as base 0 (nucToNum, main.cpp:39-44). The k-mer index of the set is NOT rebuilt here (the
 * reference removes and re-inserts the contig's k-mers): after this call the set is good for reading consensus strings only.
 * Returns the number of bases changed. */
int t4o_update_all_consensus_chars(t4o_set *s) {
  int i, j, t, changed = 0;
  for (i = 0; i < s->nseq; ++i) {
    seq_t *q = &s->seqs[i];
    if (!q->pw || q->isRef) continue;
    for (j = 0; j < q->len; ++j) {
      int mx = 0, tag = 0;
      for (t = 0; t < 4; ++t) if (q->pw[4 * j + t] > mx) { mx = q->pw[4 * j + t]; tag = t; }
      if (mx == 0) continue;
      int cur = q->cons[j] == 'N' ? 0 : nuc(q->cons[j]);
      if (cur != tag && q->pw[4 * j + cur] < mx) { q->cons[j] = "ACGT"[tag]; ++changed; }
    }
  }
  return changed;
}
int t4o_kmer_length(t4o_set *s) { return s->k; }
void t4o_seq_posweight(t4o_set *s, int i, int *out) { if (s->seqs[i].pw) memcpy(out, s->seqs[i].pw, sizeof(int) * 4 * (size_t)s->seqs[i].len); }

/* ------------------------------------------------------------------------------------------------
 * KmerCount (KmerCount.hpp): counts of canonical k-mers (AddCount, 64-97) and the per-read count statistics
 * with quality trimming (GetCountStatsAndTrim, 177-288). TEST INFRASTRUCTURE like the rest of this file.
 * The reference keeps std::map shards; the counts are a function of the multiset of k-mers only, so an
 * open-addressing table does here.
 * ------------------------------------------------------------------------------------------------ */
struct t4o_kc { int k; uint64_t cap, used; uint64_t *keys; int *cnt; int maxReadLen; };

t4o_kc *t4o_kc_new(int k) {
  t4o_kc *c = (t4o_kc *)calloc(1, sizeof *c);
  c->k = k; c->cap = 1u << 16; c->maxReadLen = -1;
  c->keys = (uint64_t *)calloc(c->cap, sizeof(uint64_t)); c->cnt = (int *)calloc(c->cap, sizeof(int));
  return c;
}
void t4o_kc_free(t4o_kc *c) { if (c) { free(c->keys); free(c->cnt); free(c); } }
static uint64_t kc_mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static int *kc_slot(t4o_kc *c, uint64_t kc, int create) {   /* keys hold code + 1 (0 = empty) */
  for (;;) {
    uint64_t h = kc_mix(kc) & (c->cap - 1);
    while (c->keys[h] && c->keys[h] != kc + 1) h = (h + 1) & (c->cap - 1);
    if (c->keys[h]) return &c->cnt[h];
    if (!create) return NULL;
    if (2 * (c->used + 1) <= c->cap) { c->keys[h] = kc + 1; ++c->used; return &c->cnt[h]; }
    uint64_t ocap = c->cap, *ok = c->keys; int *oc = c->cnt;   /* grow and retry */
    c->cap *= 2; c->used = 0;
    c->keys = (uint64_t *)calloc(c->cap, sizeof(uint64_t)); c->cnt = (int *)calloc(c->cap, sizeof(int));
    for (uint64_t i = 0; i < ocap; ++i) if (ok[i]) { *kc_slot(c, ok[i] - 1, 1) = oc[i]; }
    free(ok); free(oc);
  }
}
/* KmerCode::Append / IsValid / GetCanonicalKmerCode (KmerCode.hpp:94-109, 85-92, 54-67) rolled over a read: calls back with
 * the canonical code of every valid k-mer in read order. */
typedef void (*kc_visit)(void *ctx, uint64_t code);
static void kc_each_valid(int k, const char *r, int len, kc_visit f, void *ctx) {
  const uint64_t mask = k < 32 ? ((1ull << (2 * k)) - 1ull) : ~0ull;
  uint64_t code = 0;
  int invalidPos = -1;
  for (int i = 0; i < len; ++i) {
    if (invalidPos != -1) ++invalidPos;
    code = ((code << 2) & mask) | (uint64_t)(nuc(r[i]) & 3);
    if (r[i] == 'N') invalidPos = 0;
    if (invalidPos >= k) invalidPos = -1;
    if (i < k - 1 || invalidPos != -1) continue;
    uint64_t cr = 0;
    for (int j = 0; j < k; ++j) cr = (cr << 2) | (3ull - ((code >> (2 * j)) & 3ull));
    f(ctx, cr < code ? cr : code);
  }
}
static void kc_visit_add(void *ctx, uint64_t code) { ++*kc_slot((t4o_kc *)ctx, code, 1); }
int t4o_kc_add(t4o_kc *c, const char *read) {   /* KmerCount.hpp:64-97 */
  int len = (int)strlen(read);
  if (len < c->k) return 0;
  kc_each_valid(c->k, read, len, kc_visit_add, c);
  if (len > c->maxReadLen) c->maxReadLen = len;
  return 1;
}
struct kc_stat_ctx { t4o_kc *c; int *buf; int n, sum; };
static void kc_visit_stat(void *ctx, uint64_t code) {
  struct kc_stat_ctx *s = (struct kc_stat_ctx *)ctx;
  int *p = kc_slot(s->c, code, 0);
  int v = p ? *p : 0;
  if (v <= 0) v = 1;
  s->buf[s->n++] = v; s->sum += v;
}
static int kc_cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int t4o_kc_stats(t4o_kc *c, char *read, char *qual, int *minCnt, int *medianCnt, float *avgCnt) {   /* KmerCount.hpp:177-288 */
  if (c->maxReadLen == -1) return 0;
  int len = (int)strlen(read), i, j;
  if (len < c->k) { *minCnt = *medianCnt = -1; *avgCnt = -1; return 0; }
  struct kc_stat_ctx st; st.c = c; st.buf = (int *)calloc((size_t)(len + 1), sizeof(int)); st.n = 0; st.sum = 0;
  /* (zeroed: after a trim the reference may sort past the counts of this read into whatever its reused buffer held -- reads
   * with N AND a low-quality tail; entries it never wrote are taken as 0 here and in the engine) */
  kc_each_valid(c->k, read, len, kc_visit_stat, &st);
  int k = st.n;
  if (k == 0) { *minCnt = -len; *medianCnt = -len; *avgCnt = (float)-len; if (qual) read[0] = '\0'; free(st.buf); return 0; }
  if (qual) {
    for (i = k - 1; i >= 0; --i) if (st.buf[i] > 1) break;
    ++i;
    int badCnt = 0, trimStart = -1;
    for (j = len - 1; j >= i + c->k - 1; --j)
      if (qual[j] - 32 <= 15) { ++badCnt; if (badCnt >= 0.1 * (len - j)) trimStart = j; }
    if (trimStart > 0) { k = trimStart - c->k + 1; read[trimStart] = '\0'; qual[trimStart] = '\0'; }
    if (trimStart > 0 && trimStart < c->k) { k = 0; read[0] = '\0'; qual[0] = '\0'; }
  }
  if (k > 0) qsort(st.buf, (size_t)k, sizeof(int), kc_cmp_int);   /* std::sort(c, c + k); k <= 0 sorts nothing */
  *minCnt = st.buf[0]; *medianCnt = st.buf[k > 0 ? k / 2 : 0]; *avgCnt = (float)(st.sum / (double)k);
  for (i = 0; i < len; ++i)   /* scans the OLD length: characters behind the NUL are still there */
    if (read[i] == 'N') { if (*minCnt >= 0) *minCnt = 0; else if (*minCnt <= 0) --*minCnt; }
  free(st.buf);
  return 1;
}

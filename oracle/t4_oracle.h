/* oracle/t4_oracle.h -- CPU ORACLE (test infrastructure; see t4_oracle.c header). */
#ifndef T4_ORACLE_H
#define T4_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct t4o_set t4o_set;

typedef struct {
  int seqIdx, readStart, readEnd, seqStart, seqEnd, strand, matchCnt, indelCnt;
  double similarity;
} t4o_overlap;

t4o_set *t4o_new(int k);
void t4o_free(t4o_set *s);
void t4o_set_hit_len_required(t4o_set *s, int l);
void t4o_set_radius(t4o_set *s, int r);
void t4o_set_novel_similarity(t4o_set *s, double v);
void t4o_set_consider_barcode(t4o_set *s, int v);
int t4o_size(const t4o_set *s);
int t4o_seq_len(const t4o_set *s, int i);
const char *t4o_seq_name(const t4o_set *s, int i);
const char *t4o_seq_consensus(const t4o_set *s, int i);
int t4o_nomatch_gap_limit(const t4o_set *s);

/* InputRefFa (SeqSet.hpp:2673-2865): whole file, or one record at a time. */
int t4o_load_ref_fasta(t4o_set *s, const char *path);
int t4o_add_ref_record(t4o_set *s, const char *id, const char *seq);
/* InputNovelRead (SeqSet.hpp:3028-3073); posweight (4 ints/base) may be NULL (=> count 1 per base). */
int t4o_add_novel_seq(t4o_set *s, const char *name, const char *seq, int strand, int barcode,
                      const int *posweight);

int t4o_hits(t4o_set *s, const char *read, int strand, int barcode, int allowTotalSkip,
             int doSort, int *out5, int cap);
int t4o_overlaps_from_hits(t4o_set *s, const char *read, int strand, int barcode,
                           int allowTotalSkip, int hitLenRequired, int filter, t4o_overlap *out,
                           int cap, int *chainOff, int *coords, int coordCap);
int t4o_overlaps_from_read(t4o_set *s, const char *read, int strand, int barcode, int readType,
                           int skipRepeats, t4o_overlap *out, int cap);
int t4o_annotate_read0(t4o_set *s, const char *read, t4o_overlap out[4]);
int t4o_extend_overlap(t4o_set *s, const char *read, double mmFactor, const t4o_overlap *in,
                       t4o_overlap *out);
int t4o_assign_read(t4o_set *s, const char *read, int strand, int barcode, t4o_overlap *out);
/* RecomputePosWeight (SeqSet.hpp:4705-4738) from assigned reads; t4o_seq_posweight reads a contig's weights back (4 ints / base) */
void t4o_recompute_posweight(t4o_set *s, int n, const char *const *reads, const t4o_overlap *assign);
int t4o_update_all_consensus_chars(t4o_set *s);   /* UpdateConsensus of every contig, characters only (the index is not rebuilt) */
void t4o_seq_posweight(t4o_set *s, int i, int *out);
int t4o_kmer_length(t4o_set *s);

int t4o_global_alignment(const char *t, int lent, const char *p, int lenp, signed char *align);
int t4o_global_alignment_posweight(const int *w, int lent, const char *p, int lenp,
                                   signed char *align);
int t4o_has_hit_in_set(t4o_set *s, const char *read, int mode);   /* SeqSet::HasHitInSet (SeqSet.hpp:3144-3327) */
int t4o_process_read(const char *r1, const char *q1, const char *r2, const char *q2, char *outR, char *outQ, int *flags);   /* ProcessRead (main.cpp:224-449) */
int t4o_is_mate_overlap(const char *fr, int flen, const char *sr, int slen, int minOverlap,
                        int *offset, int *bestMatchCnt, int checkTandem);
int t4o_lis(const int *pairs, int n, int *out);

/* Batch helpers used by bench.py's cpu_baseline leg and by the parity tests: reads are fixed-stride
 * NUL-terminated records. Returns the total number of _hit records emitted (the H_r of SURVEY 8d). */
int64_t t4o_annotate_batch(t4o_set *s, const char *reads, int stride, int64_t n, t4o_overlap *out4,
                           int64_t *hitsPerRead);

/* KmerCount (KmerCount.hpp:64-97, 177-288): canonical k-mer counts of a read set; min / median / mean count of a read and the
 * quality trimming that rides on it. read / qual are modified in place like the reference's (NUL at the trim position). */
typedef struct t4o_kc t4o_kc;
t4o_kc *t4o_kc_new(int k);
void t4o_kc_free(t4o_kc *c);
int t4o_kc_add(t4o_kc *c, const char *read);
int t4o_kc_stats(t4o_kc *c, char *read, char *qual, int *minCnt, int *medianCnt, float *avgCnt);

#ifdef __cplusplus
}
#endif
#endif

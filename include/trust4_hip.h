/* include/trust4_hip.h -- C ABI of libt4hip.so, the MI355X (gfx950) engine for TRUST4's stage-1
 * k-mer seed -> hit chaining -> banded-DP scoring hot path.
 *
 * The reference has no plugin / FFI interface: its algorithms are member functions of the
 * header-only classes SeqSet / KmerIndex / AlignAlgo that main.cpp includes (main.cpp:11-13).
 * This header therefore defines the boundary a host driver binds instead of calling those
 * members; each entry point names the reference interface it replaces. All arguments are plain
 * pointers and sizes, no C++ or torch types. Calls on one t4_ctx are not re-entrant (as
 * SeqSet's Add path, SeqSet.hpp:206); use one ctx per GPU / host thread.
 * Every function returns 0 (T4_OK) or a negative error code; t4_last_error() gives the text.
 * Nothing here ever calls exit() or throws (the reference exit(1)s, e.g. main.cpp:887-891).
 */
#ifndef TRUST4_HIP_H
#define TRUST4_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define T4_OK 0
#define T4_ERR_ARG (-1)         /* bad argument */
#define T4_ERR_HIP (-2)         /* HIP runtime error (no device, OOM, launch failure) */
#define T4_ERR_IO (-3)          /* cannot read a file */
#define T4_ERR_UNSUPPORTED (-4) /* input outside the engine's limits */
#define T4_ERR_STATE (-5)       /* call order violated (e.g. index not committed) */

typedef struct t4_ctx t4_ctx;     /* one GPU + one HIP stream + scratch */
typedef struct t4_index t4_index; /* device image of one SeqSet: sequences + k-mer index */
typedef struct t4_batch t4_batch; /* 2-bit packed reads resident in HBM */

/* struct _overlap (SeqSet.hpp:76-136) without its heap members. 40 bytes. */
typedef struct {
  int32_t seqIdx, readStart, readEnd, seqStart, seqEnd, strand, matchCnt, indelCnt;
  double similarity;
} t4_overlap;

/* struct _hit (SeqSet.hpp:53-74). 20 bytes. */
typedef struct {
  int32_t idx, offset, readOffset, strand, repeats;
} t4_hit;

/* ---- context --------------------------------------------------------------------------------- */
/* device_ordinal: HIP device index. Fails with T4_ERR_HIP when no GPU is present: there is no CPU
 * fallback in this library. */
int t4_init(int device_ordinal, t4_ctx **out);
void t4_destroy(t4_ctx *ctx);
int t4_sync(t4_ctx *ctx);
const char *t4_last_error(t4_ctx *ctx);
/* number of compute units of the device (for sizing persistent grids / reporting) */
int t4_device_cus(t4_ctx *ctx);

/* ---- sequence set + k-mer index  (replaces SeqSet::SeqSet, InputRefFa, InputNovelRead and
 *      KmerIndex::BuildIndexFromRead; SeqSet.hpp:2557-2576, 2673-2865, 3028-3073;
 *      KmerIndex.hpp:118-141) ----------------------------------------------------------------- */
int t4_index_create(t4_ctx *ctx, int kmer_length, int consider_barcode, t4_index **out);
void t4_index_destroy(t4_index *ix);
/* SeqSet::SetHitLenRequired / SetRadius / SetNovelSeqSimilarity (SeqSet.hpp:2600-2614) */
int t4_index_set_params(t4_index *ix, int hit_len_required, int radius, double novel_seq_similarity);
/* SeqSet::InputRefFa(path) for a gene FASTA (plain or .gz): '/OR' filter, '.' removal, non-ACGT
 * -> N, de-duplication with name merging, then index every sequence. */
int t4_index_load_ref_fasta(t4_index *ix, const char *path);
/* The same, one record at a time (id = FASTA name token, seq = raw sequence line(s)). Returns the
 * sequence id through *seq_id (-1 when the record was filtered or merged). */
int t4_index_add_ref_record(t4_index *ix, const char *id, const char *seq, int *seq_id);
/* SeqSet::InputNovelRead semantics for a contig: consensus (forward strand), barcode (-1: none),
 * posweight = 4 int32 counts per base (NULL: count 1 on the consensus base). */
int t4_index_add_contig(t4_index *ix, const char *name, const char *consensus, int barcode,
                        const int32_t *posweight, int *seq_id);
/* Build the device image (CSR postings + lookup table + sequence table). Must be called after the
 * last add and before any query. */
int t4_index_commit(t4_index *ix);
/* Device image of a MUTATED contig set: the sequences added so far plus an explicit posting multiset
 * (k-mer code, KmerIndex bucket, seq id, offset), i.e. the state KmerIndex::Insert / Remove /
 * UpdateIndexFromRead (KmerIndex.hpp:66-181) left behind, which is not a function of the sequences alone. */
int t4_index_commit_postings(t4_index *ix, int64_t n, const uint64_t *code, const int32_t *bucket,
                             const int32_t *idx, const int32_t *offset);
/* ---- mutable contig set on the device: the image is patched, never rebuilt ---------------------
 * Replaces what KmerIndex::Insert / Remove / UpdateIndexFromRead / RemoveIndexFromRead (KmerIndex.hpp:66-101, 144-201)
 * and the consensus / posWeight edits of SeqSet::AddRead (SeqSet.hpp:3988-4115, 4168-4360) do to the reference's in-memory
 * set, for the device image of a set whose index is NOT keyed by barcode. The posting multiset of a mutated set is history
 * dependent (not a function of the sequences), so the caller -- t4_assembler, which replays those edits on its host
 * replica -- describes the change itself, by position in the image's four arrays:
 *   table     open addressing on the k-mer code, slot = first free of mix64(code) & (table_slots - 1), +1, ... ; a slot is
 *             (code, start, cnt): the key's postings are post[start .. start + cnt); keys are never removed (cnt may be 0)
 *   post      (seq id, offset) pairs; the order inside a list is free (nothing downstream of SortHits observes it)
 *   seqs      per contig: where its bases live, length, barcode, first 8 chars of the name
 *   bases     consensus chars and IsBaseEqual predicate bytes (bit x = sum < 3 * count[x], bit 4 = sum == 0;
 *             AlignAlgo.hpp:49-55) in one offset space, one terminator column after every contig
 * Every destination is written once per call with its final value; capacities only grow (contents are kept), except
 * that table_rebuilt != 0 means the table was re-hashed: it is emptied first and the delta carries every key.
 * The first call turns an empty t4_index (consider_barcode == 0) into a live set; t4_add_query / t4_overlaps / t4_extend /
 * t4_assign then run against it. Queries already launched are ordered before the patch on the ctx's stream. */
typedef struct { int64_t base_off; int32_t len, barcode; char name[8]; } t4_seq_record;
typedef struct {
  int64_t table_slots;   /* power of two */
  int32_t table_rebuilt;
  int32_t seq_cap;
  int64_t post_cap, base_cap;
  int32_t nseq, max_seq_len;
  int64_t n_slots; const int64_t *slot; const uint64_t *slot_code; const uint32_t *slot_start, *slot_cnt;
  int64_t n_post_runs; const int64_t *post_at; const int32_t *post_len; const int32_t *post_data;   /* (idx, offset) pairs of all runs, concatenated */
  int32_t n_seqs; const int32_t *seq_id; const t4_seq_record *seq;
  int64_t n_base_runs; const int64_t *base_at; const int32_t *base_len; const char *base_cons; const uint8_t *base_pw;   /* bytes of all runs, concatenated */
} t4_index_delta;
int t4_index_apply_delta(t4_index *ix, const t4_index_delta *delta);

/* Forget every sequence (keeps k, barcode mode and parameters) so the handle can be re-filled. */
int t4_index_clear(t4_index *ix);
int t4_index_size(const t4_index *ix);
int t4_index_seq_len(const t4_index *ix, int seq_id);
const char *t4_index_seq_name(const t4_index *ix, int seq_id);
const char *t4_index_seq_consensus(const t4_index *ix, int seq_id);

/* ---- reads (replaces the strdup'ed char* reads of main.cpp:845-878) -------------------------- */
/* bases: concatenated read sequences (alphabet ACGTN, upper case); offsets[n+1]: start of each read
 * in `bases`; barcode: per-read transformed barcode or NULL (= -1 for all). The reads are 2-bit
 * packed (+1 bit/base N mask) on the host and copied to HBM. */
int t4_reads_upload(t4_ctx *ctx, const char *bases, const int64_t *offsets, const int32_t *barcode,
                    int64_t n_reads, t4_batch **out);
/* The same with flags. T4_READS_KMERS_ONLY: upper-case letters other than ACGTN are taken as KmerCode::Append takes them
 * (KmerCode.hpp:99-106: nucToNum[c - 'A'] & 3 = 3, a VALID 'T'; only 'N' invalidates a k-mer). Such a batch is for the
 * queries that look at k-mer codes only -- t4_has_hit (the stage-0 filter), t4_hits, t4_kmer_count_* -- where this is exactly
 * the reference. The alignment queries are refused such letters without the flag and must not be given such a batch: the
 * reference compares the letter itself in GlobalAlignment and reads _posWeight::count[-1] for it in IsBaseEqual
 * (AlignAlgo.hpp:49-55 with nucToNum = -1), which is not behaviour a drop-in can reproduce. */
#define T4_READS_KMERS_ONLY 1
int t4_reads_upload_flags(t4_ctx *ctx, const char *bases, const int64_t *offsets, const int32_t *barcode,
                          int64_t n_reads, int flags, t4_batch **out);
void t4_batch_destroy(t4_batch *b);
int64_t t4_batch_size(const t4_batch *b);

/* ---- queries --------------------------------------------------------------------------------- */
/* SeqSet::GetHitsFromRead + SortHits (SeqSet.hpp:1341-1501, 1306-1339) for every read: hits are
 * returned ordered by (strand, idx, readOffset, offset). Parity/debug entry point.
 * hit_offsets[n+1] receives the CSR offsets. If hits == NULL only hit_offsets is filled (size
 * query). hits_cap = capacity of `hits` in records. */
int t4_hits(t4_index *ix, t4_batch *b, int strand, int allow_total_skip, int64_t *hit_offsets,
            t4_hit *hits, int64_t hits_cap);

/* SeqSet::GetOverlapsFromRead(read, strand, barcode, readType 0, skipRepeats) (SeqSet.hpp:1508-2124)
 * for every read. counts[i] receives the function's return value for read i (-1: shorter than k);
 * out receives at most max_per_read overlaps per read at out[i*max_per_read ...], in the order
 * of the reference's result vector. */
int t4_overlaps(t4_index *ix, t4_batch *b, int strand, int skip_repeats, int max_per_read,
                int32_t *counts, t4_overlap *out);

/* SeqSet::AnnotateRead(read, 0, geneOverlap, NULL, NULL) (SeqSet.hpp:6016-6321) against a reference
 * gene set: out[4*i + g] is geneOverlap[g] of read i (g: 0 V, 1 D (always -1), 2 J, 3 C). Fields of
 * entries with seqIdx == -1 are unspecified (the reference leaves stale data there). This is the
 * rough-annotation pass of main.cpp:1084-1120. */
int t4_annotate_rough(t4_index *ref, t4_batch *b, t4_overlap *out);

/* SeqSet::ExtendOverlap(r, len, seq, mismatch_factor, align, overlap, extendedOverlap) (SeqSet.hpp:1165-1277) for
 * caller-supplied overlaps of every read (layout of t4_overlaps: counts[i] overlaps at in[i*max_per_read ...]); every
 * overlap is aligned against the read strand it names. ret receives the function's return value (0 / 1) and out the
 * extendedOverlap of a call that starts from a default-constructed _overlap. Contig sets only (posWeight). */
int t4_extend(t4_index *ix, t4_batch *b, int max_per_read, const int32_t *counts, const t4_overlap *in,
              double mismatch_factor, int32_t *ret, t4_overlap *out);

/* SeqSet::AssignRead(read, strand, barcode, assign) (SeqSet.hpp:4632-4701): ret[i] = its return value (seq id or
 * -1), out[i] = `assign` (seqIdx -1 when no overlap extends over the whole read). Contig sets only. */
int t4_assign(t4_index *ix, t4_batch *b, int strand, int32_t *ret, t4_overlap *out);

/* The same with the strand argument of every read given apiece: the bulk `_final.out` tail assigns every assembled read
 * with the strand its AddRead settled on (main.cpp:2075-2116, AssignReads_Thread 607-626). */
int t4_assign_strands(t4_index *ix, t4_batch *b, const int32_t *strands, int32_t *ret, t4_overlap *out);

/* SeqSet::RecomputePosWeight (SeqSet.hpp:4705-4738; main.cpp:2118) on a committed contig set: every posWeight column is zeroed,
 * every read with assign[i].seqIdx != -1 adds mult[i] (NULL: 1) to the column of each of its non-N bases on the strand
 * assign[i].strand from column assign[i].seqStart on (UpdatePosWeightFromRead, SeqSet.hpp:2466-2474), and columns no read covers
 * get count 1 on their consensus base. posweight receives 4 int32 (A, C, G, T) per base, contigs in id order (what
 * t4_index_add_contig took), posweight_cap = its capacity in int32 values. The set's own image is not changed (rebuild it
 * with the new weights when queries against them are needed). */
int t4_posweight_recompute(t4_index *ix, t4_batch *b, const t4_overlap *assign, const int32_t *mult, int32_t *posweight,
                           int64_t posweight_cap);

/* The same followed by SeqSet::UpdateConsensus (SeqSet.hpp:4537-4588; UpdateAllConsensus, 4525-4535) of every contig from the
 * rebuilt columns (in main.cpp the tail's ExtendSeqFromReads stands between the two, 2118 ... 2154, and runs from the reference's
 * translation unit: the binding of INTEGRATION.md uses t4_posweight_recompute; this entry is the consensus kernel for callers that
 * keep their contigs in the engine): per column the base with the largest count -- the first of equals -- replaces the consensus base when
 * that one is strictly rarer; columns without counts keep theirs. consensus receives the bases of all contigs in id order, without
 * separators (consensus_cap = its capacity), *changed the number of bases that differ from the set's (either may be NULL: without
 * `consensus` this is t4_posweight_recompute). The k-mer index of a set whose consensus changes is the caller's to rebuild
 * (UpdateConsensus does it with RemoveIndexFromRead + BuildIndexFromRead of the contig; here: t4_index_clear / add_contig / commit). */
int t4_consensus_recompute(t4_index *ix, t4_batch *b, const t4_overlap *assign, const int32_t *mult, int32_t *posweight,
                           int64_t posweight_cap, char *consensus, int64_t consensus_cap, int64_t *changed);

/* AlignAlgo::GlobalAlignment (kind 0; AlignAlgo.hpp:218-424; t_data = chars) or
 * AlignAlgo::GlobalAlignment_PosWeight (kind 1; AlignAlgo.hpp:57-216; t_data = 4 int32 weights per base)
 * for n independent (target, pattern) pairs given as CSR offsets; out4[4*i..] = GetAlignStats of
 * the reference's alignment (matches, mismatches, indels) and a status word (0 ok, 1 beyond the
 * engine's gap limits). impl 0 = the forward-only LDS formulation the overlap scorer uses (falls back
 * to the traceback formulation for wide bands), impl 1 = traceback formulation only, impl 2 = one alignment per
 * wavefront (status 2 in out4[3] for bands wider than 64 columns), impl 3 = eight alignments per wavefront, two per
 * 16-lane DPP row (status 2 for bands wider than 16 columns): the two formulations overlap scoring runs on the GPU. */
int t4_gap_dp(t4_ctx *ctx, int kind, int impl, int n, const int64_t *t_off, const int64_t *p_off,
              const void *t_data, const char *p_chars, int32_t *out4);
/* impl 4, kind 1: the scratch-row aligner with its traceback; `align` receives the edit string of every alignment (AlignAlgo.hpp:160-205: 0 match, 1 mismatch,
 * 2 insert, 3 delete, terminated by -1; align_stride bytes per alignment, >= lent + lenp + 2) */
int t4_gap_dp_align(t4_ctx *ctx, int kind, int impl, int n, const int64_t *t_off, const int64_t *p_off, const void *t_data,
                    const char *p_chars, int32_t *out4, signed char *align, int align_stride);

/* The query half of SeqSet::AddRead for a (small) batch of reads in ONE launch and one host round trip:
 * GetOverlapsFromRead(read, strands[i], barcode, 0, skip_repeats) (SeqSet.hpp:3437) and the ExtendOverlap
 * (SeqSet.hpp:3597 / 3746, mismatch factor factors[i]) of every overlap it returns. Layout as t4_overlaps /
 * t4_extend. Used by t4_assembler for its speculation windows. Contig sets only. */
int t4_add_query(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes,
                 const int32_t *strands, int skip_repeats, const double *factors, int max_per_read, int32_t *counts,
                 t4_overlap *ov, t4_overlap *ext, int32_t *ext_ret);

/* SeqSet::HasHitInSet(read, mode) (SeqSet.hpp:3144-3327), the candidate test of the stage-0 extractors
 * (FastqExtractor.cpp:129-134 calls it with mode 0 on reads that pass its IsLowComplexity): out[i] = -1 / 0 / 1
 * (hit on the minus strand / no hit / hit on the plus strand). Uses the set's hit_len_required. Only mode 0 is built. */
int t4_has_hit(t4_index *ref, t4_batch *b, int mode, int32_t *out);

/* AlignAlgo::IsMateOverlap (AlignAlgo.hpp:1027-1096; ProcessRead, main.cpp:292-330) for a batch of pairs: is a suffix of
 * `first` a prefix of `second` at exactly one offset (length-dependent identity threshold, optional tandem-repeat veto)?
 * Reads as concatenated ACGTN chars with n + 1 offsets each. out3[3*i..]: the function's return value (overlap size or
 * -1), and its `offset` / `bestMatchCnt` outputs (those of the last passing offset; -1 when none passed). */
int t4_mate_overlap(t4_ctx *ctx, int n, const int64_t *first_off, const char *first_chars, const int64_t *second_off,
                    const char *second_chars, const int32_t *min_overlap, int check_tandem, int32_t *out3);

/* ProcessRead (main.cpp:224-449) with IsLowComplexity (183-205) for n mate pairs, one pair per wavefront: read-through clipping
 * (IsMateOverlap of rc(read 2) against read 1: read 1 is cut to the overlap and takes read 2's base wherever that has the better
 * quality), else mate merging (IsMateOverlap of read 1 against rc(read 2), tandem repeats refused: one read of weight 2 when at
 * least 95 % of the overlap agrees, otherwise the mate of better quality stands for both), else both mates stay. off1 / off2: n + 1
 * offsets into r1 / r2 (characters as read, any letter) and q1 / q2 (qualities; NULL when no read has any); has_qual[i]: bit 0
 * read 1 of pair i has qualities, bit 1 read 2. out_off: n + 1 offsets into out_r / out_q with room for len1 + len2 + 1 per pair.
 * meta4 receives 4 int32 per pair: kind (0 both mates stay, 1 read-through, 2 merged, 3 one mate for both), length of read 1
 * afterwards, flags (1 read 1 kept = not of low complexity, 2 read 2 kept, 4 weight 2: the caller lists the merged read twice,
 * the copy's id with ".1" appended, 8 read 1 has qualities, 16 read 1 changed: its bases / qualities are at out_off[i]), 0.
 * A read 2 that stays is what ProcessRead leaves: reverse-complemented twice, i.e. letters other than ACGT turned into N. */
int t4_process_pairs(t4_ctx *ctx, int n, const int64_t *off1, const char *r1, const char *q1, const int64_t *off2, const char *r2,
                     const char *q2, const unsigned char *has_qual, const int64_t *out_off, char *out_r, char *out_q, int32_t *meta4);

/* ---- ordered contig builder (host-side commit logic + GPU queries) ----------------------------------
 * t4_assembler owns a mutable set of novel contigs (the reference's `SeqSet seqSet`, main.cpp:642) and exposes
 * the members stage 1 calls on it, with the same arguments and return values:
 *   t4_assembler_input_novel_read   SeqSet::InputNovelRead      (SeqSet.hpp:3028-3073)
 *   t4_assembler_add_read           SeqSet::AddRead             (SeqSet.hpp:3426-4473)  >= 0 contig id, -1, -2
 *   t4_assembler_repeat_add_read    SeqSet::RepeatAddRead       (SeqSet.hpp:4477-4507)
 *   t4_assembler_update_all_consensus SeqSet::UpdateAllConsensus (SeqSet.hpp:4525-4535)
 *   t4_assembler_output             SeqSet::Output(fp, NULL)    (SeqSet.hpp:10939-10994)  the _raw.out records
 * GetOverlapsFromRead and every ExtendOverlap of an AddRead run on the GPU (t4_overlaps + t4_extend against a
 * device image of the set); the order-dependent bookkeeping runs on the host. Return values below -50 are
 * -100 + (a T4_ERR_* code). */
typedef struct t4_assembler t4_assembler;
int t4_assembler_create(t4_ctx *ctx, int kmer_length, int consider_barcode, t4_assembler **out);
void t4_assembler_destroy(t4_assembler *a);
int t4_assembler_set_params(t4_assembler *a, int hit_len_required, int radius, double novel_seq_similarity);
int t4_assembler_input_novel_read(t4_assembler *a, const char *id, const char *read, int strand, int barcode);
int t4_assembler_add_read(t4_assembler *a, const char *read, const char *gene_name, int *strand, int barcode,
                          int min_kmer_count, int repetitive_data, double similarity_threshold);
int t4_assembler_repeat_add_read(t4_assembler *a, const char *read);
/* Speculation window: announce the next n reads (in the order they will be offered to t4_assembler_add_read, with the
 * strand / barcode / repetitive_data arguments they will be offered with). Entries of an earlier announcement that still
 * stand are kept; the others are queried on the GPU in one batch. A result is consumed by add_read for as long as no commit
 * can have changed it: for a set whose index is not keyed by barcode that is tracked per entry (the posting lists of the
 * read's own k-mers, and the stretch of every contig it has three or more hits with), for a cell of a
 * t4_cellset any observable change ends the window. When the head entry has fallen, t4_assembler_window_valid returns 0 so
 * that the caller announces again (add_read would otherwise query that one read by itself). Never changes results. */
int t4_assembler_prefetch(t4_assembler *a, int n, const char *const *reads, const int *strands, const int *barcodes,
                          int repetitive_data);
int t4_assembler_window_valid(const t4_assembler *a);
/* Host threads that derive the window's dependency sets while the GPU runs a query batch (default 1). */
int t4_assembler_set_threads(t4_assembler *a, int host_threads);
/* Counters of a set whose index is not keyed by barcode (device image by deltas, sliding window), up to 23 values:
 * query rounds, reads queried, deltas, delta bytes, invalidations (total; by an index change of one of the read's keys; by a
 * list crossing 100 postings; by a changed region within reach; by a left extension; by a whole-contig change; by exhausted
 * tolerance), tolerated index changes, microseconds in deltas / dependency sets / event examination / query batches; then of
 * the ctx's AddRead query path: calls, reads, launches of the global-scratch tier, reads it served, result records,
 * microseconds of its kernels (HIP events), _hit records its seed stages emitted. */
int t4_assembler_live_counters(const t4_assembler *a, int64_t *out, int n);
/* The chain of dependent query rounds of a live set ("the floor"): up to 10 values -- rounds; rounds that carried restricted
 * re-queries only; 5th percentile and median of a round's kernel milliseconds (HIP events); 5th percentile and median of a round's
 * wall milliseconds from the launch call to its results on the host; whole queries; restricted re-queries (one contig); candidate
 * records kept with whole queries; restricted re-queries merged through the replay of the pre-filter scan. rounds x the 5th
 * percentile of the wall time is what the chain would cost if every round were as short as the shortest ones. */
int t4_assembler_chain_stats(const t4_assembler *a, double *out, int n);
int t4_assembler_counters(const t4_assembler *a, int64_t *queries, int64_t *refreshes, int64_t *window_hits);
/* host seconds spent refreshing the device image / in GPU query batches (upload + kernels + download) */
int t4_assembler_timers(const t4_assembler *a, double *sec_refresh, double *sec_query);
int t4_assembler_update_all_consensus(t4_assembler *a);
int t4_assembler_output(t4_assembler *a, const char *path);
int t4_assembler_size(const t4_assembler *a);
/* One slot of the set as the reference's `_seqWrapper` holds it (SeqSet.hpp:19-51): what a host driver needs to hand the contigs
 * on to reference code that stays on the host (SeqSet::InputSeqSet for the mate-pair extension tail, main.cpp:2047-2048;
 * INTEGRATION.md 5). A released slot has consensus == NULL. Pointers stay valid until the next call that changes the set. */
typedef struct {
  const char *name, *consensus;
  const int32_t *posweight;   /* 4 counts (A, C, G, T) per base */
  int32_t len, barcode, num_read, min_left_ext_anchor, min_right_ext_anchor, in_index;
} t4_contig_view;
int t4_assembler_contig(const t4_assembler *a, int i, t4_contig_view *out);
/* SeqSet::ChangeKmerLength (SeqSet.hpp:4624-4629): compacts the set (ids are renumbered) and re-indexes with the new k. */
int t4_assembler_change_kmer_length(t4_assembler *a, int kmer_length);
int64_t t4_assembler_index_postings(const t4_assembler *a);
/* SeqSet::ReleaseFinishedBarcodeSeq({barcode}, removeFromIndex = true, contigMinCov, earlyStop = true)
 * (SeqSet.hpp:10815-10935; main.cpp:1846-1859): the trailing contigs of that barcode leave the index and are final
 * (shallow ones, SeqSet::IsContigShallow, are dropped when contig_min_cov > 0). */
int t4_assembler_release_finished_barcode(t4_assembler *a, int barcode, int contig_min_cov);
/* SeqSet::ReleaseShallowContigs (SeqSet.hpp:10926-10936; main.cpp:1952-1955) */
int t4_assembler_release_shallow_contigs(t4_assembler *a, int min_cov);
/* SeqSet::Output(fp, &barcodeIntToStr) of a set that holds several barcodes (--keepNoBarcode, main.cpp:1968-1969) */
int t4_assembler_output_barcodes(t4_assembler *a, const char *path, const char *const *barcode_names, int n_names);

/* ---- per-barcode contig sets (barcode mode, main.cpp:1549-1559) ---------------------------------------
 * With --barcode the reference keys the k-mer index by barcode (SetConsiderBarcodeInIndexHash, KmerIndex.hpp:29-33)
 * and filters every hit to the read's barcode (SeqSet.hpp:1418, 1485): contigs of different cells never meet, so its
 * one `SeqSet seqSet` is a disjoint union of per-cell sets processed one after the other (main.cpp:1126, 1183-1192).
 * t4_cellset exposes that structure: t4_cellset_cell returns the t4_assembler of one barcode (contig ids local to the
 * cell; every t4_assembler_* call above works on it, with that barcode as the `barcode` argument), and
 * t4_cellset_prefetch runs the AddRead queries of the next reads of MANY cells in one launch against per-cell device
 * images, which is what lets the order-dependent Add path batch without conflicts. The results are what
 * the reference's sequential pass over the cells produces; t4_cellset_output numbers the contigs as that pass would
 * (cells in ascending barcode id, creation order inside a cell) and prints SeqSet::Output(fp, &barcodeIntToStr)
 * (SeqSet.hpp:10939-10994). Barcode ids must be < 1000003 (beyond that the reference lets two barcodes share lists). */
typedef struct t4_cellset t4_cellset;
int t4_cellset_create(t4_ctx *ctx, int kmer_length, t4_cellset **out);
void t4_cellset_destroy(t4_cellset *cs);   /* also destroys its cells */
int t4_cellset_set_params(t4_cellset *cs, int hit_len_required, int radius, double novel_seq_similarity);
int t4_cellset_cell(t4_cellset *cs, int barcode, t4_assembler **cell);   /* get or create */
int t4_cellset_close_cell(t4_cellset *cs, t4_assembler *cell);           /* no more queries: recycle its device slot */
/* read i is the next read cells[i] will be offered by t4_assembler_add_read (strand argument strands[i]); a cell may
 * contribute several CONSECUTIVE entries, in the order its reads will come (its speculation window: it stands until a
 * commit changes anything a query of that cell can observe). */
int t4_cellset_prefetch(t4_cellset *cs, int n, t4_assembler *const *cells, const char *const *reads, const int *strands,
                        int repetitive_data);
/* Host threads used inside t4_cellset_prefetch for the per-cell image builds and window bookkeeping (default 1).
 * Different cells may also be driven from different caller threads between two prefetch calls (t4_assembler_* on a cell
 * touches that cell only, provided every AddRead it is offered was prefetched); one cell is never re-entrant. */
int t4_cellset_set_threads(t4_cellset *cs, int host_threads);
int t4_cellset_update_all_consensus(t4_cellset *cs);
int t4_cellset_release_shallow_contigs(t4_cellset *cs, int min_cov);
int t4_cellset_size(const t4_cellset *cs);   /* contig slots over all cells == seqSet.Size() */
int t4_cellset_output(t4_cellset *cs, const char *path, const char *const *barcode_names, int n_names);
/* SeqSet::Output(fp, &barcodeIntToStr) (SeqSet.hpp:10939-10994) of a t4_cellset that holds a contiguous range of the sample's cells:
 * contig ids start at id_base (= the contig slots of the sets that hold the cells before it) and, with append != 0, the records
 * follow what `path` already holds. Several cell sets (one per t4_ctx / stream / host thread, so that one set's query batch runs
 * while the others commit) then write the file the single set would have written. */
int t4_cellset_output_at(t4_cellset *cs, const char *path, const char *const *barcode_names, int n_names, int id_base, int append);
int t4_cellset_counters(const t4_cellset *cs, int64_t *query_batches, int64_t *reads_queried, int64_t *images_staged,
                        int64_t *bytes_staged, double *sec_query, double *sec_stage);

/* ---- k-mer counts of the read set ------------------------------------------------ */
/* KmerCount (KmerCount.hpp): counts of the canonical k-mers of the reads (main.cpp:905-915 counts 21-mers of every read that
 * enters stage 1) and, per read, the minimum / median / mean count of its valid k-mers with the quality trimming that rides on
 * them (GetCountStatsAndTrim, main.cpp:980-1061; they define the read order, main.cpp:103-125, and the AddRead thresholds,
 * 1676-1701). Replaces KmerCount::AddCount (KmerCount.hpp:64-97) and GetCountStatsAndTrim (177-288) for whole batches: the
 * table lives in HBM (open addressing on the canonical code, 12 bytes per slot), one wavefront per read. Counts are exact.
 * k <= 31; `max_kmers` bounds the number of DISTINCT k-mers (the table gets at least twice as many slots; an insert that finds
 * it full makes t4_kmer_count_add fail with T4_ERR_UNSUPPORTED). the stage-1 driver calls this under T4_GPU_KMERCOUNT=1 and counts on host threads otherwise. */
typedef struct t4_kmer_counter t4_kmer_counter;
/* per_barcode != 0: one KmerCount per barcode in the same table -- the reference's `KmerCount barcodeKmerCount(21, 23)` that is
 * filled, read and cleared barcode after barcode (main.cpp:1126-1160): the read's barcode (t4_reads_upload; at most 2^20 - 2) is
 * part of the key, so reads only see the counts of their own barcode. Takes k <= 21. */
int t4_kmer_count_create(t4_ctx *ctx, int k, int64_t max_kmers, int per_barcode, t4_kmer_counter **out);
void t4_kmer_count_destroy(t4_kmer_counter *kc);
/* AddCount of every read of the batch (reads shorter than k add nothing). */
int t4_kmer_count_add(t4_kmer_counter *kc, t4_batch *reads);
/* KmerCount::AddCountFromFile (KmerCount.hpp:99-120; `-c FILE`, main.cpp:694-699) after the file was parsed on the host: the count of
 * code[i] becomes counts[i] (codes as written in the file, NOT made canonical -- the reference does not either; of two records
 * of one k-mer the later one stays). Not for per-barcode counters. */
int t4_kmer_count_set(t4_kmer_counter *kc, const uint64_t *codes, const int32_t *counts, int64_t n);
/* GetCountStatsAndTrim of every read of the batch. quals == NULL: no trimming (the reference's qual == NULL). Otherwise the
 * qualities of read i are the bytes quals[qual_off[i] .. qual_off[i + 1]) and must be as many as the read has bases.
 * Out (n entries each): min_cnt, median_cnt, avg_cnt as the reference leaves them (-1 for reads shorter than k, -len without a
 * valid k-mer, avg = +inf when the trim leaves no k-mer), new_len = the length the reference cuts read and qualities to
 * (the read's length when it is not trimmed, 0 when it is emptied). */
int t4_kmer_count_stats(t4_kmer_counter *kc, t4_batch *reads, const char *quals, const int64_t *qual_off,
                        int32_t *min_cnt, int32_t *median_cnt, float *avg_cnt, int32_t *new_len);
/* number of distinct k-mers counted so far */
int64_t t4_kmer_count_distinct(t4_kmer_counter *kc);
/* The counts of two read sets put together. KmerCount::AddCount (KmerCount.hpp:64-97) only increments, so the table of a union of
 * read sets is the sum of the sets' tables: a run whose INPUT is dealt out by cells (trust4-hip --cellShard: every rank parses the
 * files but runs ProcessRead and the 21-mer count over the reads of its own cells only, where main.cpp:787-915 does both for the
 * whole sample) counts its reads, hands the other ranks its pairs and adds theirs. export: the table's (k-mer, count) pairs in any
 * order -- *n = how many there are; with cap == 0 nothing else happens, else the first min(*n, cap) pairs are written. merge:
 * count[codes[i]] += counts[i]; only_present != 0 passes over the pairs whose k-mer the table does not hold (the statistics of a
 * rank's reads look up those reads' k-mers only). Keys of a per-barcode counter travel as the table holds them (barcode included).
 * With only_present == 0 the table grows beyond what max_kmers of t4_kmer_count_create stood for when the other set's k-mers need the
 * room (device memory permitting); on an error the pairs of the slices before the failing one have been added. */
int t4_kmer_count_export(t4_kmer_counter *kc, uint64_t *codes, int32_t *counts, int64_t cap, int64_t *n);
int t4_kmer_count_merge(t4_kmer_counter *kc, const uint64_t *codes, const int32_t *counts, int64_t n, int only_present);

/* ---- ranks of one node: the one exchange of barcode mode, inside the engine ---------------------------------
 * Barcode mode shards by cell ranges with no exchange during assembly (main.cpp:1126-1192, 1549-1559 make cells independent); what
 * remains is ONE variable-size all-gather of every rank's contig records at the end. t4_comm does it with RCCL (ncclAllGather over
 * xGMI; one process per GPU) from C++: no file, tensor or Python in the path. The communicator is bootstrapped through a file: rank 0
 * writes the ncclUniqueId to id_path (created atomically), the other ranks wait for it (same node, shared file system).
 * t4_comm_allgather_bytes: every rank contributes n bytes; *all receives a malloc'ed buffer with the contributions of ranks
 * 0 .. nranks-1 back to back (the caller frees it), sizes[r] their lengths. Collective: every rank of the communicator calls it. */
typedef struct t4_comm t4_comm;
int t4_comm_init(t4_ctx *ctx, int rank, int nranks, const char *id_path, t4_comm **out);
int t4_comm_allgather_bytes(t4_comm *cm, const void *mine, int64_t n, void **all, int64_t *sizes);
/* the same contributions received by `root` alone (*all = NULL elsewhere): lengths by ncclAllGather, payloads by grouped ncclSend / ncclRecv, nothing padded */
int t4_comm_gather_bytes(t4_comm *cm, const void *mine, int64_t n, int root, void **all, int64_t *sizes);
void t4_comm_destroy(t4_comm *cm);

/* ---- measurement ----------------------------------------------------------------------------- */
/* Per-call statistics of the last query on this ctx: kernel time measured with HIP events on the
 * ctx's stream, number of _hit records the seed stage emitted (H of SURVEY.md 8d), number of reads
 * that went through each capacity tier. */
typedef struct {
  double kernel_ms;       /* all kernels of the call, event-timed on the ctx stream */
  double chain_kernel_ms; /* the dominant probe->sort->chain->score kernels only */
  int64_t total_hits;     /* sum over reads and passes of emitted _hit records */
  int64_t reads;          /* reads processed */
  int64_t tier_reads[6];  /* reads per capacity tier (LDS 1k / 2k / 3k / 4k / 8k hits, global scratch) */
  int64_t launches;       /* kernel launches in the call */
} t4_stats;
int t4_last_stats(t4_ctx *ctx, t4_stats *out);
/* Development aid: libraries built with -DT4_PHASE_TIMING count cycles per kernel phase (dumped by t4_destroy under
 * T4_PHASE_DUMP=1); this forgets what was counted so far. A no-op in the product build. */
int t4_debug_phase_reset(void);

#ifdef __cplusplus
}
#endif
#endif

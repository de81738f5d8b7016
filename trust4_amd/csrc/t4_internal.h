// trust4_amd/csrc/t4_internal.h -- library-internal interface between the host-side contig builder
// (t4_assembler.cpp) and the device side (t4_api.hip). Not part of the public C ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/trust4_hip.h"

extern "C" {

// Device arena of per-barcode set images (in barcode mode every cell is an independent SeqSet,
// main.cpp:1556-1559 + SeqSet.hpp:1418). One slot per open cell; a slot's image is rebuilt from the host replica of
// the cell (sequences, posWeight, explicit postings) and uploaded with all other rebuilt images of a window in one
// host-to-device copy + one scatter kernel.
typedef struct t4_cellstore t4_cellstore;
int t4_cellstore_create(t4_ctx *ctx, int kmer_length, t4_cellstore **out);
void t4_cellstore_destroy(t4_cellstore *cs);
int t4_cellstore_set_params(t4_cellstore *cs, int hit_len_required, int radius, double novel_seq_similarity);
int t4_cellstore_open(t4_cellstore *cs, int *slot);
int t4_cellstore_close(t4_cellstore *cs, int slot);
// Queue the new image of one cell. names/cons: nseq C strings ("" for released slots), pw[i]: 4 counts per base of
// sequence i; the index as nkeys lists: key (key_code, key_bucket) owns the next key_cnt (idx, offset) pairs of `post`
// (idx local to the cell, list order = the order KmerIndex holds them in).
int t4_cellstore_stage(t4_cellstore *cs, int slot, int barcode, int nseq, const char *const *names, const char *const *cons,
                       const int32_t *const *pw, int64_t nkeys, const uint64_t *key_code, const int32_t *key_bucket,
                       const int32_t *key_cnt, const int32_t *post, int64_t *pw_offset_in_image,
                       const int32_t *seq_barcodes /* nullable: per-sequence barcodes of a set that is not keyed by barcode */);
// Overwrite posWeight predicate bytes of the slot's resident image (byte offsets as returned by the last stage + the
// column's index); applied after the staged images of the same flush.
int t4_cellstore_patch(t4_cellstore *cs, int slot, int n, const int64_t *byte_offsets, const unsigned char *values);
// Exact staged size of one image, and the serial reservation that must precede a group of t4_cellstore_stage calls
// (which may then run concurrently on several host threads: nothing moves while they write).
size_t t4_cellstore_image_bytes(int nseq, int64_t nkeys, int64_t npost, int64_t cons_bytes);
int t4_cellstore_prepare(t4_cellstore *cs, int max_slot, size_t bytes);
// Flush the staged images, then run the AddRead query (== t4_add_query) of read i against the image of slot[i].
int t4_cellstore_query(t4_cellstore *cs, int n, const int32_t *slots, const char *bases, const int64_t *offsets,
                       const int32_t *barcodes, const int32_t *strands, int skip_repeats, const double *factors,
                       int max_per_read, int32_t *counts, t4_overlap *ov, t4_overlap *ext, int32_t *ext_ret);
int t4_cellstore_set_big_first(t4_cellstore *cs, int on);   // first launch on the 8192-hit tier (one big set) instead of the 1024-hit tier
int64_t t4_cellstore_bytes_staged(const t4_cellstore *cs);

// t4_add_query with variable-size results (a read may overlap thousands of contigs that share a gene segment): counts[i]
// records of read i start at index base[i] of ov / ext / ext_ret, which point into pinned memory of the ctx that stays
// valid until the next query on it. tier_hint (nullable, n bytes, in/out): nonzero = the read outgrew the LDS tiers the last
// time it was queried, so it starts on the global-scratch tier, beside the launch of the others.
int t4_add_query_pool(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                      int skip_repeats, const double *factors, const int32_t **counts, const int32_t **base, const t4_overlap **ov,
                      const t4_overlap **ext, const int32_t **ext_ret, unsigned char *tier_hint);

// the same in two halves, so that the caller's host work overlaps the kernels: begin enqueues, done polls (1 = finished or idle),
// end waits and returns the result. tier_hint must stay alive until end; one call in flight per ctx.
// only_seq (nullable): only_seq[i] >= 0 asks for the overlaps of read i with that one contig alone -- all of them, both strands,
// scored and extended, none of the steps that look across contigs applied (the restricted re-query of a window entry)
int t4_add_query_pool_begin(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                            int skip_repeats, const double *factors, unsigned char *tier_hint, const int32_t *only_seq);
// The candidate store: with want_cands the call also returns, per read, EVERY scored overlap of the pass on the strand of
// the best one as it stands before the similarity cut (a restricted re-query: every overlap with its one contig) in the order of the
// scan of SeqSet.hpp:1673-2094 -- pre-score key, scored fields, whether the pre-filters of 1705-1794 cut it -- and eight statistics
// words per read: per strand (minus, plus) the groups of >= 4 hits, of >= 5 hits, the largest group (true sizes; a restricted
// re-query: of its one contig) and the novelMinHitRequired the pass used (SeqSet.hpp:784-823); a restricted re-query adds the hull of
// the read's projections along the contig's diagonals with three or more hits (lo minus / plus, hi minus / plus; lo > hi: none):
// T4_QUERY_STATS words per read. want_cands: 1 = return the candidate records; | 2 = the image's predicate bytes carry posting marks
// (bits 5-6 of the byte at offset o: postings (contig, o) in the index, bit 7 of a contig's first byte: marks not to be trusted --
// t4_assembler::makeDelta writes them), so a restricted re-query reads the contig's postings off the contig. force_min (nullable, restricted
// re-queries): that threshold for the one contig's groups, minus | plus << 16 (0: three hits).
#define T4_QUERY_STATS 12
typedef struct { int32_t seqIdx, ss, se; int16_t rs, re, m0, matchCnt, indelCnt; uint16_t flags; } t4_cand;   // flags: 1 plus strand, 2 scoring left similarity 0, 4 cut by the pre-filters
int t4_add_query_pool_begin2(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                             int skip_repeats, const double *factors, unsigned char *tier_hint, const int32_t *only_seq, const int32_t *force_min, int want_cands);
int t4_add_query_last_cands(t4_ctx *ctx, const t4_cand **pool, const int32_t **base, const int32_t **cnt, const int32_t **stats8, int *n);
int t4_add_query_last_aux(t4_ctx *ctx, const int32_t **aux, const int32_t **n4, const int32_t **status, int *n);   // see T4QueryArgs::aux / n4
int t4_add_query_pool_done(t4_ctx *ctx);
int t4_ctx_device(t4_ctx *ctx);   // the device ordinal the ctx was created on (t4_assembler opens further ctxs beside it: one per query lane)
int t4_add_query_pool_end(t4_ctx *ctx, const int32_t **counts, const int32_t **base, const t4_overlap **ov, const t4_overlap **ext, const int32_t **ext_ret);

// AddRead query path of this ctx, 7 values: calls, reads, launches of the global-scratch tier, reads it served, result records,
// microseconds of its kernels (HIP events on the ctx's stream), _hit records its seed stages emitted
int t4_add_query_stats(t4_ctx *ctx, int64_t *out7);
int t4_add_query_last_stable(t4_ctx *ctx, const int32_t **flags, int *n);   // see T4QueryArgs::statsStable
// the wide query (csrc/t4_wide.h): dependency records of a read it served -- see t4_api.hip
typedef struct { uint32_t key, cnt; int32_t lo, hi; } t4_grp;   // key = contig * 2 + (strand == 1); hits (bits 0-23); hull of the diagonals with three or more hits (lo > hi: none)
// (records of a read with lists beyond 10000 postings -- `huge` -- carry in bits 24-27 of cnt what SeqSet.hpp:796-806 reads of the group:
// bits 24-26 its hits of lists of at most 10000 postings, capped at 4; bit 27 whether its hit lowest on the read is one)
int t4_add_query_groups(t4_ctx *ctx, int i, const t4_grp **groups, int *n, int *huge, int *n4);
int t4_add_query_wide_stats(t4_ctx *ctx, int64_t *out4);
int t4_add_query_last_call(t4_ctx *ctx, double *kernel_ms, const int32_t **ticks10ns, int *n);   // development aid (T4_ROUND_LOG)

}  // extern "C"

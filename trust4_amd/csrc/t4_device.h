// trust4_amd/csrc/t4_device.h -- POD views shared by the host side (t4_api.hip) and the kernels.
#pragma once
#include <stdint.h>

#define T4_MAXL 384          // longest read the kernels take (150 bp mates merge to <= ~290 bp)
#define T4_MAXPOS (2 * T4_MAXL)
#define T4_MAXGAP 320        // longest side of one gap DP (nomatchGapLimit is 288 at k = 9)
#define T4_DIR_BYTES 49152   // per-lane traceback bytes of one gap DP
#define T4_NTIER 6

// key of a k-mer hit, sortable as (strand, seq idx, diagonal, seq offset)
#define T4_IDX_BITS 22
#define T4_C_BITS 21
#define T4_B_BITS 20
#define T4_C_BIAS (1 << 20)
#define T4_MAX_SEQS (1 << T4_IDX_BITS)
#define T4_MAX_SEQLEN (1 << T4_B_BITS)

struct T4SeqInfo {           // 24 bytes per sequence of the set
  int consOff;               // start in the consensus char array
  int len;
  int pwOff;                 // start (in bases) in the posWeight array, -1 for reference genes
  int barcode;
  unsigned char isRef, geneType /* 0 V 1 D 2 J 3 C 255 none */, name0, name1, name2, name3;
  unsigned short pad1;
};

// A posWeight column as the kernels see it. Every consumer of _posWeight on this path goes through
// AlignAlgo::IsBaseEqual (AlignAlgo.hpp:49-55): sum == 0 || c == 'N' || sum < 3 * count[c]. So a column is stored as its
// five predicate bits: bit x (0..3) = (sum < 3 * count[x]), bit 4 = (sum == 0). 1 byte / base instead of 16.
typedef unsigned char T4PW;
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
T4PW t4PwByte(int a, int c, int g, int t) {
  const int sum = a + c + g + t;
  return (T4PW)((sum == 0 ? 16 : 0) | (sum < 3 * a ? 1 : 0) | (sum < 3 * c ? 2 : 0) | (sum < 3 * g ? 4 : 0) | (sum < 3 * t ? 8 : 0));
}

struct T4HashEntC {          // slot of a per-barcode image or of a live set: the bucket is implied by the code; code == ~0 = empty
  unsigned long long code;
  unsigned start, cnt;
};

struct T4HashEnt {           // open-addressing slot of the (code, bucket) -> postings map
  unsigned long long code;
  int h;                     // KmerIndex bucket id; -1 = empty slot
  unsigned start, cnt;
  unsigned pad;
};

struct alignas(16) T4IndexView {   // 128 bytes: per-barcode views are scattered in 16-byte units
  int k, nseq, direct /* 0 htab, 1 table, 2 ctab */, considerBarcode;
  unsigned long long hashMask;
  const uint2 *table;        // direct-addressed [4^k] {start,cnt} (k <= 12, no barcode)
  const T4HashEnt *htab;
  const int2 *post;          // postings (idx, offset) -- sizeof(_indexInfo) = 8
  const T4SeqInfo *seqs;
  const char *cons;
  const T4PW *pw;            // posWeight predicate bytes of novel contigs
  const T4HashEntC *ctab;    // direct == 2
  int radius, hitLenRequired, nomatchGapLimit, firstIsRef, hasNovel /* 0 no novel contig, 1 mixed set, 2 novel contigs only */;
  int key32;                 // 0, or idxBits | cBits << 8 when (strand, idx, diagonal, read offset) fit 32 bits (t4Key32Bits)
  double novelSim, refSim, repeatSim;
};

// Small sets (the reference genes; one cell's contigs) sort their hits as 32-bit keys: strand:1 | idx:idxBits | diagonal:cBits |
// read offset:9. The diagonal is biased by 2^cBits - 512 (read offsets are < 512, sequence offsets < maxSeqLen <= 2^cBits - 512);
// for one (strand, idx, diagonal) the read offset orders like the sequence offset, so the order equals that of the 64-bit keys.
// idx never reaches all ones, which keeps 0xFFFFFFFF free for dropped hits. Returns 0 when the set does not fit.
static inline int t4Key32Bits(int nseq, int maxSeqLen) {
  int idxBits = 1, cBits = 10;
  while ((1 << idxBits) <= nseq) ++idxBits;              // nseq + 1 values
  while ((1 << cBits) - 512 < maxSeqLen) ++cBits;
  return 1 + idxBits + cBits + 9 <= 32 ? (idxBits | cBits << 8) : 0;
}

struct T4BatchView {
  const unsigned *pk;        // 2-bit bases, 16 per word, wpk words per read
  const unsigned *nm;        // N mask, 32 bases per word, wnm words per read
  const int *len;
  const int *barcode;        // may be null
  int wpk, wnm;
  long long n;
};

struct T4TierCaps { int cap[T4_NTIER - 1]; };   // hit capacity of the LDS tiers, ascending

// ---- the wide AddRead query: a read whose hits outgrow one workgroup's LDS is spread over the chip ---------------
// The hits of such a read are scattered into PARTITIONS by contig range (every (strand, contig) group of GetOverlapsFromHits lies
// in one partition); a partition is sorted, chained and scored by a workgroup of its own, the steps of GetOverlapsFromRead that
// look across contigs (SeqSet.hpp:784-823 statistics, 1597 sort, 1601-1634 strand, 1705-1794 pre-filters, 2105-2119 cut) are
// replayed over the partitions' records by one workgroup per read.
struct T4Grp { unsigned key; unsigned cnt; int lo, hi; };   // dependency set of a query: key = contig * 2 + (strand == 1); hits; hull of the diagonals with >= 3 hits (lo > hi: none)
#define T4_WIDE_MAXP 2048     // partitions of one read (11 bits of the merge's record index)
struct T4WidePlan { int pBase, P, Wd /* unused since the partitions follow the hits' distribution */, nk; unsigned H; int huge /* a list beyond 10000 postings */, read, grpBase; };
#define T4_WIDE_STAT 24      // ints per read: see wideStatsKernel
struct T4Wide {
  int enabled;
  int maxReads, maxPart, pcap, maxOvPart, safetyNum /* partitions are planned for pcap * 16 / safetyNum hits */, maxPartPerRead;
  int samplePerPart;         // hits sampled per planned partition for the partition boundaries (0: 4096 per read whatever it plans)
  int minHits;               // a read goes wide when its seed stage emits more hits than this (or meets a list beyond 10000 postings, or outgrows the global-scratch tier)
  int *ctl;                  // [0] reads, [1] partitions, [2] overflow flags (1 reads, 2 partitions, 4 keys of a partition, 8 overlaps of a partition, 16 group pool), [3] group pool cursor
  T4WidePlan *plan;          // [maxReads]
  uint2 *seed;               // [maxReads][2 * T4_MAXL]: (start, emitted postings) of every k-mer position, forward strand first
  int *bounds;               // [maxReads][T4_WIDE_MAXP + 1]: first contig of every partition of a read (quantiles of a sample of its hits' contigs)
  unsigned *pCnt;            // [maxPart] keys
  int *pRead;                // [maxPart] slot of the read
  unsigned long long *pKeys; // [maxPart][pcap]
  unsigned short *gSize;     // [maxPart][pcap] sizes of the (strand, contig) groups in key order
  unsigned char *gInfo;      // [maxPart][pcap] huge reads: bits 0-2 = hits of the group whose list holds <= 10000 postings (capped at 4), bit 3 = its hit lowest on the read is one
  int *gCount;               // [maxPart][4]: groups on the minus / plus strand, keys on the minus strand, spare
  int *gOff;                 // [maxPart][2]: position of the partition's first minus / plus group among the read's groups (strand-major)
  int *pRec;                 // [maxPart][maxOvPart] OvRec (10 ints)
  int *pRecCnt;              // [maxPart]
  int *stat;                 // [maxReads][T4_WIDE_STAT]
  unsigned short *uniqPref;  // [maxReads][pcap + 1]: prefix counts of hits with <= 10000 postings over the head of the read's hit array in the reference's order (SeqSet.hpp:934-940)
  unsigned long long *mKeys; // [maxPart * maxOvPart] sort keys of the merge
  int *mOrd;                 // [maxPart * maxOvPart]
  T4Grp *grpPool;            // pinned host memory (device pointer): dependency sets of the wide reads
  int grpCap;
  unsigned long long *sortTmp;  // [maxReads][2 * pcap] scratch of the statistics kernel (huge reads)
};

struct T4Work {              // per-launch work description
  const int *list;           // read ids of this tier
  int nList;
  int *nextList;             // overflow -> next tier
  int *nextCount;
  int *workNext;             // dynamic distribution of the list over the persistent grid (zeroed per launch); null: static stride
  int *status;               // per read: 0 ok, 1 unsupported
  unsigned long long *hitCounter;
  // scratch (per persistent block)
  int *dpRows;               // [grid][6 * (T4_MAXGAP + 2) * 64]
  unsigned char *dpDir;      // [grid][64 * T4_DIR_BYTES]
  // global-tier working arrays (per block), null for LDS tiers
  unsigned long long *gKeys; // [grid][cap]
  unsigned *gPairs;          // [grid][cap]
  unsigned *gCand;           // [grid][cap]
  int *gOv;                  // [grid][maxov * 10]
  int *gFin;                 // [grid][maxov * 10]
  unsigned short *gOrd;      // [grid][maxov]
  int gCap, gMaxOv;
  int capLimit;              // testing aid: an LDS tier pretends its hit capacity is this small (0 = off), so that small inputs reach the overflow paths
  const T4Wide *wide;        // mode 4 of a contig set: reads beyond the LDS tier are handed to the wide query (device copy of its description; null: off)
};

// KmerCount on the device (KmerCount.hpp): open-addressing table of canonical k-mer codes; keys hold code + 1 (0 = empty)
struct T4KmerTable {
  unsigned long long *keys;
  unsigned *cnt;
  unsigned long long mask;   // slots - 1 (slots is a power of two)
  int k;
  int perBarcode;            // 1: the read's barcode is part of the key (bits 42..62; k <= 21) -- one KmerCount per barcode
  int *overflow;             // set when an insert found the table full
  unsigned long long *used;  // occupied slots (one atomic per wavefront pass): the host grows the table before it fills (t4_kmer_count_add)
};

struct T4OverlapOut {        // == t4_overlap of include/trust4_hip.h
  int seqIdx, readStart, readEnd, seqStart, seqEnd, strand, matchCnt, indelCnt;
  double similarity;
};

struct T4HitOut { int idx, offset, readOffset, strand, repeats; };  // == t4_hit

// one candidate overlap of an AddRead query, 24 bytes (see T4QueryArgs::candOut); == t4_cand of t4_internal.h
struct T4Cand {
  int seqIdx, ss, se;
  short rs, re;
  short m0;                  // matchCnt of GetOverlapsFromHits: the sort key of the scan
  short matchCnt;            // scored (SeqSet.hpp:2009)
  short indelCnt;
  unsigned short flags;      // 1: plus strand, 2: scoring left similarity 0, 4: cut by the pre-filters (its scored fields are still the scored ones), bits 3-12: run size
};
#define T4_QSTATS 12         // statistics words per read: [0..7] see T4QueryArgs (stats8), [8..11] restricted re-queries: hull lo (minus, plus), hull hi (minus, plus) of the one contig's group
struct T4CandArgs {
  T4Cand *candOut;
  unsigned *candCursor;
  int *candOverflow;
  int candCap;
  int *candBase, *candCnt;   // null with candOut
  int *stats8;
  const int *forceMin;       // nullable
  int useMarks;              // the image's predicate bytes carry posting marks (T4_PW_MARK_*): restricted re-queries read a contig's postings off the contig
};
// Bits 5-6 of a contig's predicate byte at offset o: the number of postings (contig, o) the index holds (0-3); bit 7 of the byte at
// offset 0: the marks of this contig are not to be trusted (an offset with more than three postings). Written by the ordered
// builder's deltas (t4_assembler::makeDelta); AlignAlgo::IsBaseEqual's bits 0-4 are all baseEqualW looks at.
#define T4_PW_MARK_SHIFT 5
#define T4_PW_MARK_BAD 128
#define T4_CAND_PLUS 1
#define T4_CAND_SIMZERO 2
#define T4_CAND_CUT 4
#define T4_CAND_RUN_SHIFT 3   // bits 3-12: hits of the run the overlap was chained from (what novelMinHitRequired is compared with)

struct T4QueryArgs {
  int mode;                  // 0: overlaps (GetOverlapsFromRead), 1: annotate level 0, 2: AssignRead, 3: ExtendOverlap of given overlaps, 4: AddRead query
  int strand;                // strand argument of GetOverlapsFromRead
  int skipRepeats;
  int maxPerRead;            // mode 0 output stride
  int *counts;               // mode 0
  T4OverlapOut *out;
  // modes 2/3
  const T4OverlapOut *in;    // mode 3: overlaps to extend, maxPerRead per read
  const int *inCounts;       // mode 3
  int *ret;                  // mode 2: AssignRead return value per read; mode 3: ExtendOverlap return value per overlap
  double mismatchFactor;     // mode 3 (mode 2 uses 1.0 / 2.0 by barcode as AssignRead does)
  // mode 4: GetOverlapsFromRead + ExtendOverlap of every returned overlap, per-read strand argument and factor
  const int *strandPerRead;
  const double *factorPerRead;
  T4OverlapOut *outExt;
  // mode 4 results are variable-size: read r's records start at outBase[r] of out / outExt / ret (space taken from a pool by
  // one atomic per read; a read that does not fit the pool gets status 3 and the host repeats the call with a larger one)
  int *outBase;
  unsigned *poolCursor;
  int poolCap;
  // mode 4 of one big set: the ExtendOverlap calls of a read that returned more than `extendLater` overlaps (0: never) run
  // in a launch of their own (extendKernel: a read that overlaps thousands of contigs spreads over the chip instead of
  // occupying one workgroup); the query kernel leaves a device copy of those records and marks every record with its read
  // (-1: extended by the query kernel)
  int extendLater;
  T4OverlapOut *outDev;
  int *recRead;
  int leanExt;               // mode 4: records of overlaps whose extension meets an indel carry exact coordinates and return value only (extendOverlaps)
  int *readTicks;
  int *statsStable;          // per read: 1 when the group statistics of GetOverlapsFromHits (SeqSet.hpp:784-823) cannot move under index edits that leave every group of three or more hits alone (null: not wanted)            // mode 4, nullable: wall-clock ticks (10 ns) one workgroup spent on the read (the latency a dependent round pays)
  // mode 4, nullable: onlySeq[r] >= 0 asks for the overlaps of read r with that ONE contig (all of them, both strands, scored and
  // extended, no filter that looks across contigs): the ordered builder's re-query of a window entry after a commit that touched
  // one contig of the forty it meets (t4_assembler: restricted re-query). aux[r] (full queries): overlaps on the strand of the best
  // one before the similarity cut | overlaps on the other strand << 15 | (that strand is plus) << 30, -1 when the wide query served
  // the read; n4[r]: (strand, contig) groups of four or more hits.
  const int *onlySeq;
  int *aux;
  int *n4;
  // mode 4, nullable -- the candidate store of the ordered builder. cand*: EVERY overlap of the read on the strand of
  // the best one as it stands before the similarity cut (a restricted re-query: every overlap with the one contig, both strands), in
  // the order of the scan of SeqSet.hpp:1673-2094, with its pre-score key, its scored fields and whether the pre-filters of 1705-1794
  // cut it: what a host-side replay of that scan needs after ONE contig's candidates were replaced. Room is taken from a pool in
  // pinned host memory by one atomic per read (candCursor; an overflow raises candOverflow and the host repeats the call).
  // stats8[8 r ..]: per strand (minus, plus) the groups of >= 4 and of >= 5 hits (true sizes), the largest group, and the
  // novelMinHitRequired the pass used (SeqSet.hpp:784-823). forceMin[r] (restricted re-queries): that threshold for the one contig's
  // groups, minus | plus << 16 (0: the pass's own statistics -- 3 for a single contig).
  const struct T4CandArgs *cs;   // (behind a pointer: every word of this struct costs the AddRead kernel private stack, see queryKernel)
  // per-read set images (per-barcode contig sets): read r is matched against views[viewOf[r]]
  const T4IndexView *views;
  const int *viewOf;
};

struct T4BytePatch { unsigned char *dst; unsigned long long val; };   // one posWeight predicate byte of a resident image
struct T4CopyDesc { unsigned long long srcOff; unsigned char *dst; unsigned long long bytes; };  // scatter of staged set images

// trust4_amd/csrc/t4_api.hip -- C ABI of libt4hip.so (include/trust4_hip.h): host side of the engine.
// Sequence-set construction follows SeqSet::InputRefFa / InputNovelRead and
// KmerIndex::BuildIndexFromRead (SeqSet.hpp:2673-2865, 3028-3073; KmerIndex.hpp:118-141); the k-mer
// index is flattened into CSR postings + a direct-addressed (k <= 12) or open-addressing table that
// stays resident in HBM/L2. Queries run as: bin reads by hit count -> one persistent-grid launch per
// capacity tier (reads that outgrow a tier are re-queued on the next one).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#ifdef __HIPCC__
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is bound by dlopen in t4_comm_init
#include <type_traits>
#include <unistd.h>
#endif
#include "../../include/trust4_hip.h"
#include "t4_kernels.h"
#include "t4_internal.h"

static_assert(sizeof(t4_overlap) == sizeof(T4OverlapOut), "overlap layout");
static_assert(sizeof(t4_hit) == sizeof(T4HitOut), "hit layout");
static_assert(sizeof(t4k::OvRec) == 40, "OvRec layout");
static_assert(sizeof(T4IndexView) % 16 == 0, "view stride");

namespace {

// per capacity tier: LDS hit capacity, threads cooperating on one read, resident workgroups per CU
const int TIER_CAP[T4_NTIER - 1] = {1024, 2048, 3072, 4096, 8192};
// Reads in flight per CU are what counts (most phases of a read are latency bound): tier 0 runs 8 groups of 2 waves in
// 8 x 20 KB of LDS, tier 1 4 groups of 4 waves in 4 x 39 KB, both at 128 VGPRs (4 waves / SIMD); the 3072-hit tier exists
// because 3 of its groups fit a CU where only 2 of the 4096-hit tier do. 512/1024 threads per read in the upper tiers
// measured slower (barriers).
const int TIER_THREADS[T4_NTIER] = {128, 256, 256, 256, 512, 512};   // the 8192-hit tier has one group per CU (LDS): eight waves of it
const int TIER_BLOCKS_PER_CU[T4_NTIER] = {8, 4, 3, 2, 1, 2};
// global-scratch tier: hits and overlaps of one read pass (SURVEY 6 measured 50 036 hits for one AssignRead query at k = 17 and
// 100 k pairs; a read of a gene segment that thousands of contigs share meets every one of them)
const int G_CAP = 262144, G_MAXOV = 16384;
constexpr int G_THREADS = 512;

struct HostSeq {
  std::string name, cons;
  std::vector<int32_t> pw;  // 4 per base, novel only
  int barcode;
  bool isRef;
};

}  // namespace

// The call is split in two so that a caller can keep the host busy while the kernels run (t4_assembler commits reads whose cached
// queries still stand): aqBegin packs the input and enqueues copies + kernels + the header copy on the ctx's stream; aqEnd waits
// for them, runs whatever overflow launch is still needed (rare paths, synchronous) and hands out the result. One call in flight
// per ctx. tierHint must stay alive until aqEnd.
struct AqCall {
  bool active = false;
  T4IndexView base; const T4IndexView *views = nullptr; bool hasViewOf = false, smallFirst = false, lean = false;
  int n = 0, skipRepeats = 0, wpk = 0, wnm = 0, attempt = 0, nFirst = 0, nDirect = 0, threads = 512;
  unsigned char *tierHint = nullptr;
  std::vector<unsigned char> allGlobal;
  size_t oPk, oNm, oLen, oBc, oSt, oLs, oVw, oFa, oOnly, oForce, oCs, oWide, oWideA, inBytes, pCb, pCc, pS8, pCnt, pSta, pNext, pNext2, pBase, pTick, pStab, pAux, pN4, pTail, pWctl, pWplan, pWstat, pWctlA, pWplanA, pWstatA, outBytes;
  bool hasOnly = false, hasForce = false, wantCands = false, useMarks = false;
  bool extendLater = false, wide = false, onlyRestricted = false;
  bool lazyDone = false, lazyStage = false;   // the wide pipeline behind the query kernel is launched only once the kernel is known to have deferred a read (aqEnd)
  int wideSafety = 32;   // of sixteenths: partitions are planned for half of their capacity
  T4BatchView bv; T4QueryArgs qa; T4Work wk;
  std::chrono::steady_clock::time_point tf0;
};

struct t4_ctx {
  int device = 0, cus = 0;
  hipStream_t stream = 0;
  hipStream_t stream2 = 0;   // AddRead queries: reads known to need the global-scratch tier run beside the LDS tier
  hipEvent_t evIn = 0, evG = 0;
  hipEvent_t ev[4] = {0, 0, 0, 0};
  std::string err;
  t4_stats stats;
  // scratch
  int maxGrid = 0;
  int *dpRows = nullptr;
  unsigned char *dpDir = nullptr;
  unsigned long long *gKeys = nullptr;
  unsigned *gPairs = nullptr, *gCand = nullptr;
  int *gOv = nullptr, *gFin = nullptr;
  unsigned short *gOrd = nullptr;
  int gGrid = 0;
  unsigned long long *hitsKeys = nullptr;
  int hitsGrid = 0;
  // per-call buffers (grown on demand)
  int *lists = nullptr, *listCounts = nullptr, *status = nullptr, *counts = nullptr;
  long long listCap = 0;
  unsigned long long *hitCounter = nullptr;
  T4OverlapOut *result = nullptr;
  size_t resultCap = 0, resultBytes = 0;
  // lean path of t4_add_query (small batches, one launch, one round trip)
  int aqCap = 0, aqWpk = 0, aqWnm = 0, aqMax = 0;
  unsigned char *aqIn = nullptr, *aqOut = nullptr;      // device blobs
  unsigned char *aqInHost = nullptr, *aqOutHost = nullptr;   // pinned staging, mapped: aqPrologueKernel reads the one, aqEpilogueKernel writes the other
  unsigned char *aqInHostDev = nullptr, *aqOutHostDev = nullptr;   // their device addresses
  unsigned *aqFlagHost = nullptr, *aqFlagDev = nullptr;   // pinned word the epilogue of a call leaves the call's sequence number in: what the host waits for
  unsigned *aqDoneCtr = nullptr;   // device: blocks of a running epilogue that have written their share
  unsigned aqSeq = 0;
  size_t aqInBytes = 0, aqOutBytes = 0;
  unsigned char *aqPool = nullptr, *aqPoolDev = nullptr;   // result records of t4_add_query*: pinned host memory the kernels write
  int aqPoolCap = 0;
  unsigned char *candPool = nullptr; T4Cand *candPoolDev = nullptr;   // candidate store of t4_add_query_pool_begin2 (pinned host memory the kernels write)
  int candCap = 0, candGrows = 0;
  int64_t candRecords = 0;
  T4OverlapOut *aqRecDev = nullptr;   // device copy of the overlap records + the read of each, for the extension launch
  int *aqRecRead = nullptr;
  int aqRecCap = 0;
  int64_t aqCalls = 0, aqReads = 0, aqGlobalLaunches = 0, aqGlobalReads = 0, aqRecords = 0;
  double aqSecPack = 0, aqSecFirst = 0, aqSecGlobal = 0;
  double aqSecLaunch[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // host seconds inside aqLaunch: preparation, prologue, second stream, query kernel, wide + extension, events, epilogue
  int aqPoolGrows = 0;
  const int32_t *aqLastStable = nullptr;   // per-read flags of the last AddRead query call: group statistics that index edits of small groups cannot move
  const int32_t *aqLastTicks = nullptr; int aqLastN = 0;   // per-read wall-clock ticks (10 ns) of the last AddRead query call (in the pinned header blob)
  double aqLastMs = 0;
  AqCall aq;
  uint64_t syncEpoch = 1;   // bumped whenever the ctx's stream has been waited for: work queued before that is done (t4_index_apply_delta's staging buffer)
  // testing aids of the AddRead query path, read from the environment once per ctx (a query round is a few hundred microseconds; a
  // dozen getenv calls in it are not nothing)
  struct AqEnv {
    bool forceGlobal, wideNoHint, wideEager;
    int capLimit, extendDefer, poolCap, candCap, wideMinHits, wideSample;
    AqEnv() {
      auto num = [](const char *n, int d) { const char *e = getenv(n); return e ? atoi(e) : d; };
      forceGlobal = getenv("T4_AQ_FORCE_GLOBAL") != nullptr; wideNoHint = getenv("T4_WIDE_NO_HINT") != nullptr;
      wideSample = num("T4_WIDE_SAMPLE", 256);   // hits sampled per planned partition for the partition boundaries of a wide read (0: 4 096 per read, the rule until round 6 -- a read that plans four partitions does not need them: kernels of C2 23.9 -> 23.3 s, no partition overflowed, profiles/r06g)
      wideEager = getenv("T4_WIDE_EAGER") != nullptr;   // A/B aid: the five wide kernels behind every whole-query round's query kernel, as until round 6
      capLimit = num("T4_AQ_CAP_LIMIT", 0); poolCap = num("T4_AQ_POOL_CAP", 0); candCap = num("T4_AQ_CAND_CAP", 1 << 18);
      // (64 until round 5: with light rounds extendKernel runs behind the whole-query rounds anyway, and a read's 17th overlap is better
      // off there -- profiles/r05e, r05f)
      extendDefer = num("T4_AQ_EXTEND_DEFER", 16);
      wideMinHits = num("T4_WIDE_MIN_HITS", 3072);   // (round 6: 3 072 -- kernels of C2 23.9 -> 23.6 s, 2 % fewer rounds, profiles/r06g; 4 096 in round 5:)   // (8192, the LDS tier's capacity, until round 5: fresh heavy reads now start on the wide pipeline beside the query kernel, so the wide query pays from half of it on -- C2 67.5 -> 61.0 s, profiles/r05e_c2_w4k, r05f)
    }
  } aqEnv;
  // the wide query (t4_wide.h): pools of the deferred reads of one call, grown on demand
  T4Wide wide;               // device pointers + capacities (a copy travels in every call's input blob)
  bool wideInit = false;
  unsigned char *grpPoolHost = nullptr;   // pinned; T4Wide::grpPool is its device address
  int64_t wideReads = 0, wideParts = 0, wideRetries = 0, wideGroups = 0;
  int64_t wideCalls = 0, wideCallsDeferred = 0, wideCallsDirect = 0;   // calls with the wide query on; those whose query kernel deferred a read; those with reads on the second stream
  int wideSafetyKeep = 32, wideCallsSinceRepeat = 0;   // partition load factor that recent calls needed (of sixteenths: 32 = partitions planned half full); decays back when nothing overflows
  int64_t wideFlagCounts[6] = {0, 0, 0, 0, 0, 0};     // calls repeated because: reads, partitions, keys of a partition, overlaps of a partition, dependency records, other
  int wideRecentParts = 0, wideRecentReads = 0;   // the largest counts of the last calls, decayed: sizes the (persistent) grids of the next call's wide kernels
  double aqKernelMs = 0;    // HIP-event time of the query kernels of all AddRead query calls (per call: first launch .. last kernel)
  int64_t aqHits = 0;       // _hit records their seed stages emitted (H of SURVEY 8d)
};

struct t4_index {
  t4_ctx *ctx;
  int k, considerBarcode;
  int hitLenRequired = 31, radius = 10;
  double novelSim = 0.9, refSim = 0.75, repeatSim = 0.95;
  int nomatchGapLimit;
  std::vector<HostSeq> seqs;
  std::unordered_map<std::string, int> dedup;
  bool committed = false;
  // device image
  uint2 *dTable = nullptr;
  T4HashEnt *dHtab = nullptr;
  int2 *dPost = nullptr;
  T4SeqInfo *dSeqs = nullptr;
  char *dCons = nullptr;
  T4PW *dPw = nullptr;
  T4IndexView view;
  // live set (t4_index_apply_delta): capacities of the four arrays, staging of one delta
  bool live = false;
  T4HashEntC *dCtab = nullptr;
  int64_t capTable = 0, capPost = 0, capBase = 0;
  int capSeq = 0, liveNseq = 0;
  unsigned char *stHost = nullptr, *stDev = nullptr, *stHostDev = nullptr;   // (stHostDev: the pinned staging buffer as the device addresses it)
  size_t stCap = 0;
  hipEvent_t stEvent = 0;
  bool stPending = false;
  uint64_t stEpoch = 0;      // the ctx's syncEpoch when the staging buffer was last handed to a copy
};

struct t4_batch {
  t4_ctx *ctx;
  long long n = 0;
  int wpk = 0, wnm = 0, maxLen = 0;
  unsigned *dPk = nullptr, *dNm = nullptr;
  int *dLen = nullptr, *dBarcode = nullptr;
  T4BatchView view;
};

namespace {

int fail(t4_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) return fail((ctx), T4_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

template <class T> int devAlloc(t4_ctx *c, T **p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPCHK(c, hipMalloc(p, sizeof(T) * (count ? count : 1)));
  return T4_OK;
}

// SeqSet::GetChainType / GetGeneType (SeqSet.hpp:5132-5155, 5076-5100)
int chainType(const char *n) {
  if (n[0] == 'I') { if (n[2] == 'H') return 0; if (n[2] == 'K') return 1; if (n[2] == 'L') return 2; }
  else if (n[0] == 'T') { if (n[2] == 'A') return 3; if (n[2] == 'B') return 4; if (n[2] == 'G') return 5; if (n[2] == 'D') return 6; }
  return 8;
}
int geneType(const char *n) {
  if (n[0] == 'N' && n[1] == 'o') return -1;
  switch (n[3]) {
    case 'V': return 0;
    case 'D': return (n[4] >= '0' && n[4] <= '9') ? 1 : 3;
    case 'J': return 2;
    case 'L': if (chainType(n) == 2) return -1; return 3;
    default: return 3;
  }
}
inline int nucNum(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

// gap-DP scratch for `threads` resident threads (fallback path of bands wider than a wavefront)
int ensureScratch(t4_ctx *c, int threads) {
  if (threads <= c->maxGrid) return T4_OK;
  int r;
  if ((r = devAlloc(c, &c->dpRows, (size_t)(threads / 64 + 1) * 6 * T4_ROWW * 64))) return r;
  if ((r = devAlloc(c, &c->dpDir, (size_t)threads * T4_DIR_BYTES))) return r;
  c->maxGrid = threads;
  return T4_OK;
}
int ensureGlobalTier(t4_ctx *c, int grid) {
  if (grid <= c->gGrid) return T4_OK;
  { int g = c->gGrid > 0 ? c->gGrid : 32; while (g < grid) g *= 2; grid = g; }   // 5.4 MB per block: grow rarely
  int r;
  if ((r = devAlloc(c, &c->gKeys, (size_t)grid * G_CAP))) return r;
  if ((r = devAlloc(c, &c->gPairs, (size_t)grid * G_CAP * 2))) return r;   // pairs + cand, contiguous per block
  if ((r = devAlloc(c, &c->gOv, (size_t)grid * G_MAXOV * 10))) return r;
  if ((r = devAlloc(c, &c->gFin, (size_t)grid * G_MAXOV * 10))) return r;
  if ((r = devAlloc(c, &c->gOrd, (size_t)grid * G_MAXOV))) return r;
  c->gGrid = grid;
  return T4_OK;
}
int ensurePerCall(t4_ctx *c, long long n) {
  if (n <= c->listCap) return T4_OK;
  int r;
  if ((r = devAlloc(c, &c->lists, (size_t)n * T4_NTIER))) return r;
  if ((r = devAlloc(c, &c->status, (size_t)n))) return r;
  if ((r = devAlloc(c, &c->counts, (size_t)n))) return r;
  c->listCap = n;
  return T4_OK;
}


// pools of the wide query (t4_wide.h): room for `reads` deferred reads, `parts` partitions of `pcap` keys and `groups` dependency
// records per call; existing contents are never needed across calls, so growing reallocates
int wideEnv(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
int ensureWide(t4_ctx *c, int reads, int parts, int groups) {
  T4Wide &w = c->wide;
  if (!c->wideInit) {
    memset(&w, 0, sizeof w);
    w.pcap = wideEnv("T4_WIDE_PCAP", 8192);   // testing aid: small partitions, so that small inputs spread over several
    if (w.pcap < 64) w.pcap = 64;
    if (w.pcap > 8192) w.pcap = 8192;
    w.maxOvPart = 1 << T4_WIDE_OVBITS;
    w.maxPartPerRead = T4_WIDE_MAXP;
    c->wideInit = true;
  }
  int r;
  if (reads > w.maxReads) {
    int n = w.maxReads > 0 ? w.maxReads : 64;
    while (n < reads) n *= 2;
    if ((r = devAlloc(c, &w.seed, 2 * (size_t)n * T4_WIDE_SEEDS))) return r;   // (two of everything: wideHalf)
    if ((r = devAlloc(c, &w.bounds, 2 * (size_t)n * (T4_WIDE_MAXP + 1)))) return r;
    if ((r = devAlloc(c, &w.uniqPref, 2 * (size_t)n * (w.pcap + 1)))) return r;
    if ((r = devAlloc(c, &w.sortTmp, 2 * (size_t)n * 2 * w.pcap))) return r;
    w.maxReads = n;
  }
  if (parts > w.maxPart) {
    int n = w.maxPart > 0 ? w.maxPart : wideEnv("T4_WIDE_PARTS", 1024);
    while (n < parts) n *= 2;
    const size_t n2 = 2 * (size_t)n;
    if ((r = devAlloc(c, &w.pCnt, n2))) return r;
    if ((r = devAlloc(c, &w.pRead, n2))) return r;
    if ((r = devAlloc(c, &w.pKeys, n2 * w.pcap))) return r;
    if ((r = devAlloc(c, &w.gSize, n2 * w.pcap))) return r;
    if ((r = devAlloc(c, &w.gInfo, n2 * w.pcap))) return r;
    if ((r = devAlloc(c, &w.gCount, n2 * 4))) return r;
    if ((r = devAlloc(c, &w.gOff, n2 * 2))) return r;
    if ((r = devAlloc(c, &w.pRec, n2 * w.maxOvPart * 10))) return r;
    if ((r = devAlloc(c, &w.pRecCnt, n2))) return r;
    if ((r = devAlloc(c, &w.mKeys, n2 * w.maxOvPart))) return r;
    if ((r = devAlloc(c, &w.mOrd, n2 * w.maxOvPart))) return r;
    w.maxPart = n;
  }
  if (groups > w.grpCap) {
    int n = w.grpCap > 0 ? w.grpCap : wideEnv("T4_WIDE_GROUPS", 1 << 20);
    while (n < groups) n *= 2;
    if (c->grpPoolHost) (void)hipHostFree(c->grpPoolHost);
    c->grpPoolHost = nullptr; w.grpPool = nullptr;
    HIPCHK(c, hipHostMalloc(&c->grpPoolHost, sizeof(T4Grp) * 2 * (size_t)n, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **)&w.grpPool, c->grpPoolHost, 0));
    w.grpCap = n;
  }
  return T4_OK;
}
// The pools hold two pipelines' worth: half 0 serves the reads the round's query kernel defers, half 1 the reads that were known to
// be heavy and started on the second stream (wideSeedKernel).
T4Wide wideHalf(const t4_ctx *c, int half) {
  T4Wide w = c->wide;
  if (!half) return w;
  const size_t R = (size_t)w.maxReads, P = (size_t)w.maxPart;
  w.seed += R * T4_WIDE_SEEDS; w.bounds += R * (T4_WIDE_MAXP + 1); w.uniqPref += R * (w.pcap + 1); w.sortTmp += R * 2 * w.pcap;
  w.pCnt += P; w.pRead += P; w.pKeys += P * w.pcap; w.gSize += P * w.pcap; w.gInfo += P * w.pcap; w.gCount += P * 4; w.gOff += P * 2;
  w.pRec += P * w.maxOvPart * 10; w.pRecCnt += P; w.mKeys += P * w.maxOvPart; w.mOrd += P * w.maxOvPart;
  w.grpPool += w.grpCap;
  return w;
}

}  // namespace

extern "C" {

int t4_init(int device_ordinal, t4_ctx **out) {
  if (!out) return T4_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_ordinal < 0 || device_ordinal >= ndev) return T4_ERR_HIP;
  if (hipSetDevice(device_ordinal) != hipSuccess) return T4_ERR_HIP;
  t4_ctx *c = new t4_ctx();
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) { delete c; return T4_ERR_HIP; }
  c->cus = prop.multiProcessorCount;
  if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return T4_ERR_HIP; }
  for (int i = 0; i < 4; ++i) if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return T4_ERR_HIP; }
  memset(&c->stats, 0, sizeof c->stats);
  if (hipMalloc(&c->listCounts, sizeof(int) * 16) != hipSuccess || hipMalloc(&c->hitCounter, sizeof(unsigned long long)) != hipSuccess) { delete c; return T4_ERR_HIP; }
  if (hipHostMalloc(&c->aqFlagHost, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void **)&c->aqFlagDev, c->aqFlagHost, 0) != hipSuccess ||
      hipMalloc(&c->aqDoneCtr, sizeof(unsigned)) != hipSuccess || hipMemset(c->aqDoneCtr, 0, sizeof(unsigned)) != hipSuccess) { delete c; return T4_ERR_HIP; }
  *c->aqFlagHost = 0;
  *out = c;
  return T4_OK;
}

void t4_destroy(t4_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
#ifdef T4_PHASE_TIMING
  if (getenv("T4_PHASE_DUMP")) {   // development aid: cycles per kernel phase over the life of the ctx
    unsigned long long ph[T4_NPHASE];
    if (hipMemcpyFromSymbol(ph, HIP_SYMBOL(t4k::g_phaseCycles), sizeof(ph)) == hipSuccess) {
      static const char *names[32] = {"other", "seed:lookup", "expand", "sort", "stats", "runs", "bigsort", "chain", "ovsort", "score", "prefilter", "final", "annotate", "score:quick", "score:banded", "score:finish", "extend:ungapped", "after-extend", "extend:list", "extend:dp4", "extend:dp1", "extend:combine", "seed:replay", "seed:scan", "extend:launch", "-", "-", "-", "-", "-", "-", "-"};
      unsigned long long tot = 0;
      for (int i = 0; i < T4_NPHASE; ++i) tot += ph[i];
      for (int i = 0; i < T4_NPHASE; ++i) if (ph[i]) fprintf(stderr, "phase %-16s %-7s %6.2f%%  %.3e cycles\n", names[i & 31], i < 32 ? "lds" : "global", 100.0 * (double)ph[i] / (double)tot, (double)ph[i]);
    }
    static unsigned long long cls[4 * T4_NPHASE];
    if (hipMemcpyFromSymbol(cls, HIP_SYMBOL(t4k::g_phaseByOverlaps), sizeof(cls)) == hipSuccess) {
      static const char *names[32] = {"other", "seed:lookup", "expand", "sort", "stats", "runs", "bigsort", "chain", "ovsort", "score", "prefilter", "final", "annotate", "score:quick", "score:banded", "score:finish", "extend:ungapped", "after-extend", "extend:list", "extend:dp4", "extend:dp1", "extend:combine", "seed:replay", "seed:scan", "extend:launch", "-", "-", "-", "-", "-", "-", "-"};
      static const char *cname[4] = {"<5 overlaps", "5-19 overlaps", "20-64 overlaps", ">64 overlaps"};
      for (int k = 0; k < 4; ++k) {
        unsigned long long tot = 0;
        for (int i = 0; i < T4_NPHASE; ++i) tot += cls[k * T4_NPHASE + i];
        if (!tot) continue;
        fprintf(stderr, "AddRead queries with %s: %.3e cycles in the query kernel\n", cname[k], (double)tot);
        for (int i = 0; i < T4_NPHASE; ++i) if (cls[k * T4_NPHASE + i] * 200 >= tot) fprintf(stderr, "   %-16s %-7s %6.2f%%\n", names[i & 31], i < 32 ? "lds" : "global", 100.0 * (double)cls[k * T4_NPHASE + i] / (double)tot);
      }
    }
    unsigned long long dc[8];
    if (hipMemcpyFromSymbol(dc, HIP_SYMBOL(t4k::g_dbgCount), sizeof(dc)) == hipSuccess)
      fprintf(stderr, "debug counters: gap jobs %llu, banded %llu, wave-DP steps %llu, scratch fallbacks %llu; overhang DPs %llu, of which %llu leave the diagonal\n", dc[0], dc[1], dc[2], dc[3], dc[6], dc[7]);
  }
#endif
  void *ptrs[] = {c->dpRows, c->dpDir, c->gKeys, c->gPairs, c->gCand, c->gOv, c->gFin, c->gOrd, c->hitsKeys, c->lists,
                  c->listCounts, c->status, c->counts, c->hitCounter, c->result, c->aqIn, c->aqOut};
  if (c->aqInHost) (void)hipHostFree(c->aqInHost);
  if (c->aqOutHost) (void)hipHostFree(c->aqOutHost);
  if (c->aqFlagHost) (void)hipHostFree(c->aqFlagHost);
  if (c->aqDoneCtr) (void)hipFree(c->aqDoneCtr);
  if (c->aqPool) (void)hipHostFree(c->aqPool);
  if (c->candPool) (void)hipHostFree(c->candPool);
  if (c->aqRecDev) (void)hipFree(c->aqRecDev);
  if (c->aqRecRead) (void)hipFree(c->aqRecRead);
  if (c->wideInit) {
    void *wp[] = {c->wide.seed, c->wide.bounds, c->wide.pCnt, c->wide.pRead, c->wide.pKeys, c->wide.gSize, c->wide.gInfo, c->wide.gCount, c->wide.gOff, c->wide.pRec,
                  c->wide.pRecCnt, c->wide.uniqPref, c->wide.mKeys, c->wide.mOrd, c->wide.sortTmp};
    for (void *p : wp) if (p) (void)hipFree(p);
    if (c->grpPoolHost) (void)hipHostFree(c->grpPoolHost);
  }
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (int i = 0; i < 4; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->evIn) (void)hipEventDestroy(c->evIn);
  if (c->evG) (void)hipEventDestroy(c->evG);
  delete c;
}

int t4_sync(t4_ctx *c) {
  if (!c) return T4_ERR_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return T4_OK;
}
const char *t4_last_error(t4_ctx *c) { return c ? c->err.c_str() : "null ctx"; }
int t4_device_cus(t4_ctx *c) { return c ? c->cus : 0; }
int t4_ctx_device(t4_ctx *c) { return c ? c->device : 0; }
#ifdef T4_PHASE_TIMING
// development aid: cycles spent per kernel phase (summed over workgroups) since the last call
int t4_debug_phase_cycles(unsigned long long *out16) {   // T4_NPHASE entries
  unsigned long long zero[T4_NPHASE] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(t4k::g_phaseCycles), sizeof(zero)) != hipSuccess) return T4_ERR_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(t4k::g_phaseCycles), zero, sizeof(zero)) != hipSuccess) return T4_ERR_HIP;
  return T4_OK;
}
int t4_debug_counters(unsigned long long *out8) {
  unsigned long long zero[8] = {0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(t4k::g_dbgCount), sizeof(zero)) != hipSuccess) return T4_ERR_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(t4k::g_dbgCount), zero, sizeof(zero)) != hipSuccess) return T4_ERR_HIP;
  return T4_OK;
}
#endif
// development aid (T4_PHASE_TIMING builds): forget the phase cycles counted so far; a no-op otherwise
int t4_debug_phase_reset(void) {
#ifdef T4_PHASE_TIMING
  unsigned long long zero[T4_NPHASE] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(t4k::g_phaseCycles), zero, sizeof(zero)) != hipSuccess) return T4_ERR_HIP;
#endif
  return T4_OK;
}
int t4_last_stats(t4_ctx *c, t4_stats *out) {
  if (!c || !out) return T4_ERR_ARG;
  *out = c->stats;
  return T4_OK;
}

// ---- index -------------------------------------------------------------------------------------
int t4_index_create(t4_ctx *c, int k, int consider_barcode, t4_index **out) {
  if (!c || !out) return T4_ERR_ARG;
  if (k < 2 || k > 31) return fail(c, T4_ERR_ARG, "kmer_length %d outside [2,31]", k);
  t4_index *ix = new t4_index();
  ix->ctx = c; ix->k = k; ix->considerBarcode = consider_barcode ? 1 : 0;
  double kmerHitProb = pow(0.8, k);  // SeqSet::ComputeNomatchGapLimit (SeqSet.hpp:2476-2482)
  ix->nomatchGapLimit = int(k * (log(0.01) / log(1 - kmerHitProb))) + 1;
  *out = ix;
  return T4_OK;
}

void t4_index_destroy(t4_index *ix) {
  if (!ix) return;
  if (ix->live) (void)hipStreamSynchronize(ix->ctx->stream);
  void *ptrs[] = {ix->dTable, ix->dHtab, ix->dPost, ix->dSeqs, ix->dCons, ix->dPw, ix->dCtab, ix->stDev};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (ix->stHost) (void)hipHostFree(ix->stHost);
  if (ix->stEvent) (void)hipEventDestroy(ix->stEvent);
  delete ix;
}

int t4_index_set_params(t4_index *ix, int hit_len_required, int radius, double novel_sim) {
  if (!ix) return T4_ERR_ARG;
  ix->hitLenRequired = hit_len_required; ix->radius = radius; ix->novelSim = novel_sim;
  if (ix->committed) { ix->view.hitLenRequired = hit_len_required; ix->view.radius = radius; ix->view.novelSim = novel_sim; }
  return T4_OK;
}

int t4_index_add_ref_record(t4_index *ix, const char *id, const char *seq, int *seq_id) {
  if (!ix || !id || !seq) return T4_ERR_ARG;
  if (ix->committed) return fail(ix->ctx, T4_ERR_STATE, "index already committed");
  if (seq_id) *seq_id = -1;
  if (strlen(id) < 4) return fail(ix->ctx, T4_ERR_ARG, "gene name '%s' shorter than 4 characters", id);
  if (geneType(id) != 1) {  // drop "/OR" orphon genes unless they are D genes
    for (const char *p = id; *p; ++p)
      if (p[0] == '/' && p[1] == 'O' && p[2] == 'R') return T4_OK;
  }
  std::string cons;
  for (const char *p = seq; *p; ++p) {
    if (*p == '.') continue;
    int c = (signed char)*p;
    if (c >= 'a' && c <= 'z') c = (signed char)(c - ('a' + 'A'));  // reference arithmetic: lower case becomes N
    if (c >= 'A' && c <= 'Z') { if (nucNum((char)c) == -1 && c != 'N') c = 'N'; }
    else c = 'N';
    cons.push_back((char)c);
  }
  auto it = ix->dedup.find(cons);
  if (it != ix->dedup.end()) {
    HostSeq &e = ix->seqs[it->second];
    if (e.name.find(id) == std::string::npos) e.name += std::string("|") + id;
    return T4_OK;
  }
  if ((int)ix->seqs.size() >= T4_MAX_SEQS || (int)cons.size() >= T4_MAX_SEQLEN) return fail(ix->ctx, T4_ERR_UNSUPPORTED, "sequence set too large");
  HostSeq hs;
  hs.name = id; hs.cons = cons; hs.barcode = -1; hs.isRef = true;
  ix->dedup[cons] = (int)ix->seqs.size();
  if (seq_id) *seq_id = (int)ix->seqs.size();
  ix->seqs.push_back(hs);
  return T4_OK;
}

int t4_index_load_ref_fasta(t4_index *ix, const char *path) {
  if (!ix || !path) return T4_ERR_ARG;
  gzFile fp = gzopen(path, "rb");
  if (!fp) return fail(ix->ctx, T4_ERR_IO, "cannot open %s", path);
  std::string id, seq;
  bool have = false;
  std::vector<char> line(1 << 20);
  int rc = T4_OK;
  auto flush = [&]() { if (have && rc == T4_OK) rc = t4_index_add_ref_record(ix, id.c_str(), seq.c_str(), nullptr); };
  while (gzgets(fp, line.data(), (int)line.size())) {
    size_t l = strlen(line.data());
    while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
    if (line[0] == '>') {
      flush();
      size_t i = 1;
      while (line[i] && line[i] != ' ' && line[i] != '\t') ++i;
      id.assign(line.data() + 1, i - 1);
      size_t n = id.size();  // ReadFiles.hpp:180-185 strips a trailing /1 or /2
      if (n >= 2 && (id[n - 1] == '1' || id[n - 1] == '2') && id[n - 2] == '/') id.resize(n - 2);
      seq.clear(); have = true;
    } else if (have) seq.append(line.data(), l);
  }
  flush();
  gzclose(fp);
  return rc;
}

int t4_index_add_contig(t4_index *ix, const char *name, const char *consensus, int barcode, const int32_t *posweight, int *seq_id) {
  if (!ix || !name || !consensus) return T4_ERR_ARG;
  if (ix->committed) return fail(ix->ctx, T4_ERR_STATE, "index already committed");
  HostSeq hs;
  hs.name = name; hs.cons = consensus; hs.barcode = barcode; hs.isRef = false;
  size_t len = hs.cons.size();
  if ((int)ix->seqs.size() >= T4_MAX_SEQS || (int)len >= T4_MAX_SEQLEN) return fail(ix->ctx, T4_ERR_UNSUPPORTED, "sequence set too large");
  for (char ch : hs.cons) if (nucNum(ch) < 0 && ch != 'N') return fail(ix->ctx, T4_ERR_UNSUPPORTED, "contig alphabet must be ACGTN");
  hs.pw.assign(4 * len, 0);
  if (posweight) memcpy(hs.pw.data(), posweight, sizeof(int32_t) * 4 * len);
  else for (size_t i = 0; i < len; ++i) if (hs.cons[i] != 'N') hs.pw[4 * i + nucNum(hs.cons[i])] = 1;
  if (seq_id) *seq_id = (int)ix->seqs.size();
  ix->seqs.push_back(std::move(hs));
  return T4_OK;
}

}  // extern "C"

namespace {
struct Rec { unsigned long long code; int h; int idx, off; };
int commitWithPostings(t4_index *ix, std::vector<Rec> &recs);
}  // namespace

extern "C" {

int t4_index_clear(t4_index *ix) {
  if (!ix) return T4_ERR_ARG;
  ix->seqs.clear(); ix->dedup.clear(); ix->committed = false;
  return T4_OK;
}

int t4_index_commit_postings(t4_index *ix, int64_t n, const uint64_t *code, const int32_t *bucket, const int32_t *idx, const int32_t *offset) {
  if (!ix || n < 0 || (n > 0 && (!code || !bucket || !idx || !offset))) return T4_ERR_ARG;
  std::vector<Rec> recs((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (idx[i] < 0 || idx[i] >= (int)ix->seqs.size()) return fail(ix->ctx, T4_ERR_ARG, "posting %lld names sequence %d", (long long)i, idx[i]);
    recs[(size_t)i] = {code[i], bucket[i], idx[i], offset[i]};
  }
  return commitWithPostings(ix, recs);
}

int t4_index_commit(t4_index *ix) {
  if (!ix) return T4_ERR_ARG;
  const int K = ix->k;
  const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1ull) : ~0ull;
  // postings in BuildIndexFromRead order (KmerIndex.hpp:118-141), keyed by (code, bucket)
  std::vector<Rec> recs;
  for (size_t id = 0; id < ix->seqs.size(); ++id) {
    const HostSeq &s = ix->seqs[id];
    int len = (int)s.cons.size();
    if (len < K) continue;
    unsigned long long code = 0, prev = 0;
    int invalidPos = -1;
    for (int i = 0; i < len; ++i) {
      char ch = s.cons[i];
      if (invalidPos != -1) ++invalidPos;
      code = ((code << 2) & mask) | (unsigned long long)(nucNum(ch) & 3);
      if (ch == 'N') invalidPos = 0;
      if (invalidPos >= K) invalidPos = -1;
      if (i < K - 1) continue;
      if (invalidPos == -1 && (i == K || code != prev)) {
        int h = (int)((code + (unsigned long long)(long long)(ix->considerBarcode ? s.barcode + 1 : 0)) % 1000003ull);
        recs.push_back({code, h, (int)id, i - K + 1});
      }
      prev = code;
    }
  }
  return commitWithPostings(ix, recs);
}

}  // extern "C"

namespace {
int commitWithPostings(t4_index *ix, std::vector<Rec> &recs) {
  t4_ctx *c = ix->ctx;
  (void)hipSetDevice(c->device);
  const int K = ix->k;
  std::stable_sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.code != b.code ? a.code < b.code : a.h < b.h; });
  std::vector<int2> post(recs.size());
  for (size_t i = 0; i < recs.size(); ++i) post[i] = make_int2(recs[i].idx, recs[i].off);
  const bool direct = (K <= 12 && !ix->considerBarcode);
  std::vector<uint2> table;
  std::vector<T4HashEnt> htab;
  unsigned long long hashMask = 0;
  if (direct) {
    table.assign((size_t)1 << (2 * K), make_uint2(0, 0));
    for (size_t i = 0; i < recs.size();) {
      size_t j = i;
      while (j < recs.size() && recs[j].code == recs[i].code) ++j;
      table[recs[i].code] = make_uint2((unsigned)i, (unsigned)(j - i));
      i = j;
    }
  } else {
    size_t nkeys = 0;
    for (size_t i = 0; i < recs.size();) { size_t j = i; while (j < recs.size() && recs[j].code == recs[i].code && recs[j].h == recs[i].h) ++j; ++nkeys; i = j; }
    size_t sz = 1024;
    while (sz < 2 * nkeys + 2) sz <<= 1;
    hashMask = sz - 1;
    T4HashEnt empty; empty.code = 0; empty.h = -1; empty.start = 0; empty.cnt = 0; empty.pad = 0;
    htab.assign(sz, empty);
    for (size_t i = 0; i < recs.size();) {
      size_t j = i;
      while (j < recs.size() && recs[j].code == recs[i].code && recs[j].h == recs[i].h) ++j;
      unsigned long long s = t4k::mix64(recs[i].code * 1000003ull + (unsigned long long)recs[i].h) & hashMask;
      while (htab[s].h >= 0) s = (s + 1) & hashMask;
      htab[s].code = recs[i].code; htab[s].h = recs[i].h; htab[s].start = (unsigned)i; htab[s].cnt = (unsigned)(j - i);
      i = j;
    }
  }
  // sequence table
  std::vector<T4SeqInfo> infos(ix->seqs.size());
  std::string cons;
  std::vector<T4PW> pw;
  bool hasNovel = false, hasRef = false;
  for (size_t id = 0; id < ix->seqs.size(); ++id) {
    const HostSeq &s = ix->seqs[id];
    T4SeqInfo &f = infos[id];
    memset(&f, 0, sizeof f);
    f.consOff = (int)cons.size(); f.len = (int)s.cons.size(); f.barcode = s.barcode; f.isRef = s.isRef ? 1 : 0;
    cons += s.cons; cons.push_back('\0');
    char nm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    strncpy(nm, s.name.c_str(), 7);
    int gt = geneType(nm);
    f.geneType = gt < 0 ? 255 : (unsigned char)gt;
    f.name0 = (unsigned char)nm[0]; f.name1 = (unsigned char)nm[1]; f.name2 = (unsigned char)nm[2]; f.name3 = (unsigned char)nm[3];
    f.pwOff = -1;
    if (!s.isRef) {
      hasNovel = true;
      f.pwOff = (int)pw.size();
      for (int i = 0; i < f.len; ++i) pw.push_back(t4PwByte(s.pw[4 * i], s.pw[4 * i + 1], s.pw[4 * i + 2], s.pw[4 * i + 3]));
      pw.push_back(t4PwByte(0, 0, 0, 0));
    }
  }
  int r;
  void *old[] = {ix->dTable, ix->dHtab, ix->dPost, ix->dSeqs, ix->dCons, ix->dPw};
  for (void *p : old) if (p) (void)hipFree(p);
  ix->dTable = nullptr; ix->dHtab = nullptr; ix->dPost = nullptr; ix->dSeqs = nullptr; ix->dCons = nullptr; ix->dPw = nullptr;
  if ((r = devAlloc(c, &ix->dPost, post.size()))) return r;
  if ((r = devAlloc(c, &ix->dSeqs, infos.size()))) return r;
  if ((r = devAlloc(c, &ix->dCons, cons.size() + 16))) return r;
  if ((r = devAlloc(c, &ix->dPw, pw.size()))) return r;
  if (direct) { if ((r = devAlloc(c, &ix->dTable, table.size()))) return r; }
  else { if ((r = devAlloc(c, &ix->dHtab, htab.size()))) return r; }
  if (!post.empty()) HIPCHK(c, hipMemcpy(ix->dPost, post.data(), sizeof(int2) * post.size(), hipMemcpyHostToDevice));
  if (!infos.empty()) HIPCHK(c, hipMemcpy(ix->dSeqs, infos.data(), sizeof(T4SeqInfo) * infos.size(), hipMemcpyHostToDevice));
  if (!cons.empty()) HIPCHK(c, hipMemcpy(ix->dCons, cons.data(), cons.size(), hipMemcpyHostToDevice));
  if (!pw.empty()) HIPCHK(c, hipMemcpy(ix->dPw, pw.data(), sizeof(T4PW) * pw.size(), hipMemcpyHostToDevice));
  if (direct) HIPCHK(c, hipMemcpy(ix->dTable, table.data(), sizeof(uint2) * table.size(), hipMemcpyHostToDevice));
  else HIPCHK(c, hipMemcpy(ix->dHtab, htab.data(), sizeof(T4HashEnt) * htab.size(), hipMemcpyHostToDevice));
  T4IndexView &v = ix->view;
  memset(&v, 0, sizeof v);
  v.k = K; v.nseq = (int)ix->seqs.size(); v.direct = direct ? 1 : 0; v.considerBarcode = ix->considerBarcode;
  v.hashMask = hashMask; v.table = ix->dTable; v.htab = ix->dHtab; v.post = ix->dPost; v.seqs = ix->dSeqs;
  v.cons = ix->dCons; v.pw = ix->dPw;
  v.radius = ix->radius; v.hitLenRequired = ix->hitLenRequired; v.nomatchGapLimit = ix->nomatchGapLimit;
  v.firstIsRef = (!ix->seqs.empty() && ix->seqs[0].isRef) ? 1 : 0;
  for (const HostSeq &q : ix->seqs) if (q.isRef) { hasRef = true; break; }
  v.hasNovel = hasNovel ? (hasRef ? 1 : 2) : 0;
  { int maxLen = 0; for (const HostSeq &q : ix->seqs) if ((int)q.cons.size() > maxLen) maxLen = (int)q.cons.size(); v.key32 = t4Key32Bits(v.nseq, maxLen); }
  v.novelSim = ix->novelSim; v.refSim = ix->refSim; v.repeatSim = ix->repeatSim;
  ix->committed = true;
  return T4_OK;
}
}  // namespace

// ---- live set: patch the device image (see trust4_hip.h) ----------------------------------------------------------
namespace {
// grow a device array to `want` elements keeping its first `keep` elements (stream-ordered copy; the old block is freed
// once the copy is done)
template <class T> int growKeep(t4_ctx *c, T **p, int64_t keep, int64_t want) {
  T *np = nullptr;
  HIPCHK(c, hipMalloc(&np, sizeof(T) * (size_t)(want > 0 ? want : 1)));
  if (*p && keep > 0) HIPCHK(c, hipMemcpyAsync(np, *p, sizeof(T) * (size_t)keep, hipMemcpyDeviceToDevice, c->stream));
  if (*p) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(*p); }
  *p = np;
  return T4_OK;
}
}  // namespace

extern "C" {

int t4_index_apply_delta(t4_index *ix, const t4_index_delta *d) {
  if (!ix || !d) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  (void)hipSetDevice(c->device);
  if (ix->considerBarcode) return fail(c, T4_ERR_ARG, "t4_index_apply_delta: the index of a live set is not keyed by barcode");
  if (!ix->live && (!ix->seqs.empty() || ix->committed)) return fail(c, T4_ERR_STATE, "t4_index_apply_delta on an index that was filled through t4_index_add_*");
  if (d->table_slots < 2 || (d->table_slots & (d->table_slots - 1))) return fail(c, T4_ERR_ARG, "table_slots must be a power of two");
  if (d->nseq < 0 || d->nseq > T4_MAX_SEQS) return fail(c, T4_ERR_UNSUPPORTED, "more than %d sequences", T4_MAX_SEQS);
  if (d->max_seq_len > T4_MAX_SEQLEN) return fail(c, T4_ERR_UNSUPPORTED, "contig longer than %d", T4_MAX_SEQLEN);
  if (d->post_cap > 0xFFFFFFFFll) return fail(c, T4_ERR_UNSUPPORTED, "more than 2^32 postings");
  int r;
  if (!ix->stEvent) HIPCHK(c, hipEventCreate(&ix->stEvent));
  // capacities
  if (d->table_slots != ix->capTable || d->table_rebuilt) {
    if (d->table_slots != ix->capTable) {
      if (ix->dCtab) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(ix->dCtab); ix->dCtab = nullptr; }
      HIPCHK(c, hipMalloc(&ix->dCtab, sizeof(T4HashEntC) * (size_t)d->table_slots));
      ix->capTable = d->table_slots;
    }
    HIPCHK(c, hipMemsetAsync(ix->dCtab, 0xFF, sizeof(T4HashEntC) * (size_t)ix->capTable, c->stream));   // code ~0 = empty
  }
  if (d->post_cap > ix->capPost) { int64_t want = d->post_cap + d->post_cap / 2; if ((r = growKeep(c, &ix->dPost, ix->capPost, want))) return r; ix->capPost = want; }
  if (d->seq_cap > ix->capSeq) { int want = d->seq_cap + d->seq_cap / 2; if ((r = growKeep(c, &ix->dSeqs, (int64_t)ix->capSeq, (int64_t)want))) return r; ix->capSeq = want; }
  if (d->base_cap > ix->capBase) {
    int64_t want = d->base_cap + d->base_cap / 2;
    if ((r = growKeep(c, &ix->dCons, ix->capBase, want))) return r;
    if ((r = growKeep(c, &ix->dPw, ix->capBase, want))) return r;
    ix->capBase = want;
  }
  // staging: descriptors, then the payload (8-byte aligned pieces)
  auto al8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
  int64_t postTotal = 0, baseTotal = 0;
  for (int64_t i = 0; i < d->n_post_runs; ++i) {
    if (d->post_at[i] < 0 || d->post_len[i] < 0 || d->post_at[i] + d->post_len[i] > ix->capPost) return fail(c, T4_ERR_ARG, "posting run %lld outside the image", (long long)i);
    postTotal += d->post_len[i];
  }
  for (int64_t i = 0; i < d->n_base_runs; ++i) {
    if (d->base_at[i] < 0 || d->base_len[i] < 0 || d->base_at[i] + d->base_len[i] > ix->capBase) return fail(c, T4_ERR_ARG, "base run %lld outside the image", (long long)i);
    baseTotal += d->base_len[i];
  }
  const size_t nDesc = (size_t)d->n_slots + (size_t)d->n_post_runs + (size_t)d->n_seqs + 2 * (size_t)d->n_base_runs;
  if (nDesc == 0) goto view;
  {
    size_t bytes = al8(sizeof(T4CopyDesc) * nDesc) + sizeof(T4HashEntC) * (size_t)d->n_slots + 8 * (size_t)postTotal + al8(sizeof(T4SeqInfo)) * (size_t)d->n_seqs;
    for (int64_t i = 0; i < d->n_base_runs; ++i) bytes += 2 * al8((size_t)d->base_len[i]);
    // the staging buffer is free once the copy that read it is done: a wait for the stream since then (every query round ends with
    // one) says so without an event of its own; two deltas with no wait in between (query lanes) take the wait here
    if (ix->stPending && ix->stEpoch == c->syncEpoch) { HIPCHK(c, hipStreamSynchronize(c->stream)); ++c->syncEpoch; }
    ix->stPending = false;
    if (bytes > ix->stCap) {
      if (ix->stHost) (void)hipHostFree(ix->stHost);
      if (ix->stDev) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(ix->stDev); }
      ix->stHost = nullptr; ix->stDev = nullptr;
      ix->stCap = bytes * 2 > ((size_t)1 << 20) ? bytes * 2 : ((size_t)1 << 20);
      HIPCHK(c, hipHostMalloc(&ix->stHost, ix->stCap, hipHostMallocMapped));
      HIPCHK(c, hipHostGetDevicePointer((void **)&ix->stHostDev, ix->stHost, 0));
      HIPCHK(c, hipMalloc(&ix->stDev, ix->stCap));
    }
    T4CopyDesc *desc = (T4CopyDesc *)ix->stHost;
    size_t at = al8(sizeof(T4CopyDesc) * nDesc), nd = 0;
    for (int64_t i = 0; i < d->n_slots; ++i) {
      if (d->slot[i] < 0 || d->slot[i] >= ix->capTable) return fail(c, T4_ERR_ARG, "table slot %lld outside the table", (long long)d->slot[i]);
      T4HashEntC e; e.code = d->slot_code[i]; e.start = d->slot_start[i]; e.cnt = d->slot_cnt[i];
      memcpy(ix->stHost + at, &e, sizeof e);
      desc[nd].srcOff = at; desc[nd].dst = (unsigned char *)(ix->dCtab + d->slot[i]); desc[nd].bytes = sizeof e; ++nd;
      at += sizeof e;
    }
    int64_t pAt = 0;
    for (int64_t i = 0; i < d->n_post_runs; ++i) {
      const size_t b = 8 * (size_t)d->post_len[i];
      memcpy(ix->stHost + at, d->post_data + 2 * pAt, b);
      desc[nd].srcOff = at; desc[nd].dst = (unsigned char *)(ix->dPost + d->post_at[i]); desc[nd].bytes = b; ++nd;
      at += b; pAt += d->post_len[i];
    }
    for (int i = 0; i < d->n_seqs; ++i) {
      const int id = d->seq_id[i];
      if (id < 0 || id >= ix->capSeq || id >= d->nseq) return fail(c, T4_ERR_ARG, "sequence id %d outside the image", id);
      const t4_seq_record &q = d->seq[i];
      if (q.len < 0 || q.base_off < 0 || q.base_off + q.len + 1 > ix->capBase || q.base_off > 0x7FFFFFFFll) return fail(c, T4_ERR_ARG, "sequence %d outside the base arena", id);
      T4SeqInfo f;
      memset(&f, 0, sizeof f);
      f.consOff = (int)q.base_off; f.len = q.len; f.pwOff = (int)q.base_off; f.barcode = q.barcode; f.isRef = 0;
      char nm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      memcpy(nm, q.name, 8);
      const int gt = geneType(nm);
      f.geneType = gt < 0 ? 255 : (unsigned char)gt;
      f.name0 = (unsigned char)nm[0]; f.name1 = (unsigned char)nm[1]; f.name2 = (unsigned char)nm[2]; f.name3 = (unsigned char)nm[3];
      memset(ix->stHost + at, 0, al8(sizeof f));
      memcpy(ix->stHost + at, &f, sizeof f);
      desc[nd].srcOff = at; desc[nd].dst = (unsigned char *)(ix->dSeqs + id); desc[nd].bytes = sizeof f; ++nd;
      at += al8(sizeof f);
    }
    int64_t bAt = 0;
    for (int64_t i = 0; i < d->n_base_runs; ++i) {
      const size_t b = (size_t)d->base_len[i];
      memcpy(ix->stHost + at, d->base_cons + bAt, b);
      desc[nd].srcOff = at; desc[nd].dst = (unsigned char *)(ix->dCons + d->base_at[i]); desc[nd].bytes = b; ++nd;
      at += al8(b);
      memcpy(ix->stHost + at, d->base_pw + bAt, b);
      desc[nd].srcOff = at; desc[nd].dst = (unsigned char *)(ix->dPw + d->base_at[i]); desc[nd].bytes = b; ++nd;
      at += al8(b);
      bAt += d->base_len[i];
    }
    // A round's delta is a few kilobytes: the kernel reads descriptors and payload straight out of the pinned staging buffer (one
    // stream operation; an H2D copy in front of it was a hop to a copy engine and back on the ordered chain's critical path). The
    // whole image of a fresh set, or a rebuilt table, goes through the copy engine as before.
    const bool direct = at <= ((size_t)1 << 18);
    if (!direct) HIPCHK(c, hipMemcpyAsync(ix->stDev, ix->stHost, at, hipMemcpyHostToDevice, c->stream));
    ix->stPending = true; ix->stEpoch = c->syncEpoch;
    int grid = (int)((nd + 3) / 4);
    if (grid > c->cus * 8) grid = c->cus * 8;
    const unsigned char *stg = direct ? ix->stHostDev : ix->stDev;
    hipLaunchKernelGGL(t4k::deltaKernel, dim3(grid), dim3(256), 0, c->stream, stg, (const T4CopyDesc *)stg, (int)nd);
    HIPCHK(c, hipGetLastError());
  }
view:
  ix->live = true; ix->committed = true; ix->liveNseq = d->nseq;
  T4IndexView &v = ix->view;
  memset(&v, 0, sizeof v);
  v.k = ix->k; v.nseq = d->nseq; v.direct = 3; v.considerBarcode = 0;
  v.hashMask = (unsigned long long)ix->capTable - 1; v.ctab = ix->dCtab; v.post = ix->dPost; v.seqs = ix->dSeqs; v.cons = ix->dCons; v.pw = ix->dPw;
  v.radius = ix->radius; v.hitLenRequired = ix->hitLenRequired; v.nomatchGapLimit = ix->nomatchGapLimit;
  v.firstIsRef = 0; v.hasNovel = 2;
  v.key32 = t4Key32Bits(d->nseq, d->max_seq_len);
  v.novelSim = ix->novelSim; v.refSim = ix->refSim; v.repeatSim = ix->repeatSim;
  return T4_OK;
}

int t4_index_size(const t4_index *ix) { return ix ? (ix->live ? ix->liveNseq : (int)ix->seqs.size()) : 0; }
int t4_index_seq_len(const t4_index *ix, int i) { return (ix && i >= 0 && i < (int)ix->seqs.size()) ? (int)ix->seqs[i].cons.size() : -1; }
const char *t4_index_seq_name(const t4_index *ix, int i) { return (ix && i >= 0 && i < (int)ix->seqs.size()) ? ix->seqs[i].name.c_str() : nullptr; }
const char *t4_index_seq_consensus(const t4_index *ix, int i) { return (ix && i >= 0 && i < (int)ix->seqs.size()) ? ix->seqs[i].cons.c_str() : nullptr; }

// ---- reads -------------------------------------------------------------------------------------
int t4_reads_upload(t4_ctx *c, const char *bases, const int64_t *offsets, const int32_t *barcode, int64_t n, t4_batch **out) {
  return t4_reads_upload_flags(c, bases, offsets, barcode, n, 0, out);
}

int t4_reads_upload_flags(t4_ctx *c, const char *bases, const int64_t *offsets, const int32_t *barcode, int64_t n, int flags, t4_batch **out) {
  if (!c || !out || n < 0 || (n > 0 && (!bases || !offsets))) return T4_ERR_ARG;
  *out = nullptr;
  (void)hipSetDevice(c->device);
  int maxLen = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t l = offsets[i + 1] - offsets[i];
    if (l < 0) return fail(c, T4_ERR_ARG, "offsets not monotone at read %lld", (long long)i);
    if (l > T4_MAXL) return fail(c, T4_ERR_UNSUPPORTED, "read %lld is %lld bp; this engine takes reads up to %d bp", (long long)i, (long long)l, T4_MAXL);
    if (l > maxLen) maxLen = (int)l;
  }
  t4_batch *b = new t4_batch();
  b->ctx = c; b->n = n; b->maxLen = maxLen;
  b->wpk = (maxLen + 15) / 16; b->wnm = (maxLen + 31) / 32;
  if (b->wpk == 0) b->wpk = 1;
  if (b->wnm == 0) b->wnm = 1;
  std::vector<unsigned> pk((size_t)n * b->wpk, 0u), nm((size_t)n * b->wnm, 0u);
  std::vector<int> len((size_t)n);
  // 2-bit packing on a few host threads for large batches (a million reads: 0.2 s on one thread, more than their kernels take)
  std::atomic<long long> badRead(-1);
  auto packRange = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const char *s = bases + offsets[i];
      int l = (int)(offsets[i + 1] - offsets[i]);
      len[i] = l;
      unsigned *p = pk.data() + (size_t)i * b->wpk, *m = nm.data() + (size_t)i * b->wnm;
      for (int j = 0; j < l; ++j) {
        int v = nucNum(s[j]);
        if (v < 0) {
          if (s[j] == 'N') { m[j >> 5] |= 1u << (j & 31); v = 0; }
          else if ((flags & T4_READS_KMERS_ONLY) && s[j] >= 'A' && s[j] <= 'Z') v = 3;   // nucToNum[c - 'A'] & 3 of KmerCode::Append (KmerCode.hpp:99-106): a valid 'T'
          else { long long none = -1; badRead.compare_exchange_strong(none, (long long)i); return; }
        }
        p[j >> 4] |= (unsigned)v << ((j & 15) * 2);
      }
    }
  };
  {
    const int packThreads = getenv("T4_PACK_THREADS") ? atoi(getenv("T4_PACK_THREADS")) : 8;
    const long long packMin = getenv("T4_PACK_MIN") ? atoll(getenv("T4_PACK_MIN")) : 65536;   // testing aid: threads for small batches too
    int nt = n >= packMin ? packThreads : 1;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 0 && (unsigned)nt > hw) nt = (int)hw;
    if (nt <= 1) packRange(0, n);
    else {
      std::vector<std::thread> pool;
      for (int t = 0; t < nt; ++t) pool.emplace_back(packRange, n * t / nt, n * (t + 1) / nt);
      for (auto &th : pool) th.join();
    }
  }
  if (badRead.load() >= 0) {
    const long long i = badRead.load();
    char bad = '?';
    for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j) { const char ch = bases[j]; if (nucNum(ch) < 0 && ch != 'N' && !((flags & T4_READS_KMERS_ONLY) && ch >= 'A' && ch <= 'Z')) { bad = ch; break; } }
    delete b;
    return fail(c, T4_ERR_UNSUPPORTED, "read %lld has base '%c' (alphabet is ACGTN)", i, bad);
  }
  int r;
  // a HIP error below must not leak the batch and its device buffers
  #define UPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { t4_batch_destroy(b); return fail(c, T4_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } } while (0)
  // + 4 words: the packed k-mer extraction reads one word past a read's row (masked out), also for the last read
  if ((r = devAlloc(c, &b->dPk, pk.size() + 4)) || (r = devAlloc(c, &b->dNm, nm.size() + 4)) || (r = devAlloc(c, &b->dLen, len.size()))) { t4_batch_destroy(b); return r; }
  if (n > 0) {
    UPCHK(hipMemcpy(b->dPk, pk.data(), sizeof(unsigned) * pk.size(), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(b->dNm, nm.data(), sizeof(unsigned) * nm.size(), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(b->dLen, len.data(), sizeof(int) * len.size(), hipMemcpyHostToDevice));
  }
  if (barcode) {
    if ((r = devAlloc(c, &b->dBarcode, (size_t)n))) { t4_batch_destroy(b); return r; }
    if (n > 0) UPCHK(hipMemcpy(b->dBarcode, barcode, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  }
  #undef UPCHK
  b->view.pk = b->dPk; b->view.nm = b->dNm; b->view.len = b->dLen; b->view.barcode = b->dBarcode;
  b->view.wpk = b->wpk; b->view.wnm = b->wnm; b->view.n = n;
  *out = b;
  return T4_OK;
}

void t4_batch_destroy(t4_batch *b) {
  if (!b) return;
  void *ptrs[] = {b->dPk, b->dNm, b->dLen, b->dBarcode};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete b;
}
int64_t t4_batch_size(const t4_batch *b) { return b ? b->n : 0; }

}  // extern "C"

namespace {

template <int CAP, int MAXOV, int NT>
void launchTier(int grid, hipStream_t st, const T4IndexView &iv, const T4BatchView &bv, const T4Work &wk, const T4QueryArgs &qa) {
  if (qa.views) hipLaunchKernelGGL((t4k::queryKernel<CAP, MAXOV, NT, 2>), dim3(grid), dim3(NT), 0, st, iv, bv, wk, qa);
  else if (qa.mode == 5) hipLaunchKernelGGL((t4k::queryKernel<CAP, MAXOV, NT, 3>), dim3(grid), dim3(NT), 0, st, iv, bv, wk, qa);
  else if (qa.mode >= 2 && qa.mode <= 4) hipLaunchKernelGGL((t4k::queryKernel<CAP, MAXOV, NT, 1>), dim3(grid), dim3(NT), 0, st, iv, bv, wk, qa);
  else hipLaunchKernelGGL((t4k::queryKernel<CAP, MAXOV, NT, 0>), dim3(grid), dim3(NT), 0, st, iv, bv, wk, qa);
}

// Shared driver of t4_overlaps / t4_annotate_rough.
int runQuery(t4_index *ix, t4_batch *b, T4QueryArgs qa, bool useBarcode, bool noHits = false) {
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  if (b->ctx != c) return fail(c, T4_ERR_ARG, "batch belongs to another ctx");
  (void)hipSetDevice(c->device);
  const long long n = b->n;
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.reads = n;
  if (n == 0) return T4_OK;
  int r;
  int grids[T4_NTIER];
  int maxGrid = 1;
  for (int t = 0; t < T4_NTIER; ++t) {
    grids[t] = c->cus * TIER_BLOCKS_PER_CU[t];
    if (grids[t] * TIER_THREADS[t] > maxGrid) maxGrid = grids[t] * TIER_THREADS[t];
  }
  if ((r = ensureScratch(c, maxGrid))) return r;
  if ((r = ensurePerCall(c, n))) return r;
  HIPCHK(c, hipMemsetAsync(c->listCounts, 0, sizeof(int) * 16, c->stream));   // [0..8) tier counts, [8..16) tier work counters
  HIPCHK(c, hipMemsetAsync(c->status, 0, sizeof(int) * (size_t)n, c->stream));
  HIPCHK(c, hipMemsetAsync(c->hitCounter, 0, sizeof(unsigned long long), c->stream));
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  T4TierCaps caps;
  for (int t = 0; t < T4_NTIER - 1; ++t) caps.cap[t] = TIER_CAP[t];
  if (noHits) caps.cap[0] = 1 << 30;
  int binGrid = c->cus * 8;
  if ((long long)binGrid > n) binGrid = (int)n;
  hipLaunchKernelGGL(t4k::binKernel, dim3(binGrid), dim3(64), 0, c->stream, ix->view, b->view, useBarcode ? 1 : 0,
                     caps, c->lists, c->listCounts, (long long)n);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  c->stats.launches = 1;
  int hostCounts[8];
  for (int t = 0; t < T4_NTIER; ++t) {
    HIPCHK(c, hipMemcpyAsync(hostCounts, c->listCounts, sizeof(int) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int cnt = hostCounts[t];
    c->stats.tier_reads[t] = cnt;
    if (cnt == 0) continue;
    T4Work wk;
    memset(&wk, 0, sizeof wk);
    wk.list = c->lists + (size_t)t * n; wk.nList = cnt;
    wk.nextList = t + 1 < T4_NTIER ? c->lists + (size_t)(t + 1) * n : nullptr;
    wk.nextCount = t + 1 < T4_NTIER ? c->listCounts + (t + 1) : nullptr;
    wk.workNext = getenv("T4_STATIC_STRIDE") ? nullptr : c->listCounts + 8 + t;   // reads cost very different amounts: blocks fetch their next read
    wk.status = c->status; wk.hitCounter = c->hitCounter;
    wk.dpRows = c->dpRows; wk.dpDir = c->dpDir;
    int grid = grids[t] < cnt ? grids[t] : cnt;
    if (t == T4_NTIER - 1) {
      if ((r = ensureGlobalTier(c, grids[T4_NTIER - 1]))) return r;
      wk.gKeys = c->gKeys; wk.gPairs = c->gPairs; wk.gCand = c->gCand; wk.gOv = c->gOv; wk.gFin = c->gFin; wk.gOrd = c->gOrd;
      wk.gCap = G_CAP; wk.gMaxOv = G_MAXOV;
      launchTier<0, 0, G_THREADS>(grid, c->stream, ix->view, b->view, wk, qa);
    }
    // (threads per read and resident groups per CU of every tier: TIER_THREADS / TIER_BLOCKS_PER_CU above; measured in rounds 1-3)
    else if (t == 0) launchTier<1024, 64, 128>(grid, c->stream, ix->view, b->view, wk, qa);
    else if (t == 1) launchTier<2048, 128, 256>(grid, c->stream, ix->view, b->view, wk, qa);
    else if (t == 2) launchTier<3072, 128, 256>(grid, c->stream, ix->view, b->view, wk, qa);
    else if (t == 3) launchTier<4096, 256, 256>(grid, c->stream, ix->view, b->view, wk, qa);
    else launchTier<8192, 512, 512>(grid, c->stream, ix->view, b->view, wk, qa);
    HIPCHK(c, hipGetLastError());
    ++c->stats.launches;
  }
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  std::vector<int> status((size_t)n);
  unsigned long long hits = 0;
  HIPCHK(c, hipMemcpyAsync(status.data(), c->status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&hits, c->hitCounter, sizeof hits, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float msAll = 0, msChain = 0;
  HIPCHK(c, hipEventElapsedTime(&msAll, c->ev[0], c->ev[2]));
  HIPCHK(c, hipEventElapsedTime(&msChain, c->ev[1], c->ev[2]));
  c->stats.kernel_ms = msAll; c->stats.chain_kernel_ms = msChain; c->stats.total_hits = (int64_t)hits;
  for (long long i = 0; i < n; ++i)
    if (status[i] != 0)
      return fail(c, T4_ERR_UNSUPPORTED, "read %lld exceeds the engine limits (status %d: %s)", i, status[i],
                  status[i] == 2 ? "more than 262144 k-mer hits or 16384 overlaps" : "gap DP or contig count beyond scratch");
  return T4_OK;
}

int ensureResult(t4_ctx *c, size_t records) {
  if (records <= c->resultCap) return T4_OK;
  int r = devAlloc(c, &c->result, records);
  if (r) return r;
  c->resultCap = records;
  return T4_OK;
}

}  // namespace

extern "C" {

int t4_overlaps(t4_index *ix, t4_batch *b, int strand, int skip_repeats, int max_per_read, int32_t *counts, t4_overlap *out) {
  if (!ix || !b || max_per_read <= 0) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  int r;
  if ((r = ensurePerCall(c, b->n))) return r;
  if ((r = ensureResult(c, (size_t)b->n * max_per_read))) return r;
  T4QueryArgs qa;
  memset(&qa, 0, sizeof qa);
  qa.mode = 0; qa.strand = strand; qa.skipRepeats = skip_repeats; qa.maxPerRead = max_per_read;
  qa.counts = c->counts; qa.out = c->result;
  if ((r = runQuery(ix, b, qa, true))) return r;
  if (counts && b->n) HIPCHK(c, hipMemcpy(counts, c->counts, sizeof(int) * (size_t)b->n, hipMemcpyDeviceToHost));
  if (out && b->n) HIPCHK(c, hipMemcpy(out, c->result, sizeof(t4_overlap) * (size_t)b->n * max_per_read, hipMemcpyDeviceToHost));
  return T4_OK;
}

int t4_annotate_rough(t4_index *ref, t4_batch *b, t4_overlap *out) {
  if (!ref || !b) return T4_ERR_ARG;
  t4_ctx *c = ref->ctx;
  int r;
  if ((r = ensureResult(c, (size_t)b->n * 4))) return r;
  T4QueryArgs qa;
  memset(&qa, 0, sizeof qa);
  qa.mode = 1; qa.strand = 0; qa.skipRepeats = 0; qa.maxPerRead = 4; qa.counts = nullptr; qa.out = c->result;
  if ((r = runQuery(ref, b, qa, false))) return r;
  if (out && b->n) HIPCHK(c, hipMemcpy(out, c->result, sizeof(t4_overlap) * (size_t)b->n * 4, hipMemcpyDeviceToHost));
  return T4_OK;
}

int t4_hits(t4_index *ix, t4_batch *b, int strand, int allow_total_skip, int64_t *hit_offsets, t4_hit *hits, int64_t hits_cap) {
  if (!ix || !b || !hit_offsets) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  (void)hipSetDevice(c->device);
  const long long n = b->n;
  hit_offsets[0] = 0;
  if (n == 0) return T4_OK;
  int r;
  int grid = c->cus * 2;
  if ((long long)grid > n) grid = (int)n;
  const int HCAP = 65536;
  if (grid > c->hitsGrid) { if ((r = devAlloc(c, &c->hitsKeys, (size_t)grid * HCAP))) return r; c->hitsGrid = grid; }
  if ((r = ensurePerCall(c, n))) return r;
  long long *dOff = nullptr;
  T4HitOut *dHits = nullptr;
  if ((r = devAlloc(c, &dOff, (size_t)n + 1))) return r;
  HIPCHK(c, hipMemsetAsync(c->status, 0, sizeof(int) * (size_t)n, c->stream));
  hipLaunchKernelGGL(t4k::hitsKernel, dim3(grid), dim3(64), 0, c->stream, ix->view, b->view, strand, allow_total_skip, 0, dOff,
                     (T4HitOut *)nullptr, c->hitsKeys, HCAP, c->status);
  HIPCHK(c, hipGetLastError());
  std::vector<long long> off((size_t)n + 1);
  HIPCHK(c, hipMemcpyAsync(off.data(), dOff, sizeof(long long) * ((size_t)n + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  off[0] = 0;
  for (long long i = 0; i < n; ++i) off[i + 1] += off[i];
  for (long long i = 0; i <= n; ++i) hit_offsets[i] = off[i];
  std::vector<int> status((size_t)n);
  HIPCHK(c, hipMemcpy(status.data(), c->status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
  for (long long i = 0; i < n; ++i) if (status[i]) { (void)hipFree(dOff); return fail(c, T4_ERR_UNSUPPORTED, "read %lld has more than %d hits", i, HCAP); }
  if (hits) {
    if (hits_cap < off[n]) { (void)hipFree(dOff); return fail(c, T4_ERR_ARG, "hits_cap %lld < %lld", (long long)hits_cap, off[n]); }
    if ((r = devAlloc(c, &dHits, (size_t)off[n]))) { (void)hipFree(dOff); return r; }
    HIPCHK(c, hipMemcpyAsync(dOff, off.data(), sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(t4k::hitsKernel, dim3(grid), dim3(64), 0, c->stream, ix->view, b->view, strand, allow_total_skip, 1, dOff,
                       dHits, c->hitsKeys, HCAP, c->status);
    HIPCHK(c, hipGetLastError());
    if (off[n]) HIPCHK(c, hipMemcpyAsync(hits, dHits, sizeof(t4_hit) * (size_t)off[n], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(dHits);
  }
  (void)hipFree(dOff);
  return T4_OK;
}


int t4_gap_dp_align(t4_ctx *c, int kind, int impl, int n, const int64_t *t_off, const int64_t *p_off, const void *t_data,
                    const char *p_chars, int32_t *out4, signed char *align, int align_stride);
int t4_gap_dp(t4_ctx *c, int kind, int impl, int n, const int64_t *t_off, const int64_t *p_off, const void *t_data,
              const char *p_chars, int32_t *out4) {
  return t4_gap_dp_align(c, kind, impl, n, t_off, p_off, t_data, p_chars, out4, nullptr, 0);
}
// the same; impl 4 with kind 1 also returns the edit string of every alignment (align_stride bytes each, terminated by -1)
int t4_gap_dp_align(t4_ctx *c, int kind, int impl, int n, const int64_t *t_off, const int64_t *p_off, const void *t_data,
                    const char *p_chars, int32_t *out4, signed char *align, int align_stride) {
  if (!c || n < 0 || (n > 0 && (!t_off || !p_off || !t_data || !p_chars || !out4)) || (kind != 0 && kind != 1)) return T4_ERR_ARG;
  if (align && (align_stride < 2 || impl != 4 || kind != 1)) return T4_ERR_ARG;
  if (n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  int grid = (n + 63) / 64;
  if (grid > c->cus * 4) grid = c->cus * 4;
  int r;
  if ((r = ensureScratch(c, grid * 64))) return r;
  long long *dT = nullptr, *dP = nullptr;
  char *dTc = nullptr, *dPc = nullptr;
  T4PW *dTw = nullptr;
  int *dOut = nullptr;
  size_t tn = (size_t)t_off[n], pn = (size_t)p_off[n];
  if ((r = devAlloc(c, &dT, (size_t)n + 1)) || (r = devAlloc(c, &dP, (size_t)n + 1)) || (r = devAlloc(c, &dPc, pn + 16)) || (r = devAlloc(c, &dOut, (size_t)n * 4))) return r;
  if (kind == 0) { if ((r = devAlloc(c, &dTc, tn + 16))) return r; HIPCHK(c, hipMemcpy(dTc, t_data, tn, hipMemcpyHostToDevice)); }
  else {   // _posWeight columns (4 x int32) -> predicate bytes
    std::vector<T4PW> wb(tn + 1, t4PwByte(0, 0, 0, 0));
    const int32_t *w4 = (const int32_t *)t_data;
    for (size_t i = 0; i < tn; ++i) wb[i] = t4PwByte(w4[4 * i], w4[4 * i + 1], w4[4 * i + 2], w4[4 * i + 3]);
    if ((r = devAlloc(c, &dTw, tn + 16))) return r;
    HIPCHK(c, hipMemcpy(dTw, wb.data(), tn + 1, hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMemcpy(dT, t_off, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dP, p_off, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dPc, p_chars, pn, hipMemcpyHostToDevice));
  signed char *dAl = nullptr;
  if (align) { if ((r = devAlloc(c, &dAl, (size_t)n * align_stride))) return r; HIPCHK(c, hipMemsetAsync(dAl, 0xFF, (size_t)n * align_stride, c->stream)); }
  hipLaunchKernelGGL(t4k::gapDpKernel, dim3(grid), dim3(64), 0, c->stream, kind, impl, n, dT, dP, dTc, dTw, dPc, dOut, c->dpRows, c->dpDir, dAl, align_stride);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out4, dOut, sizeof(int) * (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  if (align) HIPCHK(c, hipMemcpyAsync(align, dAl, (size_t)n * align_stride, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  void *ptrs[] = {dT, dP, dTc, dPc, dTw, dOut, dAl};
  for (void *q : ptrs) if (q) (void)hipFree(q);
  return T4_OK;
}


int t4_mate_overlap(t4_ctx *c, int n, const int64_t *f_off, const char *f_chars, const int64_t *s_off, const char *s_chars,
                    const int32_t *min_overlap, int check_tandem, int32_t *out3) {
  if (!c || n < 0 || (n > 0 && (!f_off || !f_chars || !s_off || !s_chars || !min_overlap || !out3))) return T4_ERR_ARG;
  if (n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  int r;
  long long *dF = nullptr, *dS = nullptr;
  char *dFc = nullptr, *dSc = nullptr;
  int *dMo = nullptr, *dOut = nullptr;
  const size_t fn = (size_t)f_off[n], sn = (size_t)s_off[n];
  if ((r = devAlloc(c, &dF, (size_t)n + 1)) || (r = devAlloc(c, &dS, (size_t)n + 1)) || (r = devAlloc(c, &dFc, fn + 16)) ||
      (r = devAlloc(c, &dSc, sn + 16)) || (r = devAlloc(c, &dMo, (size_t)n)) || (r = devAlloc(c, &dOut, (size_t)n * 3))) return r;
  HIPCHK(c, hipMemcpy(dF, f_off, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dS, s_off, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice));
  if (fn) HIPCHK(c, hipMemcpy(dFc, f_chars, fn, hipMemcpyHostToDevice));
  if (sn) HIPCHK(c, hipMemcpy(dSc, s_chars, sn, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dMo, min_overlap, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  int grid = n < c->cus * 32 ? n : c->cus * 32;
  hipLaunchKernelGGL(t4k::mateOverlapKernel, dim3(grid), dim3(64), 0, c->stream, n, dF, dFc, dS, dSc, dMo, check_tandem ? 1 : 0, dOut);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out3, dOut, sizeof(int) * (size_t)n * 3, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  void *ptrs[] = {dF, dS, dFc, dSc, dMo, dOut};
  for (void *q : ptrs) if (q) (void)hipFree(q);
  for (int i = 0; i < n; ++i) if (out3[3 * i] == -2) return fail(c, T4_ERR_UNSUPPORTED, "pair %d has a read longer than %d bp", i, T4_MAXL);
  return T4_OK;
}

int t4_process_pairs(t4_ctx *c, int n, const int64_t *off1, const char *r1, const char *q1, const int64_t *off2, const char *r2, const char *q2,
                     const unsigned char *has_qual, const int64_t *out_off, char *out_r, char *out_q, int32_t *meta4) {
  if (!c || n < 0 || (n > 0 && (!off1 || !r1 || !off2 || !r2 || !has_qual || !out_off || !out_r || !out_q || !meta4))) return T4_ERR_ARG;
  if (n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  const size_t n1 = (size_t)off1[n], n2 = (size_t)off2[n], no = (size_t)out_off[n];
  for (int i = 0; i < n; ++i) {
    if (out_off[i + 1] - out_off[i] < (off1[i + 1] - off1[i]) + (off2[i + 1] - off2[i]) + 1) return fail(c, T4_ERR_ARG, "t4_process_pairs: pair %d needs an output slot of len1 + len2 + 1 characters", i);
    if ((has_qual[i] & 1) && !q1) return fail(c, T4_ERR_ARG, "t4_process_pairs: pair %d says read 1 has qualities, q1 is null", i);
    if ((has_qual[i] & 2) && !q2) return fail(c, T4_ERR_ARG, "t4_process_pairs: pair %d says read 2 has qualities, q2 is null", i);
  }
  int r;
  long long *dO1 = nullptr, *dO2 = nullptr, *dOo = nullptr;
  char *dR1 = nullptr, *dQ1 = nullptr, *dR2 = nullptr, *dQ2 = nullptr, *dOr = nullptr, *dOq = nullptr;
  unsigned char *dHq = nullptr;
  int4 *dMeta = nullptr;
  auto freeAll = [&] { void *ptrs[] = {dO1, dO2, dOo, dR1, dQ1, dR2, dQ2, dOr, dOq, dHq, dMeta}; for (void *q : ptrs) if (q) (void)hipFree(q); };
  #define PPCHK(x) do { if ((x) != hipSuccess) { freeAll(); return fail(c, T4_ERR_HIP, "HIP error in t4_process_pairs: %s", hipGetErrorString(hipGetLastError())); } } while (0)
  if ((r = devAlloc(c, &dO1, (size_t)n + 1)) || (r = devAlloc(c, &dO2, (size_t)n + 1)) || (r = devAlloc(c, &dOo, (size_t)n + 1)) ||
      (r = devAlloc(c, &dR1, n1 + 16)) || (r = devAlloc(c, &dR2, n2 + 16)) || (r = devAlloc(c, &dOr, no + 16)) || (r = devAlloc(c, &dOq, no + 16)) ||
      (r = devAlloc(c, &dHq, (size_t)n)) || (r = devAlloc(c, &dMeta, (size_t)n)) || (q1 && (r = devAlloc(c, &dQ1, n1 + 16))) || (q2 && (r = devAlloc(c, &dQ2, n2 + 16)))) { freeAll(); return r; }
  PPCHK(hipMemcpyAsync(dO1, off1, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
  PPCHK(hipMemcpyAsync(dO2, off2, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
  PPCHK(hipMemcpyAsync(dOo, out_off, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
  if (n1) PPCHK(hipMemcpyAsync(dR1, r1, n1, hipMemcpyHostToDevice, c->stream));
  if (n2) PPCHK(hipMemcpyAsync(dR2, r2, n2, hipMemcpyHostToDevice, c->stream));
  if (q1 && n1) PPCHK(hipMemcpyAsync(dQ1, q1, n1, hipMemcpyHostToDevice, c->stream));
  if (q2 && n2) PPCHK(hipMemcpyAsync(dQ2, q2, n2, hipMemcpyHostToDevice, c->stream));
  PPCHK(hipMemcpyAsync(dHq, has_qual, (size_t)n, hipMemcpyHostToDevice, c->stream));
  const int grid = n < c->cus * 32 ? n : c->cus * 32;
  hipLaunchKernelGGL(t4k::processPairKernel, dim3(grid), dim3(64), 0, c->stream, n, dO1, dR1, dQ1, dO2, dR2, dQ2, dHq, dOo, dOr, dOq, dMeta);
  PPCHK(hipGetLastError());
  PPCHK(hipMemcpyAsync(meta4, dMeta, sizeof(int4) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  PPCHK(hipMemcpyAsync(out_r, dOr, no, hipMemcpyDeviceToHost, c->stream));
  PPCHK(hipMemcpyAsync(out_q, dOq, no, hipMemcpyDeviceToHost, c->stream));
  PPCHK(hipStreamSynchronize(c->stream));
  #undef PPCHK
  freeAll();
  for (int i = 0; i < n; ++i) if (meta4[4 * i] == -2) return fail(c, T4_ERR_UNSUPPORTED, "pair %d has a read longer than %d bp", i, T4_MAXL);
  return T4_OK;
}

// ---- KmerCount (KmerCount.hpp) on the device ------------------------------------------------------------------------
struct t4_kmer_counter {
  t4_ctx *ctx = nullptr;
  T4KmerTable tb{};
  unsigned long long slots = 0, maxSlots = 0;
};

namespace {
// a table of `slots` slots (a power of two), zeroed; the counter's small words (overflow flag, used-slot count) stay as they are
int kmerTableAlloc(t4_ctx *c, unsigned long long slots, T4KmerTable &tb) {
  tb.keys = nullptr; tb.cnt = nullptr;
  if (hipMalloc(&tb.keys, sizeof(unsigned long long) * slots) != hipSuccess || hipMalloc(&tb.cnt, sizeof(unsigned) * slots) != hipSuccess) {
    if (tb.keys) (void)hipFree(tb.keys);
    tb.keys = nullptr;
    return fail(c, T4_ERR_HIP, "k-mer count table: no device memory for %llu slots", slots);
  }
  HIPCHK(c, hipMemsetAsync(tb.keys, 0, sizeof(unsigned long long) * slots, c->stream));
  HIPCHK(c, hipMemsetAsync(tb.cnt, 0, sizeof(unsigned) * slots, c->stream));
  tb.mask = slots - 1;
  return T4_OK;
}
// Room for `incoming` more k-mers at a load of at most 0.6: the table is rehashed on the device into one four times as large as
// often as that takes (never beyond the size the caller's max_kmers stands for: then the old behaviour -- a full table fails).
// raiseMax: the incoming k-mers are another read set's (t4_kmer_count_merge without only_present), which the caller's max_kmers at
// create time knew nothing of -- the ceiling moves so that what is there plus what comes fits.
int kmerEnsureRoom(t4_kmer_counter *kc, unsigned long long incoming, bool raiseMax = false) {
  t4_ctx *c = kc->ctx;
  unsigned long long used = 0;
  HIPCHK(c, hipMemcpyAsync(&used, kc->tb.used, sizeof used, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (raiseMax) while ((used + incoming) * 10ull > kc->maxSlots * 6ull && kc->maxSlots < (1ull << 40)) kc->maxSlots <<= 1;
  while (kc->slots < kc->maxSlots && (used + incoming) * 10ull > kc->slots * 6ull) {
    unsigned long long ns = kc->slots * 4ull;
    if (ns > kc->maxSlots) ns = kc->maxSlots;
    T4KmerTable to = kc->tb;
    int r;
    if ((r = kmerTableAlloc(c, ns, to))) return r;
    const unsigned long long blocks = (kc->slots + 255ull) / 256ull;
    const int grid = (int)(blocks < (unsigned long long)c->cus * 16ull ? blocks : (unsigned long long)c->cus * 16ull);
    hipLaunchKernelGGL(t4k::kmerRehashKernel, dim3(grid), dim3(256), 0, c->stream, kc->tb, to);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(kc->tb.keys); (void)hipFree(kc->tb.cnt);
    kc->tb = to; kc->slots = ns;
  }
  return T4_OK;
}
}  // namespace

int t4_kmer_count_create(t4_ctx *c, int k, int64_t max_kmers, int per_barcode, t4_kmer_counter **out) {
  if (!c || !out) return T4_ERR_ARG;
  if (k < 1 || k > 31 || max_kmers < 1) return fail(c, T4_ERR_ARG, "t4_kmer_count_create: k in 1..31 and max_kmers >= 1 (got %d, %lld)", k, (long long)max_kmers);
  if (per_barcode && k > 21) return fail(c, T4_ERR_UNSUPPORTED, "t4_kmer_count_create: per-barcode counts take k <= 21 (the barcode shares the 64-bit key)");
  (void)hipSetDevice(c->device);
  unsigned long long maxSlots = 1024;
  while (maxSlots < 2ull * (unsigned long long)max_kmers) maxSlots <<= 1;
  // max_kmers is an upper bound -- every k-mer position of the input -- and a read set repeats itself: a table for all of them was
  // 12 GB for a million pairs. It starts at an eighth of that (at least 4 M slots) and grows as it fills.
  unsigned long long slots = maxSlots / 8ull;
  if (slots < (1ull << 22)) slots = (1ull << 22) < maxSlots ? (1ull << 22) : maxSlots;
#ifdef T4_TEST_KNOBS   // (emulator / test builds only)
  if (getenv("T4_KC_SLOTS")) { slots = 1024; const unsigned long long want = strtoull(getenv("T4_KC_SLOTS"), nullptr, 10); while (slots < want && slots < maxSlots) slots <<= 1; }   // testing aid: a small first table (the growth path)
#endif
  t4_kmer_counter *kc = new t4_kmer_counter;
  kc->ctx = c; kc->slots = slots; kc->maxSlots = maxSlots;
  kc->tb.k = k; kc->tb.perBarcode = per_barcode ? 1 : 0;
  kc->tb.overflow = nullptr; kc->tb.used = nullptr;
  int r = kmerTableAlloc(c, slots, kc->tb);
  if (r || hipMalloc(&kc->tb.overflow, sizeof(int)) != hipSuccess || hipMalloc(&kc->tb.used, sizeof(unsigned long long)) != hipSuccess) {
    t4_kmer_count_destroy(kc);
    return r ? r : fail(c, T4_ERR_HIP, "t4_kmer_count_create: no device memory");
  }
  HIPCHK(c, hipMemsetAsync(kc->tb.overflow, 0, sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(kc->tb.used, 0, sizeof(unsigned long long), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = kc;
  return T4_OK;
}

void t4_kmer_count_destroy(t4_kmer_counter *kc) {
  if (!kc) return;
  if (kc->tb.keys) (void)hipFree(kc->tb.keys);
  if (kc->tb.cnt) (void)hipFree(kc->tb.cnt);
  if (kc->tb.overflow) (void)hipFree(kc->tb.overflow);
  if (kc->tb.used) (void)hipFree(kc->tb.used);
  delete kc;
}

int t4_kmer_count_add(t4_kmer_counter *kc, t4_batch *b) {
  if (!kc || !b) return T4_ERR_ARG;
  t4_ctx *c = kc->ctx;
  if (b->ctx != c) return fail(c, T4_ERR_ARG, "batch belongs to another ctx");
  if (b->n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  const long long n = b->n;
  if (kc->tb.perBarcode && !b->dBarcode) return fail(c, T4_ERR_ARG, "t4_kmer_count_add: per-barcode counts need a batch uploaded with barcodes");
  // in slices of reads whose k-mers (every one new, at worst) leave the table under a load of 0.6; between slices the table grows
  const long long perRead = b->maxLen >= kc->tb.k ? (long long)(b->maxLen - kc->tb.k + 1) : 1;
  for (long long lo = 0; lo < n;) {
    long long cnt = n - lo;
    if (kc->slots < kc->maxSlots) {
      long long fit = (long long)(kc->slots * 3ull / 10ull) / perRead;
      if (fit < 4096) fit = 4096;
      if (cnt > fit) cnt = fit;
      int r = kmerEnsureRoom(kc, (unsigned long long)(cnt * perRead));
      if (r) return r;
    }
    const int grid = (int)(cnt < (long long)c->cus * 32 ? cnt : (long long)c->cus * 32);
    hipLaunchKernelGGL(t4k::kmerAddKernel, dim3(grid), dim3(64), 0, c->stream, b->view, kc->tb, lo, lo + cnt);
    HIPCHK(c, hipGetLastError());
    lo += cnt;
  }
  int overflow = 0;
  HIPCHK(c, hipMemcpyAsync(&overflow, kc->tb.overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (overflow) return fail(c, T4_ERR_UNSUPPORTED, "t4_kmer_count_add: more distinct k-mers than the table was created for (%llu slots)", kc->slots);
  return T4_OK;
}

int t4_kmer_count_set(t4_kmer_counter *kc, const uint64_t *codes, const int32_t *counts, int64_t n) {
  if (!kc || n < 0 || (n > 0 && (!codes || !counts))) return T4_ERR_ARG;
  t4_ctx *c = kc->ctx;
  if (kc->tb.perBarcode) return fail(c, T4_ERR_ARG, "t4_kmer_count_set: not for per-barcode counters");
  if (n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  std::unordered_map<uint64_t, int32_t> last;   // a later record of the same k-mer overwrites an earlier one
  last.reserve((size_t)n * 2);
  for (int64_t i = 0; i < n; ++i) last[codes[i]] = counts[i];
  std::vector<unsigned long long> hc; std::vector<int> hv;
  hc.reserve(last.size()); hv.reserve(last.size());
  for (const auto &kv : last) { hc.push_back((unsigned long long)kv.first); hv.push_back((int)kv.second); }
  const long long m = (long long)hc.size();
  int r;
  if ((r = kmerEnsureRoom(kc, (unsigned long long)m))) return r;
  unsigned long long *dC = nullptr; int *dV = nullptr;
  if ((r = devAlloc(c, &dC, (size_t)m)) || (r = devAlloc(c, &dV, (size_t)m))) return r;
  HIPCHK(c, hipMemcpy(dC, hc.data(), sizeof(unsigned long long) * (size_t)m, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dV, hv.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
  const int grid = (int)((m + 255) / 256 < (long long)c->cus * 8 ? (m + 255) / 256 : (long long)c->cus * 8);
  hipLaunchKernelGGL(t4k::kmerSetKernel, dim3(grid), dim3(256), 0, c->stream, kc->tb, (const unsigned long long *)dC, (const int *)dV, m);
  HIPCHK(c, hipGetLastError());
  int overflow = 0;
  HIPCHK(c, hipMemcpyAsync(&overflow, kc->tb.overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(dC); (void)hipFree(dV);
  if (overflow) return fail(c, T4_ERR_UNSUPPORTED, "t4_kmer_count_set: more distinct k-mers than the table was created for (%llu slots)", kc->slots);
  return T4_OK;
}

int t4_kmer_count_export(t4_kmer_counter *kc, uint64_t *codes, int32_t *counts, int64_t cap, int64_t *n_out) {
  if (!kc || !n_out || cap < 0 || (cap > 0 && (!codes || !counts))) return T4_ERR_ARG;
  t4_ctx *c = kc->ctx;
  (void)hipSetDevice(c->device);
  unsigned long long used = 0;
  HIPCHK(c, hipMemcpyAsync(&used, kc->tb.used, sizeof used, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = (int64_t)used;
  if (cap == 0 || used == 0) return T4_OK;
  int r;
  unsigned long long *dC = nullptr, *dCur = nullptr; int *dV = nullptr;
  struct Free { unsigned long long *&a, *&b; int *&v; ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (v) (void)hipFree(v); } } freeOnReturn{dC, dCur, dV};   // (every path out of here)
  if ((r = devAlloc(c, &dC, (size_t)cap)) || (r = devAlloc(c, &dV, (size_t)cap)) || (r = devAlloc(c, &dCur, (size_t)1))) return r;
  HIPCHK(c, hipMemsetAsync(dCur, 0, sizeof(unsigned long long), c->stream));
  const unsigned long long blocks = (kc->slots + 255ull) / 256ull;
  const int grid = (int)(blocks < (unsigned long long)c->cus * 16ull ? blocks : (unsigned long long)c->cus * 16ull);
  hipLaunchKernelGGL(t4k::kmerExportKernel, dim3(grid), dim3(256), 0, c->stream, kc->tb, dC, dV, dCur, (unsigned long long)cap);
  HIPCHK(c, hipGetLastError());
  unsigned long long got = 0;
  HIPCHK(c, hipMemcpyAsync(&got, dCur, sizeof got, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t m = (size_t)(got < (unsigned long long)cap ? got : (unsigned long long)cap);
  if (m) {
    HIPCHK(c, hipMemcpy(codes, dC, sizeof(unsigned long long) * m, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(counts, dV, sizeof(int) * m, hipMemcpyDeviceToHost));
  }
  if (got != used) return fail(c, T4_ERR_STATE, "t4_kmer_count_export: %llu pairs in a table that counted %llu distinct k-mers", got, used);
  return T4_OK;
}

int t4_kmer_count_merge(t4_kmer_counter *kc, const uint64_t *codes, const int32_t *counts, int64_t n, int only_present) {
  if (!kc || n < 0 || (n > 0 && (!codes || !counts))) return T4_ERR_ARG;
  t4_ctx *c = kc->ctx;
  if (n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  int r;
  const int64_t SLICE = (int64_t)1 << 26;   // pairs per upload (768 MB of device memory at most)
  unsigned long long *dC = nullptr; int *dV = nullptr;
  const size_t cap = (size_t)(n < SLICE ? n : SLICE);
  struct Free { unsigned long long *&a; int *&v; ~Free() { if (a) (void)hipFree(a); if (v) (void)hipFree(v); } } freeOnReturn{dC, dV};   // (every path out of here)
  if ((r = devAlloc(c, &dC, cap)) || (r = devAlloc(c, &dV, cap))) return r;
  int overflow = 0;
  for (int64_t lo = 0; lo < n && !overflow; lo += SLICE) {
    const int64_t m = n - lo < SLICE ? n - lo : SLICE;
    if (!only_present && (r = kmerEnsureRoom(kc, (unsigned long long)m, true))) return r;
    HIPCHK(c, hipMemcpy(dC, codes + lo, sizeof(unsigned long long) * (size_t)m, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dV, counts + lo, sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
    const int grid = (int)((m + 255) / 256 < (long long)c->cus * 8 ? (m + 255) / 256 : (long long)c->cus * 8);
    hipLaunchKernelGGL(t4k::kmerMergeKernel, dim3(grid), dim3(256), 0, c->stream, kc->tb, (const unsigned long long *)dC, (const int *)dV, (long long)m, only_present ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&overflow, kc->tb.overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (overflow) return fail(c, T4_ERR_UNSUPPORTED, "t4_kmer_count_merge: more distinct k-mers than the table was created for (%llu slots)", kc->slots);
  return T4_OK;
}

int t4_kmer_count_stats(t4_kmer_counter *kc, t4_batch *b, const char *quals, const int64_t *qual_off,
                        int32_t *min_cnt, int32_t *median_cnt, float *avg_cnt, int32_t *new_len) {
  if (!kc || !b || !min_cnt || !median_cnt || !avg_cnt || !new_len || (quals && !qual_off)) return T4_ERR_ARG;
  t4_ctx *c = kc->ctx;
  if (b->ctx != c) return fail(c, T4_ERR_ARG, "batch belongs to another ctx");
  const long long n = b->n;
  if (n == 0) return T4_OK;
  if (kc->tb.perBarcode && !b->dBarcode) return fail(c, T4_ERR_ARG, "t4_kmer_count_stats: per-barcode counts need a batch uploaded with barcodes");
  (void)hipSetDevice(c->device);
  int r;
  char *dQ = nullptr; long long *dOff = nullptr;
  int *dMin = nullptr, *dMed = nullptr, *dLen = nullptr; float *dAvg = nullptr;
  if (quals) {
    std::vector<int> lens((size_t)n);
    HIPCHK(c, hipMemcpy(lens.data(), b->dLen, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    for (long long i = 0; i < n; ++i)
      if (qual_off[i + 1] - qual_off[i] != lens[(size_t)i]) return fail(c, T4_ERR_ARG, "read %lld has %d bases and %lld qualities", i, lens[(size_t)i], (long long)(qual_off[i + 1] - qual_off[i]));
    const size_t qn = (size_t)qual_off[n];
    if ((r = devAlloc(c, &dQ, qn + 16)) || (r = devAlloc(c, &dOff, (size_t)n + 1))) return r;
    if (qn) HIPCHK(c, hipMemcpy(dQ, quals, qn, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dOff, qual_off, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice));
  }
  if ((r = devAlloc(c, &dMin, (size_t)n)) || (r = devAlloc(c, &dMed, (size_t)n)) || (r = devAlloc(c, &dLen, (size_t)n)) || (r = devAlloc(c, &dAvg, (size_t)n))) return r;
  const int grid = (int)(n < (long long)c->cus * 32 ? n : (long long)c->cus * 32);
  hipLaunchKernelGGL(t4k::kmerStatsKernel, dim3(grid), dim3(64), 0, c->stream, b->view, kc->tb, (const char *)dQ, (const long long *)dOff, dMin, dMed, dAvg, dLen);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(min_cnt, dMin, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(median_cnt, dMed, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(avg_cnt, dAvg, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(new_len, dLen, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  void *ptrs[] = {dQ, dOff, dMin, dMed, dLen, dAvg};
  for (void *q : ptrs) if (q) (void)hipFree(q);
  return T4_OK;
}

int64_t t4_kmer_count_distinct(t4_kmer_counter *kc) {
  if (!kc) return -1;
  t4_ctx *c = kc->ctx;
  (void)hipSetDevice(c->device);
  std::vector<unsigned long long> keys((size_t)kc->slots);
  if (hipMemcpy(keys.data(), kc->tb.keys, sizeof(unsigned long long) * (size_t)kc->slots, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int64_t d = 0;
  for (unsigned long long x : keys) if (x) ++d;
  return d;
}

int t4_has_hit(t4_index *ref, t4_batch *b, int mode, int32_t *out) {
  if (!ref || !b || !out) return T4_ERR_ARG;
  t4_ctx *c = ref->ctx;
  if (mode != 0) return fail(c, T4_ERR_UNSUPPORTED, "t4_has_hit: only mode 0 (the stage-0 extractor's) is built");
  int r;
  if ((r = ensurePerCall(c, b->n))) return r;
  T4QueryArgs qa;
  memset(&qa, 0, sizeof qa);
  qa.mode = 5; qa.ret = c->counts;
  if ((r = runQuery(ref, b, qa, false))) return r;
  if (b->n > 0) {
    HIPCHK(c, hipMemcpyAsync(out, c->counts, sizeof(int) * (size_t)b->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return T4_OK;
}

int t4_assign(t4_index *ix, t4_batch *b, int strand, int32_t *ret, t4_overlap *out) {
  if (!ix || !b) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (ix->committed && ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_assign needs a contig set (ExtendOverlap aligns against posWeight)");
  int r;
  if ((r = ensurePerCall(c, b->n))) return r;
  if ((r = ensureResult(c, (size_t)b->n))) return r;
  T4QueryArgs qa;
  memset(&qa, 0, sizeof qa);
  memset(&qa, 0, sizeof qa);
  qa.mode = 2; qa.strand = strand; qa.skipRepeats = 0; qa.maxPerRead = 1; qa.counts = nullptr; qa.out = c->result; qa.ret = c->counts;
  if ((r = runQuery(ix, b, qa, true))) return r;
  if (ret && b->n) HIPCHK(c, hipMemcpy(ret, c->counts, sizeof(int) * (size_t)b->n, hipMemcpyDeviceToHost));
  if (out && b->n) HIPCHK(c, hipMemcpy(out, c->result, sizeof(t4_overlap) * (size_t)b->n, hipMemcpyDeviceToHost));
  return T4_OK;
}

int t4_assign_strands(t4_index *ix, t4_batch *b, const int32_t *strands, int32_t *ret, t4_overlap *out) {
  if (!ix || !b || (b->n > 0 && !strands)) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (ix->committed && ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_assign_strands needs a contig set (ExtendOverlap aligns against posWeight)");
  if (b->n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  int r;
  if ((r = ensurePerCall(c, b->n))) return r;
  if ((r = ensureResult(c, (size_t)b->n))) return r;
  int *dSt = nullptr;
  if ((r = devAlloc(c, &dSt, (size_t)b->n))) return r;
  HIPCHK(c, hipMemcpy(dSt, strands, sizeof(int) * (size_t)b->n, hipMemcpyHostToDevice));
  T4QueryArgs qa;
  memset(&qa, 0, sizeof qa);
  qa.mode = 2; qa.strand = 0; qa.strandPerRead = dSt; qa.skipRepeats = 0; qa.maxPerRead = 1; qa.counts = nullptr; qa.out = c->result; qa.ret = c->counts;
  r = runQuery(ix, b, qa, true);
  if (r == T4_OK && ret) { if (hipMemcpy(ret, c->counts, sizeof(int) * (size_t)b->n, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, T4_ERR_HIP, "copy of the AssignRead return values failed"); }
  if (r == T4_OK && out) { if (hipMemcpy(out, c->result, sizeof(t4_overlap) * (size_t)b->n, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, T4_ERR_HIP, "copy of the AssignRead results failed"); }
  (void)hipFree(dSt);
  return r;
}

int t4_posweight_recompute(t4_index *ix, t4_batch *b, const t4_overlap *assign, const int32_t *mult, int32_t *posweight, int64_t posweight_cap) {
  return t4_consensus_recompute(ix, b, assign, mult, posweight, posweight_cap, nullptr, 0, nullptr);
}

int t4_consensus_recompute(t4_index *ix, t4_batch *b, const t4_overlap *assign, const int32_t *mult, int32_t *posweight, int64_t posweight_cap,
                           char *consensus, int64_t consensus_cap, int64_t *changed) {
  if (!ix || !b || (b->n > 0 && !assign) || !posweight) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  if (ix->live) return fail(c, T4_ERR_UNSUPPORTED, "t4_posweight_recompute takes a committed contig set (the extendedSeq of main.cpp:2047), not a live one");
  if (b->ctx != c) return fail(c, T4_ERR_ARG, "batch belongs to another ctx");
  int64_t bases = 0;
  for (const HostSeq &q : ix->seqs) { if (q.isRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_posweight_recompute needs a contig set"); bases += (int64_t)q.cons.size(); }
  if (posweight_cap < 4 * bases) return fail(c, T4_ERR_ARG, "posweight buffer holds %lld values, the set needs %lld", (long long)posweight_cap, (long long)(4 * bases));
  if (consensus && consensus_cap < bases) return fail(c, T4_ERR_ARG, "consensus buffer holds %lld bases, the set has %lld", (long long)consensus_cap, (long long)bases);
  if (changed) *changed = 0;
  if (bases == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  int r;
  int *dCnt = nullptr, *dMult = nullptr;
  T4OverlapOut *dAs = nullptr;
  const size_t n = (size_t)b->n;
  const size_t cols = (size_t)bases + ix->seqs.size();   // the image's column space: every contig is followed by one terminator column (T4SeqInfo::pwOff)
  if ((r = devAlloc(c, &dCnt, cols * 4))) return r;
  char *dCons = nullptr;
  unsigned long long *dChanged = nullptr;
  auto freeAll = [&] { if (dCnt) (void)hipFree(dCnt); if (dMult) (void)hipFree(dMult); if (dAs) (void)hipFree(dAs); if (dCons) (void)hipFree(dCons); if (dChanged) (void)hipFree(dChanged); };
  #define PWCHK(x) do { if ((x) != hipSuccess) { freeAll(); return fail(c, T4_ERR_HIP, "HIP error in t4_posweight_recompute: %s", hipGetErrorString(hipGetLastError())); } } while (0)
  PWCHK(hipMemsetAsync(dCnt, 0, sizeof(int) * cols * 4, c->stream));   // posWeight.SetZero of every contig
  if (n > 0) {
    if ((r = devAlloc(c, &dAs, n))) { freeAll(); return r; }
    PWCHK(hipMemcpyAsync(dAs, assign, sizeof(t4_overlap) * n, hipMemcpyHostToDevice, c->stream));
    if (mult) {
      if ((r = devAlloc(c, &dMult, n))) { freeAll(); return r; }
      PWCHK(hipMemcpyAsync(dMult, mult, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    }
    const int grid = (int)(n < (size_t)c->cus * 32 ? n : (size_t)c->cus * 32);
    hipLaunchKernelGGL(t4k::posWeightAccumulateKernel, dim3(grid), dim3(64), 0, c->stream, ix->view, b->view, (const T4OverlapOut *)dAs, (const int *)dMult, dCnt);
    PWCHK(hipGetLastError());
  }
  {
    const int nseq = (int)ix->seqs.size();
    const int grid = nseq < c->cus * 8 ? nseq : c->cus * 8;
    hipLaunchKernelGGL(t4k::posWeightFinishKernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, c->stream, ix->view, dCnt);
    PWCHK(hipGetLastError());
  }
  std::vector<char> allCons;
  unsigned long long nChanged = 0;
  if (consensus) {   // UpdateConsensus of every contig from the columns just rebuilt
    if ((r = devAlloc(c, &dCons, cols))) { freeAll(); return r; }
    if ((r = devAlloc(c, &dChanged, 1))) { freeAll(); return r; }
    PWCHK(hipMemsetAsync(dChanged, 0, sizeof(unsigned long long), c->stream));
    const int nseq = (int)ix->seqs.size();
    const int grid = nseq < c->cus * 8 ? nseq : c->cus * 8;
    hipLaunchKernelGGL(t4k::consensusArgmaxKernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, c->stream, ix->view, (const int *)dCnt, dCons, dChanged);
    PWCHK(hipGetLastError());
    allCons.resize(cols);
    PWCHK(hipMemcpyAsync(allCons.data(), dCons, cols, hipMemcpyDeviceToHost, c->stream));
    PWCHK(hipMemcpyAsync(&nChanged, dChanged, sizeof nChanged, hipMemcpyDeviceToHost, c->stream));
  }
  std::vector<int32_t> all(cols * 4);
  PWCHK(hipMemcpyAsync(all.data(), dCnt, sizeof(int) * cols * 4, hipMemcpyDeviceToHost, c->stream));
  PWCHK(hipStreamSynchronize(c->stream));
  if (changed) *changed = (int64_t)nChanged;
  {   // contig after contig, without the terminator columns
    size_t from = 0, to = 0;
    for (const HostSeq &q : ix->seqs) {
      const size_t ln = q.cons.size();
      if (ln) memcpy(posweight + 4 * to, all.data() + 4 * from, sizeof(int32_t) * 4 * ln);
      if (ln && consensus) memcpy(consensus + to, allCons.data() + from, ln);
      from += ln + 1; to += ln;
    }
  }
  #undef PWCHK
  freeAll();
  return T4_OK;
}

int t4_extend(t4_index *ix, t4_batch *b, int max_per_read, const int32_t *counts, const t4_overlap *in, double mismatch_factor,
              int32_t *ret, t4_overlap *out) {
  if (!ix || !b || max_per_read <= 0 || (b->n > 0 && (!counts || !in))) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (ix->committed && ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_extend needs a contig set (ExtendOverlap aligns against posWeight)");
  if (max_per_read > 128) return fail(c, T4_ERR_UNSUPPORTED, "t4_extend takes at most 128 overlaps per read");
  const size_t n = (size_t)b->n, m = n * max_per_read;
  if (n == 0) return T4_OK;
  (void)hipSetDevice(c->device);
  int r;
  if ((r = ensurePerCall(c, b->n))) return r;
  if ((r = ensureResult(c, m))) return r;
  T4OverlapOut *dIn = nullptr;
  int *dCnt = nullptr, *dRet = nullptr;
  if ((r = devAlloc(c, &dIn, m)) || (r = devAlloc(c, &dCnt, n)) || (r = devAlloc(c, &dRet, m))) return r;
  HIPCHK(c, hipMemcpy(dIn, in, sizeof(t4_overlap) * m, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dCnt, counts, sizeof(int) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(dRet, 0, sizeof(int) * m));
  T4QueryArgs qa;
  memset(&qa, 0, sizeof qa);
  memset(&qa, 0, sizeof qa);
  qa.mode = 3; qa.maxPerRead = max_per_read; qa.out = c->result; qa.in = dIn; qa.inCounts = dCnt; qa.ret = dRet; qa.mismatchFactor = mismatch_factor;
  r = runQuery(ix, b, qa, true, true);
  if (r == T4_OK) {
    if (ret) HIPCHK(c, hipMemcpy(ret, dRet, sizeof(int) * m, hipMemcpyDeviceToHost));
    if (out) HIPCHK(c, hipMemcpy(out, c->result, sizeof(t4_overlap) * m, hipMemcpyDeviceToHost));
  }
  (void)hipFree(dIn); (void)hipFree(dCnt); (void)hipFree(dRet);
  return r;
}

}  // extern "C"

namespace {
// The query half of SeqSet::AddRead for a batch of reads, lean path: one blob in, a small header blob out, and the result
// records written by the kernel straight into a pinned host pool (variable number per read, SeqSet's std::vector<_overlap>):
// one launch (plus one per overflow tier that is needed) and one stream synchronisation per call. Either every read is
// matched against `base` (viewOf == nullptr; first launch on the 8192-hit LDS tier) or read i against views[viewOf[i]]
// (per-barcode images: reads meet a handful of contigs, so the first launch is the 1024-hit tier at 6 groups / CU).
// On return counts[i] records of read i start at index base[i] of ov / ext / ret (valid until the next call on this ctx).
struct AqResult { const int32_t *counts, *base; const t4_overlap *ov, *ext; const int32_t *ret; };
// tierHint (nullable, in/out): nonzero = the read is known to outgrow the LDS tiers, it is launched on the global-scratch tier
// at once, on a second stream beside the LDS tier; on return nonzero for every read the global-scratch tier served.
int aqLaunch(t4_ctx *c);
int aqBegin(t4_ctx *c, const T4IndexView &base, const T4IndexView *views, const int32_t *viewOf, bool smallFirst, int n, const char *bases,
            const int64_t *offsets, const int32_t *barcodes, const int32_t *strands, int skip_repeats, const double *factors,
            unsigned char *tierHint = nullptr, bool lean = false, const int32_t *onlySeq = nullptr, const int32_t *forceMin = nullptr, int wantCands = 0) {
  (void)hipSetDevice(c->device);
  AqCall &q = c->aq;
  if (q.active) return fail(c, T4_ERR_STATE, "an AddRead query is already in flight on this ctx");
  int maxLen = 1;
  for (int i = 0; i < n; ++i) {
    int64_t l = offsets[i + 1] - offsets[i];
    if (l < 0 || l > T4_MAXL) return fail(c, T4_ERR_UNSUPPORTED, "read %d is %lld bp; this engine takes reads up to %d bp", i, (long long)l, T4_MAXL);
    if (l > maxLen) maxLen = (int)l;
  }
  q.lean = lean;
  q.base = base; q.views = views; q.hasViewOf = viewOf != nullptr; q.smallFirst = smallFirst; q.n = n; q.skipRepeats = skip_repeats; q.tierHint = tierHint; q.attempt = 0;
  const int wpk = (maxLen + 15) / 16, wnm = (maxLen + 31) / 32;
  q.wpk = wpk; q.wnm = wnm;
  // input blob: pk | nm | len | barcode | strand | list(iota) | viewOf | factor
  // header blob: counts | status | next1 | next2 | outBase | ticks | statistics-stable flags | tail{overflow1, overflow2, hits, pool cursor}
  auto al8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
  q.oPk = 0; q.oNm = al8(q.oPk + sizeof(unsigned) * (size_t)n * wpk); q.oLen = al8(q.oNm + sizeof(unsigned) * (size_t)n * wnm);
  q.oBc = al8(q.oLen + sizeof(int) * (size_t)n); q.oSt = al8(q.oBc + sizeof(int) * (size_t)n); q.oLs = al8(q.oSt + sizeof(int) * (size_t)n);
  q.oVw = al8(q.oLs + sizeof(int) * (size_t)n); q.oFa = al8(q.oVw + sizeof(int) * (size_t)n); q.oOnly = al8(q.oFa + sizeof(double) * (size_t)n);
  q.oForce = al8(q.oOnly + sizeof(int) * (size_t)n);
  q.oCs = al8(q.oForce + sizeof(int) * (size_t)n);
  q.oWide = al8(q.oCs + sizeof(T4CandArgs)); q.hasOnly = onlySeq != nullptr; q.hasForce = forceMin != nullptr; q.wantCands = (wantCands & 1) != 0; q.useMarks = (wantCands & 2) != 0;
  q.oWideA = al8(q.oWide + sizeof(T4Wide)); q.inBytes = al8(q.oWideA + sizeof(T4Wide));
  q.pCnt = 0; q.pSta = al8(q.pCnt + sizeof(int) * (size_t)n); q.pNext = al8(q.pSta + sizeof(int) * (size_t)n);
  q.pNext2 = al8(q.pNext + sizeof(int) * (size_t)n); q.pBase = al8(q.pNext2 + sizeof(int) * (size_t)n);
  q.pTick = al8(q.pBase + sizeof(int) * (size_t)n); q.pStab = al8(q.pTick + sizeof(int) * (size_t)n); q.pAux = al8(q.pStab + sizeof(int) * (size_t)n);
  q.pN4 = al8(q.pAux + sizeof(int) * (size_t)n); q.pCb = al8(q.pN4 + sizeof(int) * (size_t)n); q.pCc = al8(q.pCb + sizeof(int) * (size_t)n);
  q.pS8 = al8(q.pCc + sizeof(int) * (size_t)n); q.pTail = al8(q.pS8 + sizeof(int) * T4_QSTATS * (size_t)n);   // tail: overflow1 | overflow2 | hits (8 B) | pool cursor | dir overflow | cand cursor | cand overflow
  q.pWctl = q.pTail + 48; q.pWplan = q.pWctl + 32; q.pWstat = al8(q.pWplan + sizeof(T4WidePlan) * (size_t)n);
  q.pWctlA = al8(q.pWstat + sizeof(int) * T4_WIDE_STAT * (size_t)n); q.pWplanA = q.pWctlA + 32; q.pWstatA = al8(q.pWplanA + sizeof(T4WidePlan) * (size_t)n);
  q.outBytes = al8(q.pWstatA + sizeof(int) * T4_WIDE_STAT * (size_t)n);
  q.wideSafety = c->wideSafetyKeep;
  if (q.inBytes > c->aqInBytes) {
    if (c->aqIn) (void)hipFree(c->aqIn);
    if (c->aqInHost) (void)hipHostFree(c->aqInHost);
    c->aqIn = nullptr; c->aqInHost = nullptr;
    HIPCHK(c, hipMalloc(&c->aqIn, q.inBytes * 2));
    HIPCHK(c, hipHostMalloc(&c->aqInHost, q.inBytes * 2, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->aqInHostDev, c->aqInHost, 0));
    c->aqInBytes = q.inBytes * 2;
  }
  if (q.outBytes > c->aqOutBytes) {
    if (c->aqOut) (void)hipFree(c->aqOut);
    if (c->aqOutHost) (void)hipHostFree(c->aqOutHost);
    c->aqOut = nullptr; c->aqOutHost = nullptr;
    HIPCHK(c, hipMalloc(&c->aqOut, q.outBytes * 2));
    HIPCHK(c, hipHostMalloc(&c->aqOutHost, q.outBytes * 2, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->aqOutHostDev, c->aqOutHost, 0));
    c->aqOutBytes = q.outBytes * 2;
  }
  auto tNow = [] { return std::chrono::steady_clock::now(); };
  auto tSince = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  auto tp0 = tNow();
  unsigned char *h = c->aqInHost;
  memset(h, 0, q.inBytes);
  unsigned *pk = (unsigned *)(h + q.oPk), *nm = (unsigned *)(h + q.oNm);
  int *len = (int *)(h + q.oLen), *bc = (int *)(h + q.oBc), *st = (int *)(h + q.oSt), *ls = (int *)(h + q.oLs), *vw = (int *)(h + q.oVw);
  double *fa = (double *)(h + q.oFa);
  for (int i = 0; i < n; ++i) {
    const char *s = bases + offsets[i];
    int l = (int)(offsets[i + 1] - offsets[i]);
    len[i] = l; bc[i] = barcodes ? barcodes[i] : -1; st[i] = strands[i]; fa[i] = factors[i]; vw[i] = viewOf ? viewOf[i] : 0;
    ((int *)(h + q.oOnly))[i] = onlySeq ? onlySeq[i] : -1;
    ((int *)(h + q.oForce))[i] = forceMin ? forceMin[i] : 0;
    unsigned *p = pk + (size_t)i * wpk, *qn = nm + (size_t)i * wnm;
    for (int j = 0; j < l; ++j) {
      int v = nucNum(s[j]);
      if (v < 0) {
        if (s[j] != 'N') return fail(c, T4_ERR_UNSUPPORTED, "read %d has base '%c' (alphabet is ACGTN)", i, s[j]);
        qn[j >> 5] |= 1u << (j & 31); v = 0;
      }
      p[j >> 4] |= (unsigned)v << ((j & 15) * 2);
    }
  }
  // one big contig set, plain passes: a read beyond the LDS tier goes to the wide query (t4_wide.h) instead of one workgroup's
  // global scratch -- decided on the device, inside the one launch every read starts in (T4_WIDE_OFF: the global-scratch tier as before)
  q.wide = !views && !smallFirst && !skip_repeats && base.hasNovel == 2 && !getenv("T4_WIDE_OFF");   // (read per call: tests/test_wide_query.py switches it between two calls on one ctx)   // (a read with a barcode stays on the old path: wideWant in processRead)
  q.onlyRestricted = false;
  if (onlySeq) {   // a round of restricted re-queries only defers nothing: the wide query's five grids stay unlaunched, and its reads
    bool anyWhole = false;   // (a few overlaps with one contig each) extend inside the query kernel -- no extendKernel behind it
    for (int i = 0; i < n && !anyWhole; ++i) anyWhole = onlySeq[i] < 0;
    if (!anyWhole) { q.wide = false; q.onlyRestricted = !views && !smallFirst; }
  }
  // work lists: the reads of the first LDS launch, then those that go to the global-scratch tier at once
  const bool forceGlobal = c->aqEnv.forceGlobal;   // testing aid: every read on the global-scratch tier
  if (forceGlobal && !smallFirst) { q.allGlobal.assign((size_t)n, 1); q.tierHint = tierHint = q.allGlobal.data(); q.wide = false; }
  // (with the wide query on, a hinted read -- served wide the last time -- starts on the second stream: wideSeedKernel)
  auto direct = [&](int i) { return tierHint && tierHint[i] && !smallFirst && !(onlySeq && onlySeq[i] >= 0) && !(q.wide && c->aqEnv.wideNoHint); };
  int nFirst = 0, nDirect = 0;
  for (int i = 0; i < n; ++i) if (!direct(i)) ls[nFirst++] = i;
  for (int i = 0; i < n; ++i) if (direct(i)) ls[nFirst + nDirect++] = i;
  q.nFirst = nFirst; q.nDirect = nDirect;
  c->aqSecPack += tSince(tp0);
  const int r = aqLaunch(c);
  if (r == T4_OK) q.active = true;
  return r;
}

// the wide query of the reads the round's query kernel deferred, on the ctx's stream. known: the header of the call is on the host
// (the counts of deferred reads and their partitions size the grids); else the grids are sized from what recent calls needed
void aqLaunchDeferredWide(t4_ctx *c, bool known) {
  AqCall &q = c->aq;
  T4Wide w;
  memcpy(&w, c->aqInHost + q.oWide, sizeof w);
  const int cus = c->cus > 0 ? c->cus : 1;
  // The grids are persistent (any size serves any number of reads / partitions); an empty grid of several hundred 100 KB-LDS workgroups
  // still takes microseconds to come and go.
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
  int parts = 2 * c->wideRecentParts + 4, reads = 2 * c->wideRecentReads + 2;
  if (known) { const int *ctl = (const int *)(c->aqOutHost + q.pWctl); parts = ctl[1] + 1; reads = ctl[0]; }
  const int gParts = clampi(parts, 4, cus * 2), gReads = clampi(reads, known ? 1 : 2, cus < 64 ? cus : 64);
  hipLaunchKernelGGL(t4k::wideScatterKernel, dim3(clampi(8 * gParts, 16, cus * 8)), dim3(256), 0, c->stream, q.base, w);   // (a partition is planned for four of the kernel's chunks)
  hipLaunchKernelGGL((t4k::wideSortKernel<8192>), dim3(gParts), dim3(512), 0, c->stream, q.base, w);
  hipLaunchKernelGGL(t4k::wideStatsKernel, dim3(gReads), dim3(512), 0, c->stream, q.base, w);
  hipLaunchKernelGGL((t4k::wideChainKernel<8192, 1 << T4_WIDE_OVBITS>), dim3(gParts < cus ? gParts : cus), dim3(512), 0, c->stream, q.base, q.bv, q.wk, q.qa, w);
  hipLaunchKernelGGL(t4k::wideMergeKernel, dim3(gReads), dim3(512), 0, c->stream, q.base, q.bv, q.wk, q.qa, w);
}
int aqLaunchEpilogue(t4_ctx *c) {   // the header block into pinned host memory, then the call's sequence number into the word the host polls (aqEpilogueKernel)
  AqCall &q = c->aq;
  const unsigned long long outW = q.outBytes >> 3;
  int grid = (int)((outW + 4095) / 4096);
  if (grid < 1) grid = 1;
  if (grid > 64) grid = 64;
  ++c->aqSeq;
  if (c->aqSeq == 0) c->aqSeq = 1;
  hipLaunchKernelGGL(t4k::aqEpilogueKernel, dim3(grid), dim3(256), 0, c->stream, (const unsigned long long *)c->aqOut, (unsigned long long *)c->aqOutHostDev, outW,
                     c->aqDoneCtr, c->aqFlagDev, c->aqSeq);
  HIPCHK(c, hipGetLastError());
  return T4_OK;
}

// copies, kernels and the header copy of one attempt, enqueued on the ctx's stream (nothing waits here)
int aqLaunch(t4_ctx *c) {
  AqCall &q = c->aq;
  const int n = q.n, nFirst = q.nFirst, nDirect = q.nDirect;
  const bool smallFirst = q.smallFirst;
  int r;
  if (!c->aqPool) {
    const int poolCap0 = c->aqEnv.poolCap;   // testing aid: a small pool forces the grow-and-repeat path
    if (!c->aqPoolCap) c->aqPoolCap = poolCap0 > 0 ? poolCap0 : 1 << 16;
    const size_t rec = (size_t)c->aqPoolCap;
    HIPCHK(c, hipHostMalloc(&c->aqPool, rec * (2 * sizeof(t4_overlap) + sizeof(int32_t)), hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->aqPoolDev, c->aqPool, 0));
  }
  const size_t rec = (size_t)c->aqPoolCap;
  if (q.wantCands && !c->candPool) {
    if (!c->candCap) c->candCap = c->aqEnv.candCap > 0 ? c->aqEnv.candCap : 1 << 18;   // (testing aid: a small pool forces the grow-and-repeat path)
    HIPCHK(c, hipHostMalloc(&c->candPool, sizeof(T4Cand) * (size_t)c->candCap, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->candPoolDev, c->candPool, 0));
  }
  q.tf0 = std::chrono::steady_clock::now();
  auto tl_ = q.tf0;
  auto lapL = [&](int i) { const auto t = std::chrono::steady_clock::now(); c->aqSecLaunch[i] += std::chrono::duration<double>(t - tl_).count(); tl_ = t; };
  if (q.wide) {
    if ((r = ensureWide(c, n, 1, 1))) return r;
    T4Wide w = wideHalf(c, 0), wa = wideHalf(c, 1);
    w.enabled = 1; w.safetyNum = q.wideSafety; w.samplePerPart = wa.samplePerPart = c->aqEnv.wideSample;
    // Reads of up to T4_WIDE_MIN_HITS emitted hits stay with one workgroup (LDS tier, then its slice of global scratch inside the same
    // launch, beside the other reads of the round): the wide query's kernels run behind the launch and cost a round about 0.25 ms
    // whatever the read is (profiles/r04b-h); from the LDS tier.s capacity on it beats one workgroup.s global scratch (C2 122 -> 93 s, profiles/r04h_*). The testing aid T4_AQ_CAP_LIMIT lowers
    // the threshold with the LDS tier's capacity.
    { const int lim = c->aqEnv.capLimit; w.minHits = lim > 0 ? lim : c->aqEnv.wideMinHits; }
    w.ctl = (int *)(c->aqOut + q.pWctl); w.plan = (T4WidePlan *)(c->aqOut + q.pWplan); w.stat = (int *)(c->aqOut + q.pWstat);
    memcpy(c->aqInHost + q.oWide, &w, sizeof w);
    wa.enabled = 1; wa.safetyNum = w.safetyNum; wa.minHits = w.minHits;
    wa.ctl = (int *)(c->aqOut + q.pWctlA); wa.plan = (T4WidePlan *)(c->aqOut + q.pWplanA); wa.stat = (int *)(c->aqOut + q.pWstatA);
    memcpy(c->aqInHost + q.oWideA, &wa, sizeof wa);
  }
  {
    T4CandArgs cs;
    memset(&cs, 0, sizeof cs);
    cs.stats8 = (int *)(c->aqOut + q.pS8);
    if (q.hasForce) cs.forceMin = (const int *)(c->aqIn + q.oForce);
    cs.useMarks = q.useMarks ? 1 : 0;
    if (q.wantCands) {
      cs.candOut = c->candPoolDev; cs.candCap = c->candCap; cs.candCursor = (unsigned *)(c->aqOut + q.pTail + 32); cs.candOverflow = (int *)(c->aqOut + q.pTail + 36);
      cs.candBase = (int *)(c->aqOut + q.pCb); cs.candCnt = (int *)(c->aqOut + q.pCc);
    }
    memcpy(c->aqInHost + q.oCs, &cs, sizeof cs);
  }
  lapL(0);
  {   // input blob into device memory, header block zeroed (counts, status, overflow lists, bases, tail): one kernel, no copy engine (t4_kernels.h: aqPrologueKernel)
    const unsigned long long inW = q.inBytes >> 3, outW = q.outBytes >> 3;
    const unsigned long long most = inW > outW ? inW : outW;
    int grid = (int)((most + 2047) / 2048);
    if (grid < 1) grid = 1;
    if (grid > c->cus * 4) grid = c->cus * 4;
    hipLaunchKernelGGL(t4k::aqPrologueKernel, dim3(grid), dim3(256), 0, c->stream, (const unsigned long long *)c->aqInHostDev, (unsigned long long *)c->aqIn, inW,
                       (unsigned long long *)c->aqOut, outW);
    HIPCHK(c, hipGetLastError());
  }
  lapL(1);
  T4BatchView &bv = q.bv;
  bv.pk = (const unsigned *)(c->aqIn + q.oPk); bv.nm = (const unsigned *)(c->aqIn + q.oNm); bv.len = (const int *)(c->aqIn + q.oLen);
  bv.barcode = (const int *)(c->aqIn + q.oBc); bv.wpk = q.wpk; bv.wnm = q.wnm; bv.n = n;
  T4QueryArgs &qa = q.qa;
  memset(&qa, 0, sizeof qa);
  qa.mode = 4; qa.skipRepeats = q.skipRepeats; qa.maxPerRead = 0;
  qa.counts = (int *)(c->aqOut + q.pCnt);
  qa.out = (T4OverlapOut *)c->aqPoolDev; qa.outExt = qa.out + rec; qa.ret = (int *)(qa.outExt + rec);
  qa.outBase = (int *)(c->aqOut + q.pBase); qa.poolCursor = (unsigned *)(c->aqOut + q.pTail + 24); qa.poolCap = c->aqPoolCap;
  qa.strandPerRead = (const int *)(c->aqIn + q.oSt); qa.factorPerRead = (const double *)(c->aqIn + q.oFa);
  qa.readTicks = (int *)(c->aqOut + q.pTick);
  qa.statsStable = (int *)(c->aqOut + q.pStab);
  qa.aux = (int *)(c->aqOut + q.pAux); qa.n4 = (int *)(c->aqOut + q.pN4);
  if (q.hasOnly) qa.onlySeq = (const int *)(c->aqIn + q.oOnly);
  {   // statistics words always; candidate records when asked for (the struct travels in the input blob: filled before the H2D copy above)
    qa.cs = (const T4CandArgs *)(c->aqIn + q.oCs);
  }
  qa.leanExt = q.lean ? 1 : 0;
  if (q.views) { qa.views = q.views; qa.viewOf = (const int *)(c->aqIn + q.oVw); }
  // one big set: the ExtendOverlap calls of reads with more than this many overlaps run in their own launch (0: never)
  int deferMin = c->aqEnv.extendDefer;   // testing aid
  if (q.wide && deferMin <= 0) deferMin = 16;   // the wide query leaves every ExtendOverlap to extendKernel
  const bool extendLater = deferMin > 0 && !q.views && !smallFirst && !q.onlyRestricted;
  q.extendLater = extendLater;
  if (extendLater) {
    if (c->aqRecCap < c->aqPoolCap) {
      if (c->aqRecDev) (void)hipFree(c->aqRecDev);
      if (c->aqRecRead) (void)hipFree(c->aqRecRead);
      c->aqRecDev = nullptr; c->aqRecRead = nullptr;
      HIPCHK(c, hipMalloc(&c->aqRecDev, sizeof(T4OverlapOut) * (size_t)c->aqPoolCap));
      HIPCHK(c, hipMalloc(&c->aqRecRead, sizeof(int) * (size_t)c->aqPoolCap));
      c->aqRecCap = c->aqPoolCap;
    }
    qa.extendLater = deferMin; qa.outDev = c->aqRecDev; qa.recRead = c->aqRecRead;
  }
  const int threads = !smallFirst ? 512 : 256;   // workgroup of the 8192-hit tier of this path
  q.threads = threads;
  const int grid0 = smallFirst ? (n < c->cus * TIER_BLOCKS_PER_CU[0] ? n : c->cus * TIER_BLOCKS_PER_CU[0])
                               : (nFirst > 0 ? (nFirst < c->cus * 2 ? nFirst : c->cus * 2) : 1);   // persistent grid: a large batch strides (and the per-block global scratch stays bounded)
  // scratch of the fallback DPs: the blocks of the LDS launch first, those of a concurrent global-tier launch behind them
  if ((r = ensureScratch(c, (grid0 > c->cus * 2 ? grid0 : c->cus * 2) * threads + (q.wide ? (nDirect > 0 ? c->cus * 512 : 0) : nDirect * G_THREADS)))) return r;
  T4Work &wk = q.wk;
  memset(&wk, 0, sizeof wk);
  wk.list = (const int *)(c->aqIn + q.oLs); wk.nList = nFirst;
  wk.nextList = (int *)(c->aqOut + q.pNext); wk.nextCount = (int *)(c->aqOut + q.pTail);
  wk.status = (int *)(c->aqOut + q.pSta); wk.hitCounter = (unsigned long long *)(c->aqOut + q.pTail + 16);
  wk.dpRows = c->dpRows; wk.dpDir = c->dpDir;
  wk.capLimit = c->aqEnv.capLimit;   // testing aid
  wk.wide = q.wide ? (const T4Wide *)(c->aqIn + q.oWide) : nullptr;
  if (!smallFirst && (r = ensureGlobalTier(c, grid0 + (q.wide ? 0 : nDirect)))) return r;   // before anything of this call runs: growing it frees the old arrays
  lapL(0);
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  lapL(5);
  if (nDirect > 0) {   // beside the LDS tier, on the second stream (its own blocks of the DP scratch)
    if (!c->stream2) { HIPCHK(c, hipStreamCreate(&c->stream2)); HIPCHK(c, hipEventCreate(&c->evIn)); HIPCHK(c, hipEventCreate(&c->evG)); }
    HIPCHK(c, hipEventRecord(c->evIn, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->evIn, 0));
    T4Work wd = wk;
    wd.list = (const int *)(c->aqIn + q.oLs) + nFirst; wd.nList = nDirect; wd.nextList = nullptr; wd.nextCount = nullptr;
    wd.dpRows = c->dpRows + (size_t)((grid0 > c->cus * 2 ? grid0 : c->cus * 2) * threads / 64) * (6 * T4_ROWW * 64);
    wd.dpDir = c->dpDir + (size_t)(grid0 > c->cus * 2 ? grid0 : c->cus * 2) * threads * T4_DIR_BYTES;
    if (q.wide) {
      // the wide query of the reads known to be heavy, beside the round's query kernel: seed stage, then the same five kernels on the
      // second half of the pools (wideHalf) and their own stretch of the DP scratch
      T4Wide wa;
      memcpy(&wa, c->aqInHost + q.oWideA, sizeof wa);
      const int cus = c->cus > 0 ? c->cus : 1;
      auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
      const int gParts = clampi(2 * c->wideRecentParts + 4 * nDirect + 4, 4, cus * 2), gReads = clampi(nDirect, 1, cus < 64 ? cus : 64);
      hipLaunchKernelGGL(t4k::wideSeedKernel, dim3(nDirect < cus ? nDirect : cus), dim3(512), 0, c->stream2, q.base, bv, wd, qa, wa, wd.list, nDirect);
      hipLaunchKernelGGL(t4k::wideScatterKernel, dim3(clampi(8 * gParts, 16, cus * 8)), dim3(256), 0, c->stream2, q.base, wa);
      hipLaunchKernelGGL((t4k::wideSortKernel<8192>), dim3(gParts), dim3(512), 0, c->stream2, q.base, wa);
      hipLaunchKernelGGL(t4k::wideStatsKernel, dim3(gReads), dim3(512), 0, c->stream2, q.base, wa);
      hipLaunchKernelGGL((t4k::wideChainKernel<8192, 1 << T4_WIDE_OVBITS>), dim3(gParts < cus ? gParts : cus), dim3(512), 0, c->stream2, q.base, bv, wd, qa, wa);
      hipLaunchKernelGGL(t4k::wideMergeKernel, dim3(gReads), dim3(512), 0, c->stream2, q.base, bv, wd, qa, wa);
    } else {
      wd.gKeys = c->gKeys; wd.gPairs = c->gPairs; wd.gCand = c->gCand; wd.gOv = c->gOv; wd.gFin = c->gFin; wd.gOrd = c->gOrd;
      wd.gCap = G_CAP; wd.gMaxOv = G_MAXOV;
      launchTier<0, 0, G_THREADS>(nDirect, c->stream2, q.base, bv, wd, qa);
      ++c->aqGlobalLaunches; c->aqGlobalReads += nDirect;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->evG, c->stream2));
  }
  lapL(2);
  if (nFirst > 0) {
    if (smallFirst) launchTier<1024, 128, 256>(grid0, c->stream, q.base, bv, wk, qa);
    else {   // a read that outgrows the LDS arrays goes on in global scratch inside the same launch
      T4Work wf = wk;
      const size_t skip = q.wide ? 0 : (size_t)nDirect;   // the blocks of a concurrent global-tier launch own the first slices
      wf.gKeys = c->gKeys + skip * G_CAP; wf.gPairs = c->gPairs + skip * G_CAP * 2; wf.gCand = c->gCand;
      wf.gOv = c->gOv + skip * G_MAXOV * 10; wf.gFin = c->gFin + skip * G_MAXOV * 10; wf.gOrd = c->gOrd + skip * G_MAXOV;
      wf.gCap = G_CAP; wf.gMaxOv = G_MAXOV;
      launchTier<8192, 512, 512>(grid0, c->stream, q.base, bv, wf, qa);
    }
    HIPCHK(c, hipGetLastError());
  }
  lapL(3);
  if (nDirect > 0 && !q.wide) HIPCHK(c, hipStreamWaitEvent(c->stream, c->evG, 0));
  q.lazyDone = false; q.lazyStage = false;
  if (q.wide) {
    // The reads the launch above deferred. Since round 5 a fresh heavy read is recognised on the host a round ahead and starts on the
    // second stream, so the query kernel defers a read in one whole-query round of twenty (config C2: 2 434 of 42 654): the five kernels
    // behind it were five empty grids in the others. They are launched when the header says that a read was deferred (aqEnd).
    if (c->aqEnv.wideEager) { aqLaunchDeferredWide(c, false); HIPCHK(c, hipGetLastError()); q.lazyDone = true; }
    if (nDirect > 0) HIPCHK(c, hipStreamWaitEvent(c->stream, c->evG, 0));   // the other pipeline's records are in the pool before the extensions run
  }
  if (extendLater) {   // all records of the batch, spread over the chip
    const int grid = c->cus * 16;
    hipLaunchKernelGGL(t4k::extendKernel, dim3(grid), dim3(64), 0, c->stream, q.base, bv, qa, 0);
    HIPCHK(c, hipGetLastError());
  }
  lapL(4);
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  lapL(5);
  r = aqLaunchEpilogue(c);
  lapL(6);
  return r;
}

// The call's results are on the host once the epilogue's word holds the call's number. The caller sits on the ordered chain's critical
// path: it polls (a blocking wait wakes up tens of microseconds after the kernels are done; a round is a few hundred). Every 2^16 polls
// -- a fraction of a millisecond -- the stream itself is asked, so that a kernel that died is an error and not a wait without end.
int aqWait(t4_ctx *c) {
  const unsigned want = c->aqSeq;
  for (unsigned long long spin = 1;; ++spin) {
    if (__atomic_load_n(c->aqFlagHost, __ATOMIC_ACQUIRE) == want) return T4_OK;
    if ((spin & 0xFFFFull) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {
        if (__atomic_load_n(c->aqFlagHost, __ATOMIC_ACQUIRE) == want) return T4_OK;
        return fail(c, T4_ERR_HIP, "the stream of an AddRead query call is idle but its epilogue never reported (word %u, expected %u)", *c->aqFlagHost, want);
      }
      if (e != hipErrorNotReady) return fail(c, T4_ERR_HIP, "AddRead query call: %s", hipGetErrorString(e));
    }
  }
}

// has the call in flight finished on the device? (1 yes or nothing in flight, 0 not yet)
int aqDone(t4_ctx *c) {
  if (!c->aq.active) return 1;
  return __atomic_load_n(c->aqFlagHost, __ATOMIC_ACQUIRE) == c->aqSeq ? 1 : 0;
}

int aqEnd(t4_ctx *c, AqResult *res) {
  AqCall &q = c->aq;
  if (!q.active) return fail(c, T4_ERR_STATE, "no AddRead query in flight on this ctx");
  q.active = false;
  (void)hipSetDevice(c->device);
  auto tNow = [] { return std::chrono::steady_clock::now(); };
  auto tSince = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  const int n = q.n;
  const bool smallFirst = q.smallFirst;
  const size_t pTail = q.pTail, pNext = q.pNext, pNext2 = q.pNext2, outBytes = q.outBytes;
  T4Work &wk = q.wk;
  T4QueryArgs &qa = q.qa;
  T4BatchView &bv = q.bv;
  int r;
  for (;;) {
    // The caller sits on the ordered chain's critical path: poll for the round's last event before the blocking wait (a wait that
    // sleeps wakes up tens of microseconds after the kernels are done; a round is a few hundred)
    if ((r = aqWait(c))) return r;
    // (everything queued on the stream before the epilogue is done: what a wait for the stream said before. The runtime is told now and
    // then too, so that it retires what it keeps per command.)
    if ((c->aqSeq & 255u) == 0) HIPCHK(c, hipStreamSynchronize(c->stream));
    ++c->syncEpoch;
    {
      float ms = 0;
      hipError_t ee = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
      if (ee == hipErrorNotReady) { (void)hipEventSynchronize(c->ev[1]); ee = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); }
      if (ee == hipSuccess) { c->aqKernelMs += ms; c->aqLastMs = q.lazyStage ? c->aqLastMs + ms : ms; }
    }
    if (q.wide && !q.lazyDone) {   // did the query kernel defer a read? then its wide query runs now, and the extensions of its records behind it
      q.lazyDone = true;
      const int *ctl0 = (const int *)(c->aqOutHost + q.pWctl);
      if (ctl0[0] > 0 && !ctl0[2] && !*(const int *)(c->aqOutHost + q.pWctlA + 8)) {
        const int recsBefore = (int)*(const unsigned *)(c->aqOutHost + pTail + 24);
        q.lazyStage = true;
        HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        aqLaunchDeferredWide(c, true);
        HIPCHK(c, hipGetLastError());
        if (q.extendLater) {
          hipLaunchKernelGGL(t4k::extendKernel, dim3(c->cus * 16), dim3(64), 0, c->stream, q.base, bv, qa, recsBefore < c->aqPoolCap ? recsBefore : c->aqPoolCap);
          HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        if ((r = aqLaunchEpilogue(c))) return r;
        continue;
      }
    }
    int overflow = *(int *)(c->aqOutHost + pTail);
    { const int inKernel = *(int *)(c->aqOutHost + pTail + 8); if (!smallFirst && inKernel > 0) { c->aqGlobalReads += inKernel; ++c->aqGlobalLaunches; } }
    c->aqSecFirst += tSince(q.tf0);
    auto tg0 = tNow();
    if (overflow > 0 && smallFirst) {   // reads beyond the 1024-hit tier: 8192-hit LDS tier
      T4Work w1 = wk;
      w1.list = (const int *)(c->aqOut + pNext); w1.nList = overflow;
      w1.nextList = (int *)(c->aqOut + pNext2); w1.nextCount = (int *)(c->aqOut + pTail + 8);
      if ((r = ensureScratch(c, (overflow > c->cus * 2 ? overflow : c->cus * 2) * q.threads))) return r;
      w1.dpRows = c->dpRows; w1.dpDir = c->dpDir;
      HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
      launchTier<8192, 512, 256>(overflow, c->stream, q.base, bv, w1, qa);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
      HIPCHK(c, hipMemcpyAsync(c->aqOutHost, c->aqOut, outBytes, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->aqKernelMs += ms; }
      overflow = *(int *)(c->aqOutHost + pTail + 8);
      wk.nextList = (int *)(c->aqOut + pNext2);
    }
    ++c->aqCalls; c->aqReads += n;
    if (overflow > 0 && q.tierHint) {   // remembered by the caller for the next query of these reads
      const int *lst = (const int *)(c->aqOutHost + (wk.nextList == (int *)(c->aqOut + pNext2) ? pNext2 : pNext));
      for (int t = 0; t < overflow; ++t) q.tierHint[lst[t]] = 1;
    }
    if (overflow > 0) {   // reads beyond the LDS tiers: global-scratch tier
      ++c->aqGlobalLaunches; c->aqGlobalReads += overflow;
      if ((r = ensureGlobalTier(c, overflow))) return r;
      if ((r = ensureScratch(c, (overflow > c->cus * 2 ? overflow : c->cus * 2) * G_THREADS))) return r;
      T4Work w2 = wk;
      w2.list = wk.nextList; w2.nList = overflow; w2.nextList = nullptr; w2.nextCount = nullptr;
      w2.gKeys = c->gKeys; w2.gPairs = c->gPairs; w2.gCand = c->gCand; w2.gOv = c->gOv; w2.gFin = c->gFin; w2.gOrd = c->gOrd;
      w2.gCap = G_CAP; w2.gMaxOv = G_MAXOV;
      w2.dpRows = c->dpRows; w2.dpDir = c->dpDir;
      HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
      const int recsBefore = (int)*(const unsigned *)(c->aqOutHost + pTail + 24);
      launchTier<0, 0, G_THREADS>(overflow, c->stream, q.base, bv, w2, qa);
      HIPCHK(c, hipGetLastError());
      if (q.extendLater) {
        hipLaunchKernelGGL(t4k::extendKernel, dim3(c->cus * 16), dim3(64), 0, c->stream, q.base, bv, qa, recsBefore < c->aqPoolCap ? recsBefore : c->aqPoolCap);
        HIPCHK(c, hipGetLastError());
      }
      HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
      HIPCHK(c, hipMemcpyAsync(c->aqOutHost, c->aqOut, outBytes, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->aqKernelMs += ms; }
    }
    c->aqSecGlobal += tSince(tg0);
    const unsigned char *o = c->aqOutHost;
    if (q.wide) {
      const int *ctl = (const int *)(o + q.pWctl), *ctlA = (const int *)(o + q.pWctlA);
      const int flags = ctl[2] | ctlA[2];
      if (flags) {   // a pool of the wide query ran out: larger pools / finer partitions, and the whole call again
        if (q.attempt >= 12 || (flags & (1 | 32))) return fail(c, T4_ERR_UNSUPPORTED, "wide query: flags %d after %d attempts", flags, q.attempt);
        if (flags & 2) {
          const int need = ctl[1] > ctlA[1] ? ctl[1] : ctlA[1];
          if (need <= c->wide.maxPart) return fail(c, T4_ERR_UNSUPPORTED, "a read of this batch needs more than %d partitions of %d k-mer hits", T4_WIDE_MAXP, c->wide.pcap);
          if ((r = ensureWide(c, n, need, 1))) return r;
        }
        for (int b = 0; b < 6; ++b) if (flags & (1 << b)) ++c->wideFlagCounts[b];
        if (flags & (4 | 8)) {
          if (q.wideSafety >= 32 * 256) return fail(c, T4_ERR_UNSUPPORTED, "wide query: a contig range of one contig overflows a partition");
          q.wideSafety *= 2;
          // keys: the partitions are quantiles of a SAMPLE of the hits' contigs; when a sample misjudged one, the next calls (the same
          // reads again, mostly) are planned less full as well. Overlaps of a partition: one read's matter (hundreds of short chains in
          // one contig range) -- finer partitions for this call only.
          if (flags & 4) { c->wideSafetyKeep = q.wideSafety < 128 ? q.wideSafety : 128; c->wideCallsSinceRepeat = 0; }
        }
        if (flags & 16) { if ((r = ensureWide(c, n, 1, ctl[3] > ctlA[3] ? ctl[3] : ctlA[3]))) return r; }
        ++c->wideRetries; ++q.attempt;
        if ((r = aqLaunch(c))) return r;
        continue;
      }
      c->wideReads += ctl[0] + ctlA[0]; c->wideParts += ctl[1] + ctlA[1]; c->wideGroups += ctl[3] + ctlA[3];
      ++c->wideCalls; if (ctl[0] > 0) ++c->wideCallsDeferred; if (ctlA[0] > 0) ++c->wideCallsDirect;
      if (ctl[0] + ctlA[0] > 0 && ++c->wideCallsSinceRepeat >= 2048 && c->wideSafetyKeep > 32) { c->wideSafetyKeep /= 2; c->wideCallsSinceRepeat = 0; }
      c->wideRecentParts = ctl[1] > c->wideRecentParts ? ctl[1] : (c->wideRecentParts * 7 + ctl[1]) / 8;
      c->wideRecentReads = ctl[0] > c->wideRecentReads ? ctl[0] : (c->wideRecentReads * 7 + ctl[0]) / 8;
      if (q.tierHint) {   // remembered by the caller: these reads start on the second stream the next time they are queried
        const T4WidePlan *plans[2] = {(const T4WidePlan *)(o + q.pWplan), (const T4WidePlan *)(o + q.pWplanA)};
        const int cnt[2] = {ctl[0] < n ? ctl[0] : n, ctlA[0] < n ? ctlA[0] : n};
        for (int hlf = 0; hlf < 2; ++hlf) for (int w2 = 0; w2 < cnt[hlf]; ++w2) if (plans[hlf][w2].read >= 0 && plans[hlf][w2].read < n) q.tierHint[plans[hlf][w2].read] = 1;
      }
    }
    const int *status = (const int *)(o + q.pSta);
    bool poolFull = false;
    for (int i = 0; i < n; ++i) {
      if (status[i] == 3) poolFull = true;
      else if (status[i] == 5) continue;   // a restricted re-query that one workgroup's arrays could not hold: reported to the caller (t4_add_query_last_aux)
      else if (status[i]) return fail(c, T4_ERR_UNSUPPORTED, "read %d exceeds the engine limits (status %d: %s)", i, status[i],
                                      status[i] == 2 ? "more k-mer hits or overlaps than the global tier holds" : "gap DP or contig count beyond scratch");
    }
    if (*(const unsigned *)(o + pTail + 28)) return fail(c, T4_ERR_UNSUPPORTED, "an overhang alignment of this batch exceeds the extension kernel's direction buffer");
    if (q.wantCands && *(const int *)(o + pTail + 36)) {   // more candidate records than their pool holds: a larger pool, and the whole call again
      if (q.attempt >= 12) return fail(c, T4_ERR_UNSUPPORTED, "candidate pool of %d records overflows", c->candCap);
      const unsigned want = *(const unsigned *)(o + pTail + 32);
      (void)hipHostFree(c->candPool);
      c->candPool = nullptr; c->candPoolDev = nullptr; ++c->candGrows;
      while ((unsigned)c->candCap < want && c->candCap < (1 << 29)) c->candCap *= 2;
      c->candCap *= 2;
      ++q.attempt;
      if ((r = aqLaunch(c))) return r;
      continue;
    }
    if (poolFull) {   // more result records than the pool holds: a larger pool, and the whole call again
      if (q.attempt >= 8) return fail(c, T4_ERR_UNSUPPORTED, "result pool of %d records overflows", c->aqPoolCap);
      (void)hipHostFree(c->aqPool);
      c->aqPool = nullptr; c->aqPoolDev = nullptr; c->aqPoolCap *= 4; ++c->aqPoolGrows;
      ++q.attempt;
      if ((r = aqLaunch(c))) return r;
      continue;
    }
    const size_t rec = (size_t)c->aqPoolCap;
    c->aqRecords += *(const unsigned *)(o + pTail + 24);
    if (q.wantCands) c->candRecords += *(const unsigned *)(o + pTail + 32);
    c->aqHits += (int64_t) * (const unsigned long long *)(o + pTail + 16);
    c->aqLastTicks = (const int32_t *)(o + q.pTick); c->aqLastN = n;
    c->aqLastStable = (const int32_t *)(o + q.pStab);
    res->counts = (const int32_t *)(o + q.pCnt); res->base = (const int32_t *)(o + q.pBase);
    res->ov = (const t4_overlap *)c->aqPool; res->ext = res->ov + rec; res->ret = (const int32_t *)(res->ext + rec);
    return T4_OK;
  }
}

int addQueryPool(t4_ctx *c, const T4IndexView &base, const T4IndexView *views, const int32_t *viewOf, bool smallFirst, int n, const char *bases,
                 const int64_t *offsets, const int32_t *barcodes, const int32_t *strands, int skip_repeats, const double *factors, AqResult *res,
                 unsigned char *tierHint = nullptr, bool lean = false) {
  int r = aqBegin(c, base, views, viewOf, smallFirst, n, bases, offsets, barcodes, strands, skip_repeats, factors, tierHint, lean);
  if (r) return r;
  return aqEnd(c, res);
}

// the same with the fixed-stride result layout of t4_add_query / t4_cellstore_query
int addQueryImpl(t4_ctx *c, const T4IndexView &base, const T4IndexView *views, const int32_t *viewOf, bool smallFirst, int n, const char *bases,
                 const int64_t *offsets, const int32_t *barcodes, const int32_t *strands, int skip_repeats, const double *factors,
                 int max_per_read, int32_t *counts, t4_overlap *ov, t4_overlap *ext, int32_t *ext_ret, bool lean = false) {
  AqResult res;
  int r = addQueryPool(c, base, views, viewOf, smallFirst, n, bases, offsets, barcodes, strands, skip_repeats, factors, &res, nullptr, lean);
  if (r) return r;
  for (int i = 0; i < n; ++i) {
    counts[i] = res.counts[i];
    if (res.counts[i] > max_per_read) return fail(c, T4_ERR_UNSUPPORTED, "read %d has %d overlaps, the caller made room for %d", i, res.counts[i], max_per_read);
    const int k2 = res.counts[i] > 0 ? res.counts[i] : 0;
    if (!k2) continue;
    const size_t at = (size_t)i * max_per_read;
    memcpy(ov + at, res.ov + res.base[i], sizeof(t4_overlap) * k2);
    memcpy(ext + at, res.ext + res.base[i], sizeof(t4_overlap) * k2);
    memcpy(ext_ret + at, res.ret + res.base[i], sizeof(int) * k2);
  }
  return T4_OK;
}
}  // namespace

extern "C" {

int t4_add_query_stats(t4_ctx *c, int64_t *out5) {   // 7 values
  if (!c || !out5) return T4_ERR_ARG;
  out5[0] = c->aqCalls; out5[1] = c->aqReads; out5[2] = c->aqGlobalLaunches; out5[3] = c->aqGlobalReads; out5[4] = c->aqRecords;
  out5[5] = (int64_t)(c->aqKernelMs * 1e3); out5[6] = c->aqHits;
  if (getenv("T4_TIMING")) fprintf(stderr, "timing: AddRead query path host seconds: pack %.3f, first launch to sync %.3f, overflow tiers %.3f; result pool grown %d times\n", c->aqSecPack, c->aqSecFirst, c->aqSecGlobal, c->aqPoolGrows);
  if (getenv("T4_TIMING")) fprintf(stderr, "timing: AddRead query path, host seconds inside the launch: preparation %.3f, prologue %.3f, second stream %.3f, query kernel %.3f, wide + extension %.3f, two event records %.3f, epilogue %.3f\n",
                                   c->aqSecLaunch[0], c->aqSecLaunch[1], c->aqSecLaunch[2], c->aqSecLaunch[3], c->aqSecLaunch[4], c->aqSecLaunch[5], c->aqSecLaunch[6]);
  if (getenv("T4_TIMING")) fprintf(stderr, "timing: wide query: %lld calls with the wide pipeline behind the query kernel, %lld of them had a read deferred to it; %lld with reads on the second stream\n", (long long)c->wideCalls, (long long)c->wideCallsDeferred, (long long)c->wideCallsDirect);
  if (getenv("T4_TIMING")) fprintf(stderr, "timing: wide query: %lld calls repeated -- partition pool %lld, keys of a partition %lld, overlaps of a partition %lld, dependency records %lld; partitions planned %d/16 full at the end\n",
                                   (long long)c->wideRetries, (long long)c->wideFlagCounts[1], (long long)c->wideFlagCounts[2], (long long)c->wideFlagCounts[3], (long long)c->wideFlagCounts[4], 16 * 16 / (c->wideSafetyKeep > 0 ? c->wideSafetyKeep : 32));
  return T4_OK;
}

// development aid (T4_ROUND_LOG): kernel milliseconds of the last AddRead query call on this ctx and, per read of that call, the
// microseconds one workgroup spent on it
int t4_add_query_last_call(t4_ctx *c, double *kernel_ms, const int32_t **ticks10ns, int *n) {
  if (!c) return T4_ERR_ARG;
  if (kernel_ms) *kernel_ms = c->aqLastMs;
  if (ticks10ns) *ticks10ns = c->aqLastTicks;
  if (n) *n = c->aqLastN;
  return T4_OK;
}

// Dependency records of read i of the last finished AddRead query call on this ctx, when the wide query served it (t4_wide.h):
// per (strand, contig) group its hits and the hull of its diagonals with three or more hits, minus-strand groups (even keys)
// first, ascending by contig within a strand. Returns 1 and the records (pinned memory, valid until the next call), 0 when the read
// was served by the LDS tier. huge: one of its lists holds more than 10000 postings; n4: groups of four or more hits.
int t4_add_query_groups(t4_ctx *c, int i, const t4_grp **groups, int *n, int *huge, int *n4) {
  if (!c || !groups || !n) return T4_ERR_ARG;
  const AqCall &q = c->aq;
  if (!q.wide || !c->aqOutHost) return 0;
  const unsigned char *o = c->aqOutHost;
  for (int half = 0; half < 2; ++half) {
    const int *ctl = (const int *)(o + (half ? q.pWctlA : q.pWctl));
    const T4WidePlan *plan = (const T4WidePlan *)(o + (half ? q.pWplanA : q.pWplan));
    const int *stat = (const int *)(o + (half ? q.pWstatA : q.pWstat));
    const int nw = ctl[0] < q.n ? ctl[0] : q.n;
    for (int w = 0; w < nw; ++w) {
      if (plan[w].read != i) continue;
      *groups = (const t4_grp *)c->grpPoolHost + (half ? (size_t)c->wide.grpCap : 0) + plan[w].grpBase;
      *n = stat[(size_t)w * T4_WIDE_STAT + WS_GROUPS];
      if (huge) *huge = plan[w].huge;
      if (n4) *n4 = stat[(size_t)w * T4_WIDE_STAT + WS_N4];
      return 1;
    }
  }
  return 0;
}

// per read of the last finished AddRead query call on this ctx (n entries, valid until the next call): aux and n4 as T4QueryArgs
// describes them, status (5: a restricted re-query that did not fit -- ask for the whole query)
int t4_add_query_last_aux(t4_ctx *c, const int32_t **aux, const int32_t **n4, const int32_t **status, int *n) {
  if (!c || !c->aqOutHost) return T4_ERR_ARG;
  const AqCall &q = c->aq;
  if (aux) *aux = (const int32_t *)(c->aqOutHost + q.pAux);
  if (n4) *n4 = (const int32_t *)(c->aqOutHost + q.pN4);
  if (status) *status = (const int32_t *)(c->aqOutHost + q.pSta);
  if (n) *n = q.n;
  return T4_OK;
}

// wide query of this ctx, 4 values: reads it served, partitions, calls repeated with larger pools, dependency records
int t4_add_query_wide_stats(t4_ctx *c, int64_t *out4) {
  if (!c || !out4) return T4_ERR_ARG;
  out4[0] = c->wideReads; out4[1] = c->wideParts; out4[2] = c->wideRetries; out4[3] = c->wideGroups;
  return T4_OK;
}

// per read of the last finished AddRead query call on this ctx: 1 when T4QueryArgs::statsStable says so (n entries, valid until the next call)
int t4_add_query_last_stable(t4_ctx *c, const int32_t **flags, int *n) {
  if (!c || !flags) return T4_ERR_ARG;
  *flags = c->aqLastStable;
  if (n) *n = c->aqLastN;
  return T4_OK;
}

int t4_add_query_pool(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                      int skip_repeats, const double *factors, const int32_t **counts, const int32_t **base, const t4_overlap **ov,
                      const t4_overlap **ext, const int32_t **ext_ret, unsigned char *tier_hint) {
  if (!ix || n <= 0 || !bases || !offsets || !strands || !factors || !counts || !base || !ov || !ext || !ext_ret) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  if (ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_add_query needs a contig set");
  AqResult res;
  int r = addQueryPool(c, ix->view, nullptr, nullptr, false, n, bases, offsets, barcodes, strands, skip_repeats, factors, &res, tier_hint, true);
  if (r) return r;
  *counts = res.counts; *base = res.base; *ov = res.ov; *ext = res.ext; *ext_ret = res.ret;
  return T4_OK;
}

// The two halves of t4_add_query_pool: begin enqueues the call on ix's ctx and returns; the host goes on; end waits and hands the
// result out (same lifetime rules). tier_hint must stay alive until end. One call in flight per ctx.
int t4_add_query_pool_begin(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                            int skip_repeats, const double *factors, unsigned char *tier_hint, const int32_t *only_seq) {
  if (!ix || n <= 0 || !bases || !offsets || !strands || !factors) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  if (ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_add_query needs a contig set");
  return aqBegin(c, ix->view, nullptr, nullptr, false, n, bases, offsets, barcodes, strands, skip_repeats, factors, tier_hint, true, only_seq);   // for t4_assembler: lean records (extendOverlaps)
}
int t4_add_query_pool_begin2(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                             int skip_repeats, const double *factors, unsigned char *tier_hint, const int32_t *only_seq, const int32_t *force_min, int want_cands) {
  if (!ix || n <= 0 || !bases || !offsets || !strands || !factors) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  if (ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_add_query needs a contig set");
  return aqBegin(c, ix->view, nullptr, nullptr, false, n, bases, offsets, barcodes, strands, skip_repeats, factors, tier_hint, true, only_seq, force_min, want_cands);
}
// per read of the last finished call begun with want_cands: its candidate records (cnt[i] of them from base[i] of pool, pinned
// memory valid until the next call) and its eight statistics words (T4QueryArgs::stats8)
int t4_add_query_last_cands(t4_ctx *c, const t4_cand **pool, const int32_t **base, const int32_t **cnt, const int32_t **stats8, int *n) {
  if (!c || !c->aqOutHost) return T4_ERR_ARG;
  const AqCall &q = c->aq;
  static_assert(sizeof(t4_cand) == sizeof(T4Cand) && sizeof(T4Cand) == 24 && T4_QUERY_STATS == T4_QSTATS, "candidate record layout");
  if (pool) *pool = q.wantCands ? (const t4_cand *)c->candPool : nullptr;
  if (base) *base = (const int32_t *)(c->aqOutHost + q.pCb);
  if (cnt) *cnt = q.wantCands ? (const int32_t *)(c->aqOutHost + q.pCc) : nullptr;
  if (stats8) *stats8 = (const int32_t *)(c->aqOutHost + q.pS8);
  if (n) *n = q.n;
  return T4_OK;
}
int t4_add_query_pool_done(t4_ctx *c) { return c ? aqDone(c) : 1; }
int t4_add_query_pool_end(t4_ctx *c, const int32_t **counts, const int32_t **base, const t4_overlap **ov, const t4_overlap **ext, const int32_t **ext_ret) {
  if (!c || !counts || !base || !ov || !ext || !ext_ret) return T4_ERR_ARG;
  AqResult res;
  int r = aqEnd(c, &res);
  if (r) return r;
  *counts = res.counts; *base = res.base; *ov = res.ov; *ext = res.ext; *ext_ret = res.ret;
  return T4_OK;
}

int t4_add_query(t4_index *ix, int n, const char *bases, const int64_t *offsets, const int32_t *barcodes, const int32_t *strands,
                 int skip_repeats, const double *factors, int max_per_read, int32_t *counts, t4_overlap *ov, t4_overlap *ext, int32_t *ext_ret) {
  if (!ix || n < 0 || max_per_read <= 0 || (n > 0 && (!bases || !offsets || !strands || !factors || !counts || !ov || !ext || !ext_ret))) return T4_ERR_ARG;
  t4_ctx *c = ix->ctx;
  if (!ix->committed) return fail(c, T4_ERR_STATE, "index not committed");
  if (ix->view.firstIsRef) return fail(c, T4_ERR_UNSUPPORTED, "t4_add_query needs a contig set");
  if (n == 0) return T4_OK;
  return addQueryImpl(c, ix->view, nullptr, nullptr, false, n, bases, offsets, barcodes, strands, skip_repeats, factors, max_per_read, counts, ov, ext, ext_ret);
}

}  // extern "C"

// ---- device arena of per-barcode set images ------------------------------------------------------------------
struct t4_cellstore {
  t4_ctx *ctx = nullptr;
  int k = 9, hitLenRequired = 31, radius = 10, nomatchGapLimit = 0;
  double novelSim = 0.9;
  struct Slot { unsigned char *base = nullptr; size_t cap = 0; bool live = false; bool pending = false; };
  std::vector<Slot> slots;
  std::vector<int> freeIds;
  std::vector<unsigned char *> chunks;
  size_t chunkUsed = 0;
  std::map<size_t, std::vector<unsigned char *>> freeBySize;
  T4IndexView *dViews = nullptr;
  int viewCap = 0;
  unsigned char *stHost = nullptr, *stDev = nullptr;   // staging of the images rebuilt since the last flush
  size_t stCap = 0, stUsed = 0;
  std::vector<T4CopyDesc> descs;
  T4CopyDesc *dDescs = nullptr;
  size_t descCap = 0;
  std::vector<T4BytePatch> patches;
  T4BytePatch *dPatches = nullptr;
  size_t patchCap = 0;
  int64_t bytesPatched = 0;
  int64_t bytesStaged = 0;
  bool bigFirst = false;   // one big set (bulk mode): reads meet hundreds of contigs, start on the 8192-hit tier
  std::mutex mu;   // t4_cellstore_stage may run on several host threads after t4_cellstore_prepare
  static constexpr size_t CHUNK = (size_t)256 << 20;
};

namespace {
int cellAlloc(t4_cellstore *cs, size_t cap, unsigned char **out) {
  auto it = cs->freeBySize.find(cap);
  if (it != cs->freeBySize.end() && !it->second.empty()) { *out = it->second.back(); it->second.pop_back(); return T4_OK; }
  if (cap > t4_cellstore::CHUNK) return fail(cs->ctx, T4_ERR_UNSUPPORTED, "a per-barcode set image of %zu bytes exceeds the arena chunk", cap);
  if (cs->chunks.empty() || cs->chunkUsed + cap > t4_cellstore::CHUNK) {
    unsigned char *p = nullptr;
    HIPCHK(cs->ctx, hipMalloc(&p, t4_cellstore::CHUNK));
    cs->chunks.push_back(p); cs->chunkUsed = 0;
  }
  *out = cs->chunks.back() + cs->chunkUsed;
  cs->chunkUsed += cap;
  return T4_OK;
}
int cellStagingReserve(t4_cellstore *cs, size_t more) {
  if (cs->stUsed + more <= cs->stCap) return T4_OK;
  size_t ncap = cs->stCap ? cs->stCap : ((size_t)4 << 20);
  while (ncap < cs->stUsed + more) ncap *= 2;
  unsigned char *nh = nullptr, *nd = nullptr;
  HIPCHK(cs->ctx, hipHostMalloc(&nh, ncap, hipHostMallocDefault));
  HIPCHK(cs->ctx, hipMalloc(&nd, ncap));
  if (cs->stUsed) memcpy(nh, cs->stHost, cs->stUsed);
  if (cs->stHost) (void)hipHostFree(cs->stHost);
  if (cs->stDev) { (void)hipStreamSynchronize(cs->ctx->stream); (void)hipFree(cs->stDev); }
  cs->stHost = nh; cs->stDev = nd; cs->stCap = ncap;
  return T4_OK;
}
int cellFlushPatches(t4_cellstore *cs) {
  if (cs->patches.empty()) return T4_OK;
  t4_ctx *c = cs->ctx;
  if (cs->patches.size() > cs->patchCap) {
    if (cs->dPatches) { (void)hipStreamSynchronize(c->stream); (void)hipFree(cs->dPatches); cs->dPatches = nullptr; }
    cs->patchCap = cs->patches.size() * 2;
    HIPCHK(c, hipMalloc(&cs->dPatches, sizeof(T4BytePatch) * cs->patchCap));
  }
  HIPCHK(c, hipMemcpyAsync(cs->dPatches, cs->patches.data(), sizeof(T4BytePatch) * cs->patches.size(), hipMemcpyHostToDevice, c->stream));
  const int n = (int)cs->patches.size();
  int grid = (n + 255) / 256;
  if (grid > c->cus * 4) grid = c->cus * 4;
  hipLaunchKernelGGL(t4k::patchKernel, dim3(grid), dim3(256), 0, c->stream, (const T4BytePatch *)cs->dPatches, n);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the host vector is reused
  cs->bytesPatched += n;
  cs->patches.clear();
  return T4_OK;
}
int cellFlush(t4_cellstore *cs) {
  if (cs->descs.empty()) return cellFlushPatches(cs);
  t4_ctx *c = cs->ctx;
  if (cs->descs.size() > cs->descCap) {
    if (cs->dDescs) { (void)hipStreamSynchronize(c->stream); (void)hipFree(cs->dDescs); cs->dDescs = nullptr; }
    cs->descCap = cs->descs.size() * 2;
    HIPCHK(c, hipMalloc(&cs->dDescs, sizeof(T4CopyDesc) * cs->descCap));
  }
  HIPCHK(c, hipMemcpyAsync(cs->stDev, cs->stHost, cs->stUsed, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(cs->dDescs, cs->descs.data(), sizeof(T4CopyDesc) * cs->descs.size(), hipMemcpyHostToDevice, c->stream));
  int grid = (int)cs->descs.size();
  if (grid > c->cus * 8) grid = c->cus * 8;
  hipLaunchKernelGGL(t4k::scatterKernel, dim3(grid), dim3(256), 0, c->stream, (const unsigned char *)cs->stDev, (const T4CopyDesc *)cs->dDescs, (int)cs->descs.size());
  HIPCHK(c, hipGetLastError());
  // the pageable descriptor vector and the pinned staging are reused by the next stage() calls
  HIPCHK(c, hipStreamSynchronize(c->stream));
  cs->bytesStaged += (int64_t)cs->stUsed;
  cs->descs.clear(); cs->stUsed = 0;
  for (t4_cellstore::Slot &sl : cs->slots) sl.pending = false;
  return cellFlushPatches(cs);
}
}  // namespace

extern "C" {

int t4_cellstore_create(t4_ctx *c, int k, t4_cellstore **out) {
  if (!c || !out) return T4_ERR_ARG;
  if (k < 2 || k > 31) return fail(c, T4_ERR_ARG, "kmer_length %d outside [2,31]", k);
  t4_cellstore *cs = new t4_cellstore();
  cs->ctx = c; cs->k = k;
  double kmerHitProb = pow(0.8, k);  // SeqSet::ComputeNomatchGapLimit (SeqSet.hpp:2476-2482)
  cs->nomatchGapLimit = int(k * (log(0.01) / log(1 - kmerHitProb))) + 1;
  *out = cs;
  return T4_OK;
}
void t4_cellstore_destroy(t4_cellstore *cs) {
  if (!cs) return;
  (void)hipSetDevice(cs->ctx->device);
  (void)hipStreamSynchronize(cs->ctx->stream);
  for (unsigned char *p : cs->chunks) (void)hipFree(p);
  if (cs->dViews) (void)hipFree(cs->dViews);
  if (cs->stDev) (void)hipFree(cs->stDev);
  if (cs->stHost) (void)hipHostFree(cs->stHost);
  if (cs->dDescs) (void)hipFree(cs->dDescs);
  if (cs->dPatches) (void)hipFree(cs->dPatches);
  delete cs;
}
int t4_cellstore_set_params(t4_cellstore *cs, int hit_len_required, int radius, double novel_sim) {
  if (!cs) return T4_ERR_ARG;
  cs->hitLenRequired = hit_len_required; cs->radius = radius; cs->novelSim = novel_sim;
  return T4_OK;
}
int t4_cellstore_open(t4_cellstore *cs, int *slot) {
  if (!cs || !slot) return T4_ERR_ARG;
  if (!cs->freeIds.empty()) { *slot = cs->freeIds.back(); cs->freeIds.pop_back(); }
  else { *slot = (int)cs->slots.size(); cs->slots.push_back(t4_cellstore::Slot()); }
  cs->slots[*slot].live = true;
  return T4_OK;
}
int t4_cellstore_close(t4_cellstore *cs, int slot) {
  if (!cs || slot < 0 || slot >= (int)cs->slots.size() || !cs->slots[slot].live) return T4_ERR_ARG;
  t4_cellstore::Slot &s = cs->slots[slot];
  if (s.base) cs->freeBySize[s.cap].push_back(s.base);
  s = t4_cellstore::Slot();
  cs->freeIds.push_back(slot);
  return T4_OK;
}

size_t t4_cellstore_image_bytes(int nseq, int64_t nkeys, int64_t npost, int64_t cons_bytes) {
  auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t sz = 64;
  while (2 * sz < 3 * (size_t)nkeys + 2) sz <<= 1;   // load factor <= 2/3
  const size_t pwCount = (size_t)cons_bytes;   // one predicate byte per consensus byte incl. the terminators
  const size_t oPost = al16(sizeof(T4HashEntC) * sz), oSeq = al16(oPost + sizeof(int2) * (size_t)npost),
               oPw = al16(oSeq + sizeof(T4SeqInfo) * (size_t)nseq), oCons = al16(oPw + sizeof(T4PW) * pwCount);
  return al16(oCons + (size_t)cons_bytes + 16) + al16(sizeof(T4IndexView));
}

// Serial step before a group of (possibly concurrent) t4_cellstore_stage calls: room for `bytes` more staged bytes and a
// view entry for every slot id up to max_slot, so that no buffer moves while images are being written.
int t4_cellstore_prepare(t4_cellstore *cs, int max_slot, size_t bytes) {
  if (!cs) return T4_ERR_ARG;
  t4_ctx *c = cs->ctx;
  (void)hipSetDevice(c->device);
  int r;
  if (max_slot >= cs->viewCap) {
    int ncap = cs->viewCap ? cs->viewCap : 1024;
    while (ncap <= max_slot) ncap *= 2;
    T4IndexView *nv = nullptr;
    HIPCHK(c, hipMalloc(&nv, sizeof(T4IndexView) * (size_t)ncap));
    if ((r = cellFlush(cs))) return r;   // pending descriptors may point into the old array
    if (cs->dViews) {
      HIPCHK(c, hipMemcpyAsync(nv, cs->dViews, sizeof(T4IndexView) * (size_t)cs->viewCap, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(cs->dViews);
    }
    cs->dViews = nv; cs->viewCap = ncap;
  }
  return cellStagingReserve(cs, bytes);
}

int t4_cellstore_stage(t4_cellstore *cs, int slot, int barcode, int nseq, const char *const *names, const char *const *cons,
                       const int32_t *const *pw, int64_t nkeys64, const uint64_t *keyCode, const int32_t *keyBucket, const int32_t *keyCnt,
                       const int32_t *postIn, int64_t *pwOffsetInImage, const int32_t *seqBarcodes) {
  if (!cs || slot < 0 || slot >= (int)cs->slots.size() || !cs->slots[slot].live || nseq < 0 || nkeys64 < 0) return T4_ERR_ARG;
  t4_ctx *c = cs->ctx;
  (void)hipSetDevice(c->device);
  if (nseq > T4_MAX_SEQS) return fail(c, T4_ERR_UNSUPPORTED, "more than %d sequences in one barcode", T4_MAX_SEQS);
  const size_t nkeys = (size_t)nkeys64;
  int64_t npost = 0;
  for (size_t i = 0; i < nkeys; ++i) npost += keyCnt[i];
  size_t sz = 64;
  while (2 * sz < 3 * nkeys + 2) sz <<= 1;   // load factor <= 2/3
  size_t consBytes = 0, pwCount = 0;
  for (int i = 0; i < nseq; ++i) {
    size_t l = strlen(cons[i]);
    if (l > T4_MAX_SEQLEN) return fail(c, T4_ERR_UNSUPPORTED, "contig longer than %d", T4_MAX_SEQLEN);
    consBytes += l + 1; pwCount += l + 1;
  }
  auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t oHt = 0, oPost = al16(oHt + sizeof(T4HashEntC) * sz), oSeq = al16(oPost + sizeof(int2) * (size_t)npost),
               oPw = al16(oSeq + sizeof(T4SeqInfo) * (size_t)nseq), oCons = al16(oPw + sizeof(T4PW) * pwCount),
               blobBytes = al16(oCons + consBytes + 16);
  int r;
  const size_t viewBytes = al16(sizeof(T4IndexView));
  unsigned char *b = nullptr, *slotBase = nullptr;
  size_t stOff = 0;
  {
    std::lock_guard<std::mutex> lock(cs->mu);
    t4_cellstore::Slot &sl = cs->slots[slot];
    // two images of one slot in the same flush would be scattered concurrently: the caller must launch (flush) in between
    if (sl.pending) return fail(c, T4_ERR_STATE, "slot %d already has an image staged for the next launch", slot);
    sl.pending = true;
    if (blobBytes > sl.cap) {
      if (sl.base) cs->freeBySize[sl.cap].push_back(sl.base);
      size_t cap = (size_t)64 << 10;
      while (cap < blobBytes + blobBytes / 2) cap <<= 1;
      sl.base = nullptr; sl.cap = 0;
      unsigned char *p = nullptr;
      if ((r = cellAlloc(cs, cap, &p))) return r;
      sl.base = p; sl.cap = cap;
    }
    slotBase = sl.base;
    if (slot >= cs->viewCap || cs->stUsed + blobBytes + viewBytes > cs->stCap)
      return fail(c, T4_ERR_STATE, "t4_cellstore_stage without a sufficient t4_cellstore_prepare (slot %d, %zu bytes)", slot, blobBytes + viewBytes);
    stOff = cs->stUsed;
    b = cs->stHost + stOff;
    T4CopyDesc d0; d0.srcOff = stOff; d0.dst = slotBase; d0.bytes = blobBytes;
    T4CopyDesc d1; d1.srcOff = stOff + blobBytes; d1.dst = (unsigned char *)(cs->dViews + slot); d1.bytes = viewBytes;
    cs->descs.push_back(d0); cs->descs.push_back(d1);
    cs->stUsed += blobBytes + viewBytes;
  }
  memset(b, 0, blobBytes + viewBytes);
  T4HashEntC *ht = (T4HashEntC *)(b + oHt);
  for (size_t t = 0; t < sz; ++t) ht[t].code = ~0ull;   // empty slots
  int2 *post = (int2 *)(b + oPost);
  const unsigned long long hashMask = sz - 1;
  size_t at = 0;
  for (size_t i = 0; i < nkeys; ++i) {
    const unsigned long long cd = keyCode[i];
    const int hb = keyBucket[i], cnt = keyCnt[i];
    for (int j = 0; j < cnt; ++j) {
      const int id = postIn[2 * (at + j)];
      if (id < 0 || id >= nseq) return fail(c, T4_ERR_ARG, "posting names sequence %d of %d", id, nseq);
      post[at + j] = make_int2(id, postIn[2 * (at + j) + 1]);
    }
    if (cnt <= 0) continue;
    if (hb != (int)((cd + (unsigned long long)(long long)(barcode + 1)) % 1000003ull)) return fail(c, T4_ERR_ARG, "key of another barcode in the image of barcode %d", barcode);
    unsigned long long s = t4k::mix64(cd) & hashMask;
    while (ht[s].code != ~0ull) s = (s + 1) & hashMask;
    ht[s].code = cd; ht[s].start = (unsigned)at; ht[s].cnt = (unsigned)cnt;
    at += (size_t)cnt;
  }
  T4SeqInfo *infos = (T4SeqInfo *)(b + oSeq);
  T4PW *pwOut = (T4PW *)(b + oPw);
  char *consOut = (char *)(b + oCons);
  size_t consAt = 0, pwAt = 0;
  for (int i = 0; i < nseq; ++i) {
    T4SeqInfo &f = infos[i];
    const int l = (int)strlen(cons[i]);
    f.consOff = (int)consAt; f.len = l; f.barcode = seqBarcodes ? seqBarcodes[i] : barcode; f.isRef = 0;
    memcpy(consOut + consAt, cons[i], (size_t)l + 1);
    consAt += (size_t)l + 1;
    char nm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    strncpy(nm, names[i], 7);
    int gt = geneType(nm);
    f.geneType = gt < 0 ? 255 : (unsigned char)gt;
    f.name0 = (unsigned char)nm[0]; f.name1 = (unsigned char)nm[1]; f.name2 = (unsigned char)nm[2]; f.name3 = (unsigned char)nm[3];
    f.pwOff = (int)pwAt;
    if (l > 0 && pw[i]) for (int t = 0; t < l; ++t) pwOut[pwAt + t] = t4PwByte(pw[i][4 * t], pw[i][4 * t + 1], pw[i][4 * t + 2], pw[i][4 * t + 3]);
    for (size_t t = (l > 0 && pw[i]) ? (size_t)l : 0; t <= (size_t)l; ++t) pwOut[pwAt + t] = t4PwByte(0, 0, 0, 0);
    pwAt += (size_t)l + 1;
  }
  if (pwOffsetInImage) *pwOffsetInImage = (int64_t)oPw;
  T4IndexView &v = *(T4IndexView *)(b + blobBytes);
  v.k = cs->k; v.nseq = nseq; v.direct = 2; v.considerBarcode = 1;
  v.hashMask = hashMask; v.table = nullptr; v.htab = nullptr; v.ctab = (const T4HashEntC *)(slotBase + oHt); v.post = (const int2 *)(slotBase + oPost);
  v.seqs = (const T4SeqInfo *)(slotBase + oSeq); v.cons = (const char *)(slotBase + oCons); v.pw = (const T4PW *)(slotBase + oPw);
  v.radius = cs->radius; v.hitLenRequired = cs->hitLenRequired; v.nomatchGapLimit = cs->nomatchGapLimit;
  v.firstIsRef = 0; v.hasNovel = 2;
  { int maxLen = 0; for (int i = 0; i < nseq; ++i) if (infos[i].len > maxLen) maxLen = infos[i].len; v.key32 = t4Key32Bits(nseq, maxLen); }
  v.novelSim = cs->novelSim; v.refSim = 0.75; v.repeatSim = 0.95;
  return T4_OK;
}

int t4_cellstore_patch(t4_cellstore *cs, int slot, int n, const int64_t *byteOffsets, const unsigned char *values) {
  if (!cs || slot < 0 || slot >= (int)cs->slots.size() || !cs->slots[slot].base || n < 0 || (n > 0 && (!byteOffsets || !values))) return T4_ERR_ARG;
  const t4_cellstore::Slot &sl = cs->slots[slot];
  for (int i = 0; i < n; ++i) {
    if (byteOffsets[i] < 0 || (size_t)byteOffsets[i] >= sl.cap) return fail(cs->ctx, T4_ERR_ARG, "patch outside the image of slot %d", slot);
    T4BytePatch p; p.dst = sl.base + byteOffsets[i]; p.val = values[i];
    cs->patches.push_back(p);
  }
  return T4_OK;
}

int t4_cellstore_query(t4_cellstore *cs, int n, const int32_t *slots, const char *bases, const int64_t *offsets, const int32_t *barcodes,
                       const int32_t *strands, int skip_repeats, const double *factors, int max_per_read, int32_t *counts,
                       t4_overlap *ov, t4_overlap *ext, int32_t *ext_ret) {
  if (!cs || n < 0 || max_per_read <= 0 || (n > 0 && (!slots || !bases || !offsets || !strands || !factors || !counts || !ov || !ext || !ext_ret))) return T4_ERR_ARG;
  t4_ctx *c = cs->ctx;
  (void)hipSetDevice(c->device);
  int r;
  if ((r = cellFlush(cs))) return r;
  if (n == 0) return T4_OK;
  for (int i = 0; i < n; ++i)
    if (slots[i] < 0 || slots[i] >= (int)cs->slots.size() || !cs->slots[slots[i]].base) return fail(c, T4_ERR_STATE, "read %d names cell slot %d which has no image", i, slots[i]);
  T4IndexView base;
  memset(&base, 0, sizeof base);
  base.k = cs->k;
  return addQueryImpl(c, base, cs->dViews, slots, !cs->bigFirst, n, bases, offsets, barcodes, strands, skip_repeats, factors, max_per_read, counts, ov, ext, ext_ret, true);   // consumed by t4_assembler::addRead only: lean records
}
int t4_cellstore_set_big_first(t4_cellstore *cs, int on) { if (!cs) return T4_ERR_ARG; cs->bigFirst = on != 0; return T4_OK; }
int64_t t4_cellstore_bytes_staged(const t4_cellstore *cs) { return cs ? cs->bytesStaged + cs->bytesPatched : 0; }

}  // extern "C"

// ---- t4_comm: RCCL all-gather of byte strings between the ranks of one node (one process per GPU) -------------------------------
#ifdef __HIPCC__
// RCCL is bound when the first communicator is asked for (dlopen), not when libt4hip.so is loaded: a single-GPU host without
// librccl still loads the library and runs everything but t4_comm. <rccl/rccl.h> gives the types only.
struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclCommAbort) commAbort = nullptr;
  decltype(&ncclCommGetAsyncError) commGetAsyncError = nullptr;
  decltype(&ncclAllGather) allGather = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  std::string error;
};
static RcclApi *rcclApi() {
  static std::mutex mu;
  static RcclApi api;
  std::lock_guard<std::mutex> lk(mu);
  if (api.lib || !api.error.empty()) return &api;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void *h = nullptr;
  for (const char *nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) { const char *e = dlerror(); api.error = std::string("librccl.so not found (") + (e ? e : "dlopen failed") + ")"; return &api; }
  bool ok = true;
  auto bind = [&](auto &fp, const char *sym) { fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(h, sym)); if (!fp) { ok = false; api.error = std::string("librccl.so lacks ") + sym; } };
  bind(api.getUniqueId, "ncclGetUniqueId"); bind(api.commInitRank, "ncclCommInitRank"); bind(api.commDestroy, "ncclCommDestroy"); bind(api.commAbort, "ncclCommAbort"); bind(api.commGetAsyncError, "ncclCommGetAsyncError");
  bind(api.allGather, "ncclAllGather"); bind(api.send, "ncclSend"); bind(api.recv, "ncclRecv");
  bind(api.groupStart, "ncclGroupStart"); bind(api.groupEnd, "ncclGroupEnd");
  if (!ok) { dlclose(h); return &api; }
  api.lib = h;
  return &api;
}
#endif

struct t4_comm {
  t4_ctx *ctx = nullptr;
  int rank = 0, nranks = 1;
#ifdef __HIPCC__
  ncclComm_t comm = nullptr;
  RcclApi *rccl = nullptr;
#endif
};

#ifdef __HIPCC__
// The wait of a collective, bounded: a rank that died before it leaves the others in the kernel RCCL enqueued for ever.
// Polls the stream; an asynchronous RCCL error or T4_COMM_TIMEOUT_S seconds (default 3600: ranks of a sharded sample may finish far
// apart) abort the communicator and come back as an error instead of a hang.
static int commWait(t4_comm *cm, const char *what, double limitS = 0) {
  t4_ctx *c = cm->ctx;
  const char *ev = getenv("T4_COMM_TIMEOUT_S");
  const double limit = limitS > 0 ? limitS : (ev && atof(ev) > 0 ? atof(ev) : 3600.0);
  const auto t0 = std::chrono::steady_clock::now();
  for (long long spin = 0;; ++spin) {
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) return T4_OK;
    if (q != hipErrorNotReady) return fail(c, T4_ERR_HIP, "t4_comm (%s): %s", what, hipGetErrorString(q));
    ncclResult_t ar = ncclSuccess;
    const bool asyncBad = cm->comm && (spin & 1023) == 1023 && cm->rccl->commGetAsyncError(cm->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress;
    const bool late = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit;
    if (asyncBad || late) {
      if (cm->comm) { (void)cm->rccl->commAbort(cm->comm); cm->comm = nullptr; }
      return fail(c, T4_ERR_HIP, late ? "t4_comm (%s): rank %d of %d waited %.0f s for the other ranks (T4_COMM_TIMEOUT_S); communicator aborted" : "t4_comm (%s): RCCL reported an asynchronous error on rank %d of %d (%.0f s); communicator aborted",
                  what, cm->rank, cm->nranks, limit);
    }
    if (spin < 2000) std::this_thread::yield(); else usleep(200);
  }
}
#endif

#ifdef __HIPCC__
namespace {
// device / host buffers of one collective: released on every path out of it (the error returns leaked them)
struct CommBufs {
  std::vector<void *> dev; void *host = nullptr;
  ~CommBufs() { for (void *p : dev) if (p) (void)hipFree(p); if (host) free(host); }
  template <class T> hipError_t devAlloc(T **p, size_t bytes) { *p = nullptr; const hipError_t e = hipMalloc((void **)p, bytes ? bytes : 8); if (e == hipSuccess) dev.push_back(*p); return e; }
  void *releaseHost() { void *h = host; host = nullptr; return h; }
};
// lengths of every rank's contribution: one 8-byte ncclAllGather. The bounded wait comes BEFORE any device-to-host copy is queued: a
// copy into pageable memory runs synchronously behind the RCCL kernel, and with a dead peer the host would hang inside hipMemcpyAsync
// and never reach commWait.
int commLengths(t4_comm *cm, int64_t n, std::vector<unsigned long long> &lens) {
  t4_ctx *c = cm->ctx;
  const int R = cm->nranks;
  CommBufs bufs;
  unsigned long long *dLen = nullptr;
  HIPCHK(c, bufs.devAlloc(&dLen, sizeof(unsigned long long) * (size_t)(R + 1)));
  const unsigned long long myLen = (unsigned long long)n;
  HIPCHK(c, hipMemcpy(dLen + R, &myLen, sizeof myLen, hipMemcpyHostToDevice));
  if (cm->rccl->allGather(dLen + R, dLen, 1, ncclUint64, cm->comm, c->stream) != ncclSuccess) return fail(c, T4_ERR_HIP, "ncclAllGather (lengths) failed");
  if (int wr = commWait(cm, "the lengths")) return wr;
  lens.resize((size_t)R);
  HIPCHK(c, hipMemcpy(lens.data(), dLen, sizeof(unsigned long long) * (size_t)R, hipMemcpyDeviceToHost));
  return T4_OK;
}
}  // namespace
#endif

extern "C" {

int t4_comm_init(t4_ctx *c, int rank, int nranks, const char *id_path, t4_comm **out) {
  if (!c || !out || nranks < 1 || rank < 0 || rank >= nranks || !id_path) return T4_ERR_ARG;
#ifdef __HIPCC__
  (void)hipSetDevice(c->device);
  RcclApi *rccl = rcclApi();
  if (!rccl->lib) return fail(c, T4_ERR_UNSUPPORTED, "t4_comm: %s", rccl->error.c_str());
  ncclUniqueId id;
  if (rank == 0) {
    (void)unlink(id_path);   // (an id left by an earlier run must not be taken for this one's)
    if (rccl->getUniqueId(&id) != ncclSuccess) return fail(c, T4_ERR_HIP, "ncclGetUniqueId failed");
    const std::string tmp = std::string(id_path) + ".tmp";
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp || fwrite(&id, sizeof id, 1, fp) != 1) { if (fp) fclose(fp); return fail(c, T4_ERR_IO, "cannot write %s", tmp.c_str()); }
    fclose(fp);
    if (rename(tmp.c_str(), id_path)) return fail(c, T4_ERR_IO, "cannot create %s", id_path);
  } else {
    bool got = false;
    for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to ten minutes: rank 0 may still be parsing its input
      FILE *fp = fopen(id_path, "rb");
      if (fp) { got = fread(&id, sizeof id, 1, fp) == 1; fclose(fp); }
      if (!got) usleep(100000);
    }
    if (!got) return fail(c, T4_ERR_IO, "no communicator id appeared at %s", id_path);
  }
  t4_comm *cm = new t4_comm();
  cm->ctx = c; cm->rank = rank; cm->nranks = nranks; cm->rccl = rccl;
  if (rccl->commInitRank(&cm->comm, nranks, id, rank) != ncclSuccess) { delete cm; return fail(c, T4_ERR_HIP, "ncclCommInitRank(%d of %d) failed", rank, nranks); }
  // Self-check before anything depends on the communicator (RCCL had only ever run with one rank): every rank
  // contributes its number, everybody must receive 0..N-1 in order, within T4_COMM_INIT_TIMEOUT_S (default 120 s).
  {
    CommBufs bufs;
    int *d = nullptr;
    bool ok = bufs.devAlloc(&d, sizeof(int) * (size_t)(nranks + 1)) == hipSuccess && hipMemcpy(d + nranks, &rank, sizeof rank, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && rccl->allGather(d + nranks, d, 1, ncclInt32, cm->comm, c->stream) == ncclSuccess;
    const char *lim = getenv("T4_COMM_INIT_TIMEOUT_S");
    ok = ok && commWait(cm, "the self-check", lim && atof(lim) > 0 ? atof(lim) : 120.0) == T4_OK;
    std::vector<int> got((size_t)nranks, -1);
    ok = ok && hipMemcpy(got.data(), d, sizeof(int) * (size_t)nranks, hipMemcpyDeviceToHost) == hipSuccess;
    for (int r = 0; ok && r < nranks; ++r) ok = got[(size_t)r] == r;
    if (!ok) {
      if (cm->comm) (void)rccl->commAbort(cm->comm);
      delete cm;
      return fail(c, T4_ERR_HIP, "t4_comm: the self-check of the new communicator failed on rank %d of %d (a 1-int all-gather did not return the rank numbers)", rank, nranks);
    }
  }
  *out = cm;
  return T4_OK;
#else
  (void)out;
  return fail(c, T4_ERR_UNSUPPORTED, "t4_comm needs the hipcc build (RCCL)");
#endif
}


int t4_comm_allgather_bytes(t4_comm *cm, const void *mine, int64_t n, void **all, int64_t *sizes) {
  if (!cm || n < 0 || (n > 0 && !mine) || !all || !sizes) return T4_ERR_ARG;
#ifdef __HIPCC__
  t4_ctx *c = cm->ctx;
  if (!cm->comm) return fail(c, T4_ERR_ARG, "t4_comm: the communicator was aborted");
  (void)hipSetDevice(c->device);
  const int R = cm->nranks;
  // lengths first (8 bytes per rank), then the payloads padded to the longest: two ncclAllGather calls on the ctx's stream
  std::vector<unsigned long long> lens;
  if (int r = commLengths(cm, n, lens)) return r;
  size_t cap = 8, total = 0;
  for (int r = 0; r < R; ++r) { sizes[r] = (int64_t)lens[(size_t)r]; total += (size_t)lens[(size_t)r]; if ((size_t)lens[(size_t)r] > cap) cap = (size_t)lens[(size_t)r]; }
  cap = (cap + 7) & ~(size_t)7;
  CommBufs bufs;
  unsigned char *dSend = nullptr, *dRecv = nullptr;
  HIPCHK(c, bufs.devAlloc(&dSend, cap));
  if (bufs.devAlloc(&dRecv, cap * (size_t)R) != hipSuccess) return fail(c, T4_ERR_HIP, "out of device memory for the gather");
  if (n > 0) HIPCHK(c, hipMemcpy(dSend, mine, (size_t)n, hipMemcpyHostToDevice));
  if (cm->rccl->allGather(dSend, dRecv, cap, ncclUint8, cm->comm, c->stream) != ncclSuccess) return fail(c, T4_ERR_HIP, "ncclAllGather (payload) failed");
  if (int wr = commWait(cm, "the all-gather")) return wr;   // (before the copies below: see commLengths)
  bufs.host = malloc(total ? total : 1);
  if (!bufs.host) return fail(c, T4_ERR_HIP, "out of host memory for the gather");
  size_t at = 0;
  for (int r = 0; r < R; ++r) {
    if (lens[(size_t)r]) HIPCHK(c, hipMemcpy((unsigned char *)bufs.host + at, dRecv + cap * (size_t)r, (size_t)lens[(size_t)r], hipMemcpyDeviceToHost));
    at += (size_t)lens[(size_t)r];
  }
  *all = bufs.releaseHost();
  return T4_OK;
#else
  return T4_ERR_UNSUPPORTED;
#endif
}

// The same contributions, received by `root` only (the others get *all = NULL): lengths by one small ncclAllGather, payloads by one
// group of ncclSend / ncclRecv -- nothing is padded and nothing lands on ranks that do not write the files.
int t4_comm_gather_bytes(t4_comm *cm, const void *mine, int64_t n, int root, void **all, int64_t *sizes) {
  if (!cm || n < 0 || (n > 0 && !mine) || !all || !sizes || root < 0 || root >= cm->nranks) return T4_ERR_ARG;
  *all = nullptr;
#ifdef __HIPCC__
  t4_ctx *c = cm->ctx;
  if (!cm->comm) return fail(c, T4_ERR_ARG, "t4_comm: the communicator was aborted");
  (void)hipSetDevice(c->device);
  const int R = cm->nranks;
  std::vector<unsigned long long> lens;
  if (int r = commLengths(cm, n, lens)) return r;
  size_t total = 0;
  for (int r = 0; r < R; ++r) { sizes[r] = (int64_t)lens[(size_t)r]; total += (size_t)lens[(size_t)r]; }
  CommBufs bufs;
  unsigned char *dSend = nullptr, *dRecv = nullptr;
  HIPCHK(c, bufs.devAlloc(&dSend, (size_t)n));
  if (n > 0) HIPCHK(c, hipMemcpy(dSend, mine, (size_t)n, hipMemcpyHostToDevice));
  if (cm->rank == root && bufs.devAlloc(&dRecv, total) != hipSuccess) return fail(c, T4_ERR_HIP, "out of device memory for the gather");
  bool ok = cm->rccl->groupStart() == ncclSuccess;
  if (ok && n > 0) ok = cm->rccl->send(dSend, (size_t)n, ncclUint8, root, cm->comm, c->stream) == ncclSuccess;
  if (ok && cm->rank == root) {
    size_t at = 0;
    for (int r = 0; r < R && ok; ++r) { if (lens[(size_t)r]) ok = cm->rccl->recv(dRecv + at, (size_t)lens[(size_t)r], ncclUint8, r, cm->comm, c->stream) == ncclSuccess; at += (size_t)lens[(size_t)r]; }
  }
  ok = (cm->rccl->groupEnd() == ncclSuccess) && ok;
  if (!ok) return fail(c, T4_ERR_HIP, "ncclSend / ncclRecv (gather) failed");
  if (int wr = commWait(cm, "the gather")) return wr;   // (before the copy below: see commLengths)
  if (cm->rank == root) {
    bufs.host = malloc(total ? total : 1);
    if (!bufs.host) return fail(c, T4_ERR_HIP, "out of host memory for the gather");
    if (total) HIPCHK(c, hipMemcpy(bufs.host, dRecv, total, hipMemcpyDeviceToHost));
  }
  *all = bufs.releaseHost();
  return T4_OK;
#else
  return T4_ERR_UNSUPPORTED;
#endif
}

void t4_comm_destroy(t4_comm *cm) {
  if (!cm) return;
#ifdef __HIPCC__
  if (cm->comm) (void)cm->rccl->commDestroy(cm->comm);
#endif
  delete cm;
}

}  // extern "C"

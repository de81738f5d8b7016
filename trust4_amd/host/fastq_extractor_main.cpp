// trust4_amd/host/fastq_extractor_main.cpp -- `fastq-extractor-hip`: the stage-0 candidate filter of the reference's
// `fastq-extractor` (FastqExtractor.cpp) with the candidate test SeqSet::HasHitInSet on the MI355X (t4_has_hit):
//   good = (!IsLowComplexity(read1) && HasHitInSet(read1, 0)) || (mate && !IsLowComplexity(read2) && HasHitInSet(read2, 0))
// (FastqExtractor.cpp:105-134, 516-519), hitLenRequired from the first 1000 reads (436-455), outputs
// <prefix>_1.fq / <prefix>_2.fq or <prefix>.fq in input order (136-143, 470-480), and with --barcode / --UMI the
// <prefix>_bc.fa / <prefix>_umi.fa records of the kept reads (OutputBarcode, 147-203) after --readFormat / range
// extraction, whitelist correction and translation (host/read_format.h).
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/trust4_hip.h"
#include "seq_reader.h"
#include "read_format.h"

namespace {
inline int nucNum(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

bool isLowComplexity(const std::string &s) {   // FastqExtractor.cpp:105-127
  int cnt[5] = {0, 0, 0, 0, 0};
  const int n = (int)s.size();
  // letters other than ACGTN: the reference increments cnt[nucToNum = -1], i.e. nothing this function looks at
  for (char ch : s) { if (ch == 'N') ++cnt[4]; else { int v = nucNum(ch); if (v >= 0) ++cnt[v]; } }
  if (cnt[0] >= n / 2 || cnt[1] >= n / 2 || cnt[2] >= n / 2 || cnt[3] >= n / 2 || cnt[4] >= n / 10) return true;
  int low = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}

struct Rec { std::string id, seq, qual, comment; bool hasQual; };

// One input file set read ahead on its own thread, handed over in chunks of records: parsing (and inflating) the read, mate,
// barcode and UMI files is what this program spends its time on (the candidate test of a million reads takes 0.06 s on the GPU),
// so the files are parsed side by side instead of record by record in turn. Chunks have the same size in every stream, i.e.
// chunk n of each stream holds the same records; a shorter chunk means that file has ended.
struct ChunkStream {
  static constexpr size_t CHUNK = 16384, DEPTH = 4;
  SeqReader *rd = nullptr;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::vector<Rec>> q;
  bool done = false, stop = false;
  void start(SeqReader *r) {
    rd = r;
    th = std::thread([this]() {
      for (;;) {
        std::vector<Rec> v;
        v.reserve(CHUNK);
        while (v.size() < CHUNK && rd->next()) v.push_back(Rec{rd->id, rd->seq, rd->qual, rd->comment, rd->hasQual});
        const bool last = v.size() < CHUNK;
        std::unique_lock<std::mutex> lk(m);
        if (!v.empty()) q.push_back(std::move(v));
        if (last) { done = true; cv.notify_all(); return; }
        cv.notify_all();
        cv.wait(lk, [this]() { return q.size() < DEPTH || stop; });
        if (stop) { done = true; return; }
      }
    });
  }
  bool pop(std::vector<Rec> &out) {   // false: the stream has ended
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this]() { return !q.empty() || done; });
    if (q.empty()) return false;
    out = std::move(q.front());
    q.pop_front();
    cv.notify_all();
    return true;
  }
  void finish() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv.notify_all();
    th.join();
  }
};

void die(t4_ctx *ctx, const char *what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, ctx ? t4_last_error(ctx) : "");
  exit(EXIT_FAILURE);
}
}  // namespace

int main(int argc, char *argv[]) {
  static struct option long_options[] = {{"barcode", required_argument, 0, 10000}, {"barcodeStart", required_argument, 0, 10001},
                                         {"barcodeEnd", required_argument, 0, 10002}, {"barcodeRevComp", no_argument, 0, 10003},
                                         {"barcodeWhitelist", required_argument, 0, 10004}, {"read1Start", required_argument, 0, 10005},
                                         {"read1End", required_argument, 0, 10006}, {"read2Start", required_argument, 0, 10007},
                                         {"read2End", required_argument, 0, 10008}, {"UMI", required_argument, 0, 10009},
                                         {"umiStart", required_argument, 0, 10010}, {"umiEnd", required_argument, 0, 10011},
                                         {"umiRevComp", no_argument, 0, 10012}, {"readFormat", required_argument, 0, 10013},
                                         {"barcodeTranslate", required_argument, 0, 10014}, {"skipBarcodeErrorRead", no_argument, 0, 10015},
                                         {(char *)0, 0, 0, 0}};
  std::string refFa, prefix = "toassemble";
  SeqReader reads, mateReads, barcodeFile, umiFile;
  ReadFormat fmt;
  BarcodeWhitelist whitelist;
  BarcodeTranslate translator;
  bool hasMate = false, hasBarcode = false, hasUmi = false, hasWhitelist = false, skipBarcodeErrorRead = false;
  int barcodeStart = 0, barcodeEnd = -1, read1Start = 0, read1End = -1, read2Start = 0, read2End = -1, umiStart = 0, umiEnd = -1;
  bool barcodeRevComp = false, umiRevComp = false;
  int c, oi = 0;
  while ((c = getopt_long(argc, argv, "f:u:1:2:o:t:", long_options, &oi)) != -1) {
    if (c == 'f') refFa = optarg;
    else if (c == 'u') reads.files.push_back(optarg);
    else if (c == '1') { reads.files.push_back(optarg); hasMate = true; }
    else if (c == '2') { mateReads.files.push_back(optarg); hasMate = true; }
    else if (c == 'o') prefix = optarg;
    else if (c == 't') {   // the candidate test runs on the GPU; what -t changes in the reference's OUTPUT is kept: with more than one
      // thread it reads batches through NextWithBuffer, which leaves a trailing /1 or /2 on the read ids (FastqExtractor.cpp:549-660)
      if (atoi(optarg) > 1) reads.stripMateSuffix = false;
    }
    else if (c == 10000) { hasBarcode = true; barcodeFile.files.push_back(optarg); }
    else if (c == 10001) barcodeStart = atoi(optarg);
    else if (c == 10002) barcodeEnd = atoi(optarg);
    else if (c == 10003) barcodeRevComp = true;
    else if (c == 10004) { hasWhitelist = true; whitelist.load(optarg); }
    else if (c == 10005) read1Start = atoi(optarg);
    else if (c == 10006) read1End = atoi(optarg);
    else if (c == 10007) read2Start = atoi(optarg);
    else if (c == 10008) read2End = atoi(optarg);
    else if (c == 10009) { hasUmi = true; umiFile.files.push_back(optarg); }
    else if (c == 10010) umiStart = atoi(optarg);
    else if (c == 10011) umiEnd = atoi(optarg);
    else if (c == 10012) umiRevComp = true;
    else if (c == 10013) fmt.init(optarg);
    else if (c == 10014) translator.load(optarg);
    else if (c == 10015) skipBarcodeErrorRead = true;
    else { fprintf(stderr, "Unknown parameter\n"); return EXIT_FAILURE; }
  }
  if (refFa.empty() || reads.files.empty()) { fprintf(stderr, "usage: fastq-extractor-hip -f ref.fa (-u reads.fq | -1 r_1.fq -2 r_2.fq) [-o prefix] [--barcode F] [--UMI F] [--readFormat S] [--barcodeWhitelist F] [--barcodeTranslate F] [--skipBarcodeErrorRead]\n"); return EXIT_FAILURE; }
  // range options become segments (FastqExtractor.cpp:422-429; the read-2 range takes the read-1 numbers there, kept)
  if (read1Start != 0 || read1End != -1) fmt.addSegment(read1Start, read1End, 1, FMT_READ1);
  if (read2Start != 0 || read2End != -1) fmt.addSegment(read1Start, read1End, 1, FMT_READ2);
  if (barcodeStart != 0 || barcodeEnd != -1 || barcodeRevComp) fmt.addSegment(barcodeStart, barcodeEnd, barcodeRevComp ? -1 : 1, FMT_BARCODE);
  if (umiStart != 0 || umiEnd != -1 || umiRevComp) fmt.addSegment(umiStart, umiEnd, umiRevComp ? -1 : 1, FMT_UMI);

  // The device, its runtime and the reference set come up on their own thread while the first batch of reads is parsed;
  // gpuReady() (first candidate test) joins it, reports its errors as the serial code did and commits the set.
  t4_ctx *ctx = nullptr;
  t4_index *refSet = nullptr;
  int rc = 0, initRc = 0, hitLenRequired = 27;
  const char *initWhat = nullptr;
  std::thread initThread([&]() {
    if ((initRc = t4_init(getenv("T4_DEVICE") ? atoi(getenv("T4_DEVICE")) : 0, &ctx))) { initWhat = "t4_init"; return; }
    if ((initRc = t4_index_create(ctx, 9, 0, &refSet))) { initWhat = "t4_index_create"; return; }
    if ((initRc = t4_index_load_ref_fasta(refSet, refFa.c_str()))) initWhat = "t4_index_load_ref_fasta";
  });
  auto gpuReady = [&]() {
    if (!initThread.joinable()) return;
    initThread.join();
    if (initRc && !ctx) { fprintf(stderr, "fastq-extractor-hip needs an MI355X (t4_init failed: %d); there is no CPU path.\n", initRc); exit(EXIT_FAILURE); }
    if (initRc) die(ctx, initWhat, initRc);
    if ((rc = t4_index_set_params(refSet, hitLenRequired, 10, 0.9))) die(ctx, "t4_index_set_params", rc);
    if ((rc = t4_index_commit(refSet))) die(ctx, "t4_index_commit", rc);
  };

  // hitLenRequired: 27 with mates, 23 without, at least a fifth of the mean length of the first 1000 reads
  // (FastqExtractor.cpp:436-455)
  if (!hasMate) hitLenRequired = 23;
  int i, len = 0;
  for (i = 0; i < 1000; ++i) { if (!reads.next()) break; len += (int)reads.seq.size(); }
  if (i == 0) { fprintf(stderr, "Read file is empty.\n"); gpuReady(); return EXIT_FAILURE; }
  if (len / (i * 5) > hitLenRequired) hitLenRequired = len / (i * 5);
  if (hitLenRequired > 101) hitLenRequired = 101;
  reads.rewind();
  if (hasBarcode && hasWhitelist) {   // BarcodeCorrector::CollectBackgroundDistribution (BarcodeCorrector.hpp:141-153): first 2 M barcodes
    int n = 0;
    while (barcodeFile.next()) { whitelist.searchAndUpdate(fmt.extract(&barcodeFile.seq, FMT_BARCODE, true), 1); if (++n >= 2000000) break; }
    barcodeFile.rewind();
  }

  FILE *fpBc = hasBarcode ? fopen((prefix + "_bc.fa").c_str(), "w") : nullptr;
  FILE *fpUmi = hasUmi ? fopen((prefix + "_umi.fa").c_str(), "w") : nullptr;
  FILE *fp1 = fopen((prefix + (hasMate ? "_1.fq" : ".fq")).c_str(), "w");
  FILE *fp2 = hasMate ? fopen((prefix + "_2.fq").c_str(), "w") : nullptr;
  if (!fp1 || (hasMate && !fp2)) { fprintf(stderr, "Could not open the output files of %s\n", prefix.c_str()); gpuReady(); return EXIT_FAILURE; }
  auto outputSeq = [&](FILE *fp, const std::string &name, const Rec &r, int cat) {   // OutputSeq (FastqExtractor.cpp:136-143)
    if (r.hasQual) fprintf(fp, "@%s\n%s\n+\n%s\n", name.c_str(), fmt.extract(&r.seq, cat, true).c_str(), fmt.extract(&r.qual, cat, false).c_str());
    else fprintf(fp, ">%s\n%s\n", name.c_str(), fmt.extract(&r.seq, cat, true).c_str());
  };
  // OutputBarcode (FastqExtractor.cpp:147-203); 0 = the read is to be skipped
  auto outputBarcode = [&](FILE *fp, const std::string &name, const Rec &b, int cat, bool correct, bool translate, bool skipError) -> int {
    if (!b.seq.empty()) {
      const bool inComment = fmt.isInComment(cat);
      std::string bc = fmt.extract(inComment ? (b.comment.empty() ? nullptr : &b.comment) : &b.seq, cat, true);
      int result = 0;
      if (correct) result = whitelist.correct(bc, b.hasQual ? &b.qual : nullptr);
      if (result >= 0) {
        if (translate) {
          std::string nb = translator.translate(bc);
          if (nb.empty()) {
            if (skipError) return 0;
            fprintf(stderr, "Barcode %s does not exist in the translation table.\n", bc.c_str());
            exit(-1);
          }
          fprintf(fp, ">%s\n%s\n", name.c_str(), nb.c_str());
        } else fprintf(fp, ">%s\n%s\n", name.c_str(), bc.c_str());
      } else {
        if (skipError) return 0;
        fprintf(fp, ">%s\nmissing_barcode\n", name.c_str());
      }
    } else {
      if (skipError) return 0;
      fprintf(fp, ">%s\nmissing_barcode\n", name.c_str());
    }
    return 1;
  };

  // pairs per candidate-test batch: small enough that the parser threads work on the next batch while this one is tested
  const size_t BATCH = getenv("T4_BATCH") ? (size_t)atol(getenv("T4_BATCH")) : (size_t)1 << 18;
  std::vector<Rec> r1, r2, rb, ru;
  long long total = 0, kept = 0;
  auto test = [&](const std::vector<Rec> &rs, const std::vector<char> *already, std::vector<char> &good) {
    // HasHitInSet for the reads that still need it and pass the low-complexity filter
    std::string bases; std::vector<int64_t> off(1, 0); std::vector<int> which;
    for (size_t k = 0; k < rs.size(); ++k) {
      if (already && (*already)[k]) continue;
      if (isLowComplexity(rs[k].seq)) continue;
      bases += rs[k].seq; off.push_back((int64_t)bases.size()); which.push_back((int)k);
    }
    if (which.empty()) return;
    gpuReady();
    t4_batch *b = nullptr;
    if ((rc = t4_reads_upload_flags(ctx, bases.data(), off.data(), nullptr, (int64_t)which.size(), T4_READS_KMERS_ONLY, &b))) die(ctx, "t4_reads_upload", rc);   // HasHitInSet looks at k-mer codes only: other letters count as T (KmerCode.hpp:99-106)
    std::vector<int32_t> out(which.size());
    if ((rc = t4_has_hit(refSet, b, 0, out.data()))) die(ctx, "t4_has_hit", rc);
    t4_batch_destroy(b);
    for (size_t k = 0; k < which.size(); ++k) if (out[k] != 0) good[which[k]] = 1;
  };
  auto flush = [&]() {
    std::vector<char> good(r1.size(), 0);
    test(r1, nullptr, good);
    if (hasMate) test(r2, &good, good);
    for (size_t k = 0; k < r1.size(); ++k) {
      if (!good[k]) continue;
      // a barcode record that is the read itself (reads carrying their barcode) and of low complexity (FastqExtractor.cpp:521-527)
      if (hasBarcode && (rb[k].seq == r1[k].seq || (hasMate && rb[k].seq == r2[k].seq)) && isLowComplexity(rb[k].seq)) continue;
      if (hasBarcode && !outputBarcode(fpBc, r1[k].id, rb[k], FMT_BARCODE, hasWhitelist, translator.set, skipBarcodeErrorRead)) continue;
      outputSeq(fp1, r1[k].id, r1[k], FMT_READ1);
      if (hasMate) outputSeq(fp2, r1[k].id, r2[k], FMT_READ2);
      if (hasUmi) outputBarcode(fpUmi, r1[k].id, ru[k], FMT_UMI, false, false, false);
      ++kept;
    }
    total += (long long)r1.size();
    r1.clear(); r2.clear(); rb.clear(); ru.clear();
  };
  ChunkStream s1, s2, sb, su;
  s1.start(&reads);
  if (hasMate) s2.start(&mateReads);
  if (hasBarcode) sb.start(&barcodeFile);
  if (hasUmi) su.start(&umiFile);
  auto bail = [&](const char *msg) {
    fprintf(stderr, "%s", msg);
    s1.finish(); s2.finish(); sb.finish(); su.finish();
    if (initThread.joinable()) initThread.join();
    exit(1);
  };
  std::vector<Rec> c1, c2, cb, cu;
  auto append = [](std::vector<Rec> &dst, std::vector<Rec> &src, size_t n) { for (size_t k = 0; k < n; ++k) dst.push_back(std::move(src[k])); src.clear(); };
  double secWait = 0, secFlush = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tw = now();
  while (s1.pop(c1)) {
    const size_t n = c1.size();
    if (hasMate && (!s2.pop(c2) || c2.size() < n)) bail("The two mate-pair read files have different number of reads.\n");
    if (hasBarcode && (!sb.pop(cb) || cb.size() < n)) bail("Read file and barcode file  have different number of reads.\n");
    if (hasUmi && (!su.pop(cu) || cu.size() < n)) bail("Read file and UMI file have different number of reads.\n");
    append(r1, c1, n);
    if (hasMate) append(r2, c2, n);
    if (hasBarcode) append(rb, cb, n);
    if (hasUmi) append(ru, cu, n);
    secWait += now() - tw;
    if (r1.size() >= BATCH) { const double tf = now(); flush(); secFlush += now() - tf; }
    tw = now();
  }
  s1.finish(); s2.finish(); sb.finish(); su.finish();
  { const double tf = now(); flush(); secFlush += now() - tf; }
  if (getenv("T4_TIMING")) fprintf(stderr, "timing: %.2f s waiting for parsed chunks, %.2f s in candidate tests and output\n", secWait, secFlush);
  fclose(fp1);
  if (fp2) fclose(fp2);
  if (fpBc) fclose(fpBc);
  if (fpUmi) fclose(fpUmi);
  fprintf(stderr, "fastq-extractor-hip: %lld of %lld %s kept (hitLenRequired %d)\n", kept, total, hasMate ? "pairs" : "reads", hitLenRequired);
  gpuReady();   // also when no read needed the candidate test: a missing GPU is still an error
  t4_index_destroy(refSet);
  t4_destroy(ctx);
  return 0;
}

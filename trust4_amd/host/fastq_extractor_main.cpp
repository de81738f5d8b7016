// trust4_amd/host/fastq_extractor_main.cpp -- `fastq-extractor-hip`: the stage-0 candidate filter of the reference's
// `fastq-extractor` (FastqExtractor.cpp) with the candidate test SeqSet::HasHitInSet on the MI355X (t4_has_hit):
//   good = (!IsLowComplexity(read1) && HasHitInSet(read1, 0)) || (mate && !IsLowComplexity(read2) && HasHitInSet(read2, 0))
// (FastqExtractor.cpp:105-134, 516-519), hitLenRequired from the first 1000 reads (436-455), outputs
// <prefix>_1.fq / <prefix>_2.fq or <prefix>.fq in input order (136-143, 470-480). Plain reads only in this round: the
// barcode / UMI / readFormat options of the reference are refused, not ignored.
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/trust4_hip.h"
#include "seq_reader.h"

namespace {
inline int nucNum(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

bool isLowComplexity(const std::string &s) {   // FastqExtractor.cpp:105-127
  int cnt[5] = {0, 0, 0, 0, 0};
  const int n = (int)s.size();
  for (char ch : s) { if (ch == 'N') ++cnt[4]; else { int v = nucNum(ch); ++cnt[v < 0 ? 0 : v]; } }
  if (cnt[0] >= n / 2 || cnt[1] >= n / 2 || cnt[2] >= n / 2 || cnt[3] >= n / 2 || cnt[4] >= n / 10) return true;
  int low = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}

struct Rec { std::string id, seq, qual; bool hasQual; };

void die(t4_ctx *ctx, const char *what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, ctx ? t4_last_error(ctx) : "");
  exit(EXIT_FAILURE);
}
}  // namespace

int main(int argc, char *argv[]) {
  static struct option long_options[] = {{"barcode", required_argument, 0, 10000}, {"UMI", required_argument, 0, 10009},
                                         {"readFormat", required_argument, 0, 10013}, {"barcodeWhitelist", required_argument, 0, 10004},
                                         {"barcodeTranslate", required_argument, 0, 10014}, {"skipBarcodeErrorRead", no_argument, 0, 10015},
                                         {(char *)0, 0, 0, 0}};
  std::string refFa, prefix = "toassemble";
  SeqReader reads, mateReads;
  bool hasMate = false;
  int c, oi = 0;
  while ((c = getopt_long(argc, argv, "f:u:1:2:o:t:", long_options, &oi)) != -1) {
    if (c == 'f') refFa = optarg;
    else if (c == 'u') reads.files.push_back(optarg);
    else if (c == '1') { reads.files.push_back(optarg); hasMate = true; }
    else if (c == '2') { mateReads.files.push_back(optarg); hasMate = true; }
    else if (c == 'o') prefix = optarg;
    else if (c == 't') { /* the candidate test runs on the GPU */ }
    else { fprintf(stderr, "fastq-extractor-hip: barcode / UMI / readFormat options are not built yet.\n"); return EXIT_FAILURE; }
  }
  if (refFa.empty() || reads.files.empty()) { fprintf(stderr, "usage: fastq-extractor-hip -f ref.fa (-u reads.fq | -1 r_1.fq -2 r_2.fq) [-o prefix]\n"); return EXIT_FAILURE; }

  t4_ctx *ctx = nullptr;
  int rc = t4_init(getenv("T4_DEVICE") ? atoi(getenv("T4_DEVICE")) : 0, &ctx);
  if (rc) { fprintf(stderr, "fastq-extractor-hip needs an MI355X (t4_init failed: %d); there is no CPU path.\n", rc); return EXIT_FAILURE; }
  t4_index *refSet = nullptr;
  if ((rc = t4_index_create(ctx, 9, 0, &refSet))) die(ctx, "t4_index_create", rc);
  if ((rc = t4_index_load_ref_fasta(refSet, refFa.c_str()))) die(ctx, "t4_index_load_ref_fasta", rc);

  // hitLenRequired from the first 1000 reads (FastqExtractor.cpp:436-455)
  int hitLenRequired = 27, i, len = 0;
  for (i = 0; i < 1000; ++i) { if (!reads.next()) break; len += (int)reads.seq.size(); }
  if (i == 0) { fprintf(stderr, "Read file is empty.\n"); return EXIT_FAILURE; }
  if (len / (i * 5) > hitLenRequired) hitLenRequired = len / (i * 5);
  if (hitLenRequired > 101) hitLenRequired = 101;
  if ((rc = t4_index_set_params(refSet, hitLenRequired, 10, 0.9))) die(ctx, "t4_index_set_params", rc);
  if ((rc = t4_index_commit(refSet))) die(ctx, "t4_index_commit", rc);
  reads.rewind();

  FILE *fp1 = fopen((prefix + (hasMate ? "_1.fq" : ".fq")).c_str(), "w");
  FILE *fp2 = hasMate ? fopen((prefix + "_2.fq").c_str(), "w") : nullptr;
  if (!fp1 || (hasMate && !fp2)) { fprintf(stderr, "Could not open the output files of %s\n", prefix.c_str()); return EXIT_FAILURE; }
  auto outputSeq = [](FILE *fp, const Rec &r) {   // OutputSeq (FastqExtractor.cpp:136-143)
    if (r.hasQual) fprintf(fp, "@%s\n%s\n+\n%s\n", r.id.c_str(), r.seq.c_str(), r.qual.c_str());
    else fprintf(fp, ">%s\n%s\n", r.id.c_str(), r.seq.c_str());
  };

  const size_t BATCH = getenv("T4_BATCH") ? (size_t)atol(getenv("T4_BATCH")) : (size_t)1 << 20;
  std::vector<Rec> r1, r2;
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
    t4_batch *b = nullptr;
    if ((rc = t4_reads_upload(ctx, bases.data(), off.data(), nullptr, (int64_t)which.size(), &b))) die(ctx, "t4_reads_upload", rc);
    std::vector<int32_t> out(which.size());
    if ((rc = t4_has_hit(refSet, b, 0, out.data()))) die(ctx, "t4_has_hit", rc);
    t4_batch_destroy(b);
    for (size_t k = 0; k < which.size(); ++k) if (out[k] != 0) good[which[k]] = 1;
  };
  auto flush = [&]() {
    std::vector<char> good(r1.size(), 0);
    test(r1, nullptr, good);
    if (hasMate) test(r2, &good, good);
    for (size_t k = 0; k < r1.size(); ++k)
      if (good[k]) { outputSeq(fp1, r1[k]); if (hasMate) outputSeq(fp2, r2[k]); ++kept; }
    total += (long long)r1.size();
    r1.clear(); r2.clear();
  };
  while (reads.next()) {
    if (hasMate && !mateReads.next()) { fprintf(stderr, "The two mate-pair read files have different number of reads.\n"); exit(1); }
    r1.push_back(Rec{reads.id, reads.seq, reads.qual, reads.hasQual});
    if (hasMate) r2.push_back(Rec{mateReads.id, mateReads.seq, mateReads.qual, mateReads.hasQual});
    if (r1.size() >= BATCH) flush();
  }
  flush();
  fclose(fp1);
  if (fp2) fclose(fp2);
  fprintf(stderr, "fastq-extractor-hip: %lld of %lld %s kept (hitLenRequired %d)\n", kept, total, hasMate ? "pairs" : "reads", hitLenRequired);
  t4_index_destroy(refSet);
  t4_destroy(ctx);
  return 0;
}

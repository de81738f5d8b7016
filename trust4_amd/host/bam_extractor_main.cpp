// trust4_amd/host/bam_extractor_main.cpp -- `bam-extractor-hip`: the stage-0 candidate extraction of the reference's
// `bam-extractor` (BamExtractor.cpp) with the candidate test SeqSet::HasHitInSet on the MI355X (t4_has_hit).
//
// Same command line (-b -f -o -t -u --barcode --UMI --mateIdSuffixLen), same files: PREFIX_1.fq / _2.fq (or PREFIX.fq for
// single-end data), PREFIX_bc.fa, PREFIX_umi.fa. What is kept is what the reference keeps: reads aligned inside a V/J/C gene of the
// -f file, and reads that are unaligned (or aligned to an alternative contig) and pass the k-mer candidate test; for paired
// data a second scan fetches both mates of every kept name. The record order of the outputs is the reference's with one thread
// (-t is accepted; with more threads the reference hands unaligned reads to a work queue and writes them as they finish).
// BAM is read by trust4_amd/host/bam_reader.h (the format itself; the reference goes through its vendored samtools).
// The reference tests read after read; here the reads that need the test are collected in file order, tested in batches of up to
// T4_BATCH reads in one kernel launch each, and the decisions are then replayed in file order (they only depend on earlier records).
#include <getopt.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/trust4_hip.h"
#include "bam_reader.h"

namespace {
int nucNum(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
void PrintLog(const char *fmt, ...) {
  char msg[2048];
  va_list args;
  va_start(args, fmt);
  vsnprintf(msg, sizeof msg, fmt, args);
  va_end(args);
  time_t t = time(NULL);
  char stime[200];
  strftime(stime, sizeof stime, "%c", localtime(&t));
  fprintf(stderr, "[%s] %s\n", stime, msg);
}
bool isLowComplexity(const std::string &s) {   // BamExtractor.cpp:140-162
  int cnt[5] = {0, 0, 0, 0, 0};
  const int n = (int)s.size();
  for (char ch : s) { if (ch == 'N') ++cnt[4]; else { int v = nucNum(ch); ++cnt[v < 0 ? 0 : v]; } }
  if (cnt[0] >= n / 2 || cnt[1] >= n / 2 || cnt[2] >= n / 2 || cnt[3] >= n / 2 || cnt[4] >= n / 10) return true;
  int low = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}
bool validAlternativeChrom(const std::string &c) { return c.find('_') != std::string::npos || c.find('.') != std::string::npos; }   // 114-125
void trimName(std::string &name, int trimLen) {   // 164-179
  const int len = (int)name.size();
  if (trimLen == -1) { if (len >= 2 && (name[len - 1] == '1' || name[len - 1] == '2') && name[len - 2] == '/') name.erase(len - 2, 2); }
  else name.erase(len - trimLen, trimLen);
}
void outputSeq(FILE *fp, const std::string &name, const std::string &seq, const std::string &qual) { fprintf(fp, "@%s\n%s\n+\n%s\n", name.c_str(), seq.c_str(), qual.c_str()); }
void outputBarcode(FILE *fp, const std::string &name, bool has, const std::string &v) { fprintf(fp, ">%s\n%s\n", name.c_str(), has ? v.c_str() : "missing_barcode"); }
struct Interval { int chrId, start, end; bool operator<(const Interval &b) const { return chrId != b.chrId ? chrId < b.chrId : start != b.start ? start < b.start : end < b.end; } };
void die(t4_ctx *ctx, const char *what, int rc) { fprintf(stderr, "%s failed (%d): %s\n", what, rc, ctx ? t4_last_error(ctx) : ""); exit(EXIT_FAILURE); }

// one decision of the first scan, kept until its candidate tests are back
struct Event {
  int kind;                 // 0 unaligned pair, 1 paired read to test -> candidate name, 2 single-end read to test -> output,
                            // 3 gene read of paired data -> candidate name, 4 gene read of single-end data -> output
  std::string name, s1, q1, s2, q2, bc, umi;
  bool hasBc = false, hasUmi = false, aligned = false, secondIsFirstMate = false;
  int t1 = -1, t2 = -1;     // positions in the test batch (-1: not tested, i.e. the answer is "no hit")
};
}  // namespace

int main(int argc, char *argv[]) {
  static const char usage[] = "./bam-extractor-hip [OPTIONS]:\n\t-b STRING: path to BAM file\n\t-f STRING: path to the reference gene sequence file\n"
                              "\t-o STRING: prefix to the output file\n\t-t INT: accepted (the candidate test runs on the GPU)\n\t-u: filter alignment records without setting the unaligned flag properly\n"
                              "\t--barcode STRING: the barcode field in the bam file\n\t--UMI STRING: the UMI field in the bam file\n\t--mateIdSuffixLen INT: the suffix length in read id for mate\n";
  if (argc <= 1) { fprintf(stderr, "%s", usage); return 0; }
  static struct option long_options[] = {{"barcode", required_argument, 0, 10000}, {"UMI", required_argument, 0, 10001}, {"mateIdSuffixLen", required_argument, 0, 10002}, {0, 0, 0, 0}};
  std::string refFa, bamPath, prefix = "toassemble", bcField, umiField;
  bool abnormalUnaligned = false;
  int mateIdLen = -1, c, oi = 0;
  while ((c = getopt_long(argc, argv, "f:b:o:t:u", long_options, &oi)) != -1) {
    if (c == 'f') refFa = optarg;
    else if (c == 'b') bamPath = optarg;
    else if (c == 'o') prefix = optarg;
    else if (c == 'u') abnormalUnaligned = true;
    else if (c == 't') { }
    else if (c == 10000) bcField = optarg;
    else if (c == 10001) umiField = optarg;
    else if (c == 10002) mateIdLen = atoi(optarg);
    else { fprintf(stderr, "Unknown parameter\n"); return EXIT_FAILURE; }
  }
  if (refFa.empty()) { fprintf(stderr, "Need to use -f to specify the receptor genome sequence.\n"); return EXIT_FAILURE; }
  if (bamPath.empty()) { fprintf(stderr, "Need to use -b to specify the BAM file.\n"); return EXIT_FAILURE; }
  BamReader bam;
  if (!bam.open(bamPath.c_str())) { fprintf(stderr, "Can not open %s.\n", bamPath.c_str()); return 1; }

  // the device comes up beside the header work and the sampling scan
  t4_ctx *ctx = nullptr;
  t4_index *refSet = nullptr;
  int rc = 0, initRc = 0;
  const char *initWhat = nullptr;
  std::thread initThread([&]() {
    if ((initRc = t4_init(getenv("T4_DEVICE") ? atoi(getenv("T4_DEVICE")) : 0, &ctx))) { initWhat = "t4_init"; return; }
    if ((initRc = t4_index_create(ctx, 9, 0, &refSet))) { initWhat = "t4_index_create"; return; }
    if ((initRc = t4_index_load_ref_fasta(refSet, refFa.c_str()))) initWhat = "t4_index_load_ref_fasta";
  });

  // gene intervals from the headers of the -f file (BamExtractor.cpp:543-565): ">NAME CHROM START END STRAND" + the sequence token
  std::vector<Interval> genes;
  {
    FILE *fpRef = fopen(refFa.c_str(), "r");
    if (!fpRef) { fprintf(stderr, "Need to use -f to specify the receptor genome sequence.\n"); initThread.join(); return EXIT_FAILURE; }
    static char geneName[100001], chrom[100001], strand[100001], seqTok[1000001];
    int start, end;
    PrintLog("Start to extract candidate reads from bam file.");
    while (fscanf(fpRef, "%100000s %100000s %d %d %100000s", geneName, chrom, &start, &end, strand) == 5) {
      genes.push_back(Interval{bam.chromId(chrom), start, end});
      if (fscanf(fpRef, "%1000000s", seqTok) != 1) break;
    }
    fclose(fpRef);
  }
  const int geneCnt = (int)genes.size();
  std::sort(genes.begin(), genes.end());

  // Alignments::GetGeneralInfo(true) (alignments.hpp:559-648): read length and pairedness from the first 100 000 primary records
  int readLen = 0;
  bool paired = false;
  {
    BamRecord r;
    long long total = 0, hasMate = 0;
    while (bam.next(r)) {
      if (!r.isPrimary()) continue;
      if (r.lseq > readLen) readLen = r.lseq;
      if (r.flag & 0x1) ++hasMate;
      if (++total >= 100000) break;
    }
    paired = hasMate > total / 2;
    bam.rewind();
  }
  int hitLenRequired = paired ? 21 : 17;
  if (readLen / 5 > hitLenRequired) hitLenRequired = readLen / 5;
  if (hitLenRequired > 101) hitLenRequired = 101;
  initThread.join();
  if (initRc && !ctx) { fprintf(stderr, "bam-extractor-hip needs an MI355X (t4_init failed: %d); there is no CPU path.\n", initRc); return EXIT_FAILURE; }
  if (initRc) die(ctx, initWhat, initRc);
  if ((rc = t4_index_set_params(refSet, hitLenRequired, 10, 0.9))) die(ctx, "t4_index_set_params", rc);
  if ((rc = t4_index_commit(refSet))) die(ctx, "t4_index_commit", rc);

  FILE *fp1 = fopen((prefix + (paired ? "_1.fq" : ".fq")).c_str(), "w");
  FILE *fp2 = paired ? fopen((prefix + "_2.fq").c_str(), "w") : nullptr;
  FILE *fpBc = bcField.empty() ? nullptr : fopen((prefix + "_bc.fa").c_str(), "w");
  FILE *fpUmi = umiField.empty() ? nullptr : fopen((prefix + "_umi.fa").c_str(), "w");
  if (!fp1 || (paired && !fp2)) { fprintf(stderr, "Could not open the output files of %s\n", prefix.c_str()); return EXIT_FAILURE; }

  std::map<std::string, int> candidates;   // name -> bit 0: mate 1 stored, bit 1: mate 2 stored (second scan)
  std::map<std::string, int> usedName;
  std::vector<Event> events;
  std::string testBases;
  std::vector<int64_t> testOff(1, 0);
  const size_t BATCH = getenv("T4_BATCH") ? (size_t)atol(getenv("T4_BATCH")) : (size_t)1 << 20;
  auto addTest = [&](const std::string &s) -> int {
    if (isLowComplexity(s)) return -1;
    testBases += s; testOff.push_back((int64_t)testBases.size());
    return (int)testOff.size() - 2;
  };
  auto tags = [&](const BamRecord &r, Event &e) {
    if (fpBc) { const char *v = r.fieldZ(bcField.c_str()); e.hasBc = v != nullptr; if (v) e.bc = v; }
    if (fpUmi) { const char *v = r.fieldZ(umiField.c_str()); e.hasUmi = v != nullptr; if (v) e.umi = v; }
  };
  auto replay = [&]() {   // candidate tests of the pending events in one launch, then their decisions in file order
    std::vector<int32_t> hit(testOff.size() - 1, 0);
    if (!hit.empty()) {
      t4_batch *b = nullptr;
      if ((rc = t4_reads_upload_flags(ctx, testBases.data(), testOff.data(), nullptr, (int64_t)hit.size(), T4_READS_KMERS_ONLY, &b))) die(ctx, "t4_reads_upload", rc);
      if ((rc = t4_has_hit(refSet, b, 0, hit.data()))) die(ctx, "t4_has_hit", rc);
      t4_batch_destroy(b);
    }
    auto good = [&](int t) { return t >= 0 && hit[(size_t)t] != 0; };
    for (const Event &e : events) {
      if (e.kind == 0) {
        // (!low(s2) && !low(s1)) && (hit(s2) || hit(s1)): a low-complexity mate was not tested (t == -1) and vetoes the pair
        if (e.t1 >= 0 && e.t2 >= 0 && (good(e.t2) || good(e.t1))) {
          if (!e.secondIsFirstMate) { outputSeq(fp1, e.name, e.s1, e.q1); outputSeq(fp2, e.name, e.s2, e.q2); }
          else { outputSeq(fp1, e.name, e.s2, e.q2); outputSeq(fp2, e.name, e.s1, e.q1); }
          if (fpBc) outputBarcode(fpBc, e.name, e.hasBc, e.bc);
          if (fpUmi) outputBarcode(fpUmi, e.name, e.hasUmi, e.umi);
        }
      } else if (e.kind == 1) { if (good(e.t1)) candidates.insert({e.name, 0}); }
      else if (e.kind == 2) {
        if (e.aligned && usedName.count(e.name)) continue;
        if (good(e.t1)) {
          if (e.aligned) usedName[e.name] = 1;
          outputSeq(fp1, e.name, e.s1, e.q1);
          if (fpBc) outputBarcode(fpBc, e.name, e.hasBc, e.bc);
          if (fpUmi) outputBarcode(fpUmi, e.name, e.hasUmi, e.umi);
        }
      } else if (e.kind == 3) candidates.insert({e.name, 0});
      else {
        if (usedName.count(e.name)) continue;
        usedName[e.name] = 1;
        outputSeq(fp1, e.name, e.s1, e.q1);
        if (fpBc) outputBarcode(fpBc, e.name, e.hasBc, e.bc);
        if (fpUmi) outputBarcode(fpUmi, e.name, e.hasUmi, e.umi);
      }
    }
    events.clear(); testBases.clear(); testOff.assign(1, 0);
  };

  // ---- first scan (BamExtractor.cpp:621-838)
  int tag = 0;
  BamRecord r, r2;
  while (bam.next(r)) {
    const bool alignedAlt = r.isAligned() && validAlternativeChrom(bam.refNames[(size_t)r.tid]);
    if (!r.isTemplateAligned() || alignedAlt) {
      if (!r.isTemplateAligned() && paired && !abnormalUnaligned) {   // the two reads of an unaligned template come together
        Event e; e.kind = 0;
        e.s1 = r.readSeq(); e.q1 = r.readQual();
        std::string name = r.name;
        if (!bam.next(r2)) { fprintf(stderr, "Two reads from the unaligned fragment are not showing up together. Please use -u(--abnormalUnmapFlag from wrapper) option.\n"); return EXIT_FAILURE; }
        std::string mateName = r2.name;
        e.s2 = r2.readSeq(); e.q2 = r2.readQual();
        trimName(name, mateIdLen); trimName(mateName, mateIdLen);
        if (name != mateName) {
          fprintf(stderr, "%s\n%s\n", name.c_str(), mateName.c_str());
          fprintf(stderr, "Two reads from the unaligned fragment are not showing up together. Please use -u(--abnormalUnmapFlag from wrapper) option.\n");
          return EXIT_FAILURE;
        }
        e.name = name; e.secondIsFirstMate = r2.isFirstMate();
        tags(r2, e);   // the reference reads the fields of the record it stands on: the second one
        e.t2 = addTest(e.s2); e.t1 = addTest(e.s1);
        events.push_back(std::move(e));
      } else if (paired) {
        Event e; e.kind = 1;
        e.t1 = addTest(r.readSeq());
        if (e.t1 >= 0) { e.name = r.name; trimName(e.name, mateIdLen); events.push_back(std::move(e)); }
      } else {
        Event e; e.kind = 2;
        e.aligned = r.isAligned();
        e.name = r.name;
        e.s1 = r.readSeq(); e.q1 = r.readQual();
        e.t1 = addTest(e.s1);
        if (e.t1 >= 0) { tags(r, e); events.push_back(std::move(e)); }
      }
      if (testOff.size() - 1 >= BATCH) replay();
      continue;
    }
    if (!r.isAligned()) continue;   // paired data, the other mate is aligned
    int64_t start, end;
    r.span(start, end);
    while (tag < geneCnt && (r.tid > genes[(size_t)tag].chrId || (r.tid == genes[(size_t)tag].chrId && start > genes[(size_t)tag].end))) ++tag;
    if (tag >= geneCnt) continue;
    if (r.tid < genes[(size_t)tag].chrId || (r.tid == genes[(size_t)tag].chrId && end <= genes[(size_t)tag].start)) continue;
    const std::string seq = r.readSeq();
    if (isLowComplexity(seq)) continue;
    Event e;
    e.name = r.name;
    if (paired) { e.kind = 3; trimName(e.name, mateIdLen); }
    else { e.kind = 4; e.s1 = seq; e.q1 = r.readQual(); tags(r, e); }
    events.push_back(std::move(e));
  }
  replay();
  bam.rewind();
  if (!paired) {
    fclose(fp1); if (fpBc) fclose(fpBc); if (fpUmi) fclose(fpUmi);
    bam.close();
    PrintLog("Finish extracting reads.");
    t4_index_destroy(refSet); t4_destroy(ctx);
    return 0;
  }
  PrintLog("Finish obtaining the candidate read ids.");

  // ---- second scan: both mates of every candidate name, written when the second one shows up (BamExtractor.cpp:870-941)
  struct Mates { std::string s1, q1, s2, q2; bool has1 = false, has2 = false; };
  std::map<std::string, Mates> store;
  const size_t candidateCnt = candidates.size();
  size_t outputCnt = 0;
  while (candidateCnt > 0 && bam.next(r)) {
    if (!r.isPrimary()) continue;
    if (!r.isTemplateAligned() && !abnormalUnaligned) continue;
    std::string name = r.name;
    trimName(name, mateIdLen);
    if (!candidates.count(name)) continue;
    Mates &m = store[name];
    if (r.isFirstMate()) { m.s1 = r.readSeq(); m.q1 = r.readQual(); m.has1 = true; }
    else { m.s2 = r.readSeq(); m.q2 = r.readQual(); m.has2 = true; }
    if (m.has1 && m.has2) {
      outputSeq(fp1, name, m.s1, m.q1);
      outputSeq(fp2, name, m.s2, m.q2);
      if (fpBc) { const char *v = r.fieldZ(bcField.c_str()); outputBarcode(fpBc, name, v != nullptr, v ? v : ""); }
      if (fpUmi) { const char *v = r.fieldZ(umiField.c_str()); outputBarcode(fpUmi, name, v != nullptr, v ? v : ""); }
      m = Mates();
      if (++outputCnt == candidateCnt) break;
    }
  }
  fclose(fp1); if (fp2) fclose(fp2); if (fpBc) fclose(fpBc); if (fpUmi) fclose(fpUmi);
  bam.close();
  PrintLog("Finish extracting reads.");
  t4_index_destroy(refSet); t4_destroy(ctx);
  return 0;
}

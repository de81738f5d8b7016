// trust4_amd/host/bam_reader.h -- minimal BAM reader for bam-extractor-hip (the reference reads BAM through its vendored
// samtools 0.1.19; this is the format itself, SAM/BAM specification section 4: a BGZF file is a series of gzip members, which
// zlib's gz* functions read as one stream).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <map>
#include <string>
#include <vector>

struct BamRecord {
  int32_t tid = -1, pos = -1, mtid = -1, mpos = -1, lseq = 0;
  uint16_t flag = 0, ncigar = 0;
  std::string name;
  std::vector<uint32_t> cigar;
  std::vector<uint8_t> seq, qual, aux;
  // Alignments accessors (alignments.hpp:357-427)
  bool isFirstMate() const { return flag & 0x40; }
  bool isReverse() const { return flag & 0x10; }
  bool isTemplateAligned() const { return !((flag & 0xd) == 0xd || (flag & 0x5) == 0x4 || tid < 0); }
  bool isAligned() const { return !((flag & 0x4) || tid < 0); }
  bool isPrimary() const { return (flag & 0x900) == 0; }
  // GetReadSeq / GetQual (alignments.hpp:489-543): the read as sequenced (reverse-strand alignments are turned back)
  std::string readSeq() const {
    std::string s((size_t)lseq, 'N');
    for (int i = 0; i < lseq; ++i) {
      const int j = isReverse() ? lseq - 1 - i : i;
      const int bit = (seq[(size_t)j >> 1] >> ((~j & 1) << 2)) & 0xf;
      char c = 'N';
      if (bit == 1) c = isReverse() ? 'T' : 'A'; else if (bit == 2) c = isReverse() ? 'G' : 'C';
      else if (bit == 4) c = isReverse() ? 'C' : 'G'; else if (bit == 8) c = isReverse() ? 'A' : 'T';
      s[(size_t)i] = c;
    }
    return s;
  }
  std::string readQual() const {
    std::string q((size_t)lseq, '!');
    for (int i = 0; i < lseq; ++i) q[(size_t)i] = (char)(qual[(size_t)(isReverse() ? lseq - 1 - i : i)] + 33);
    return q;
  }
  // first / last reference position covered, as Alignments::Next builds its segments (alignments.hpp:185-246)
  void span(int64_t &first, int64_t &last) const {
    int64_t start = pos, len = 0;
    bool any = false;
    first = pos; last = pos - 1;
    for (uint32_t v : cigar) {
      const int op = (int)(v & 0xf); int64_t num = v >> 4;
      if (op == 0 || op == 2) len += num;                         // M, D
      else if (op == 4 || op == 5 || op == 6 || op == 1) { }      // S, H, P, I
      else if (op == 3) { if (!any) { first = start; any = true; } last = start + len - 1; start = start + len + num; len = 0; }   // N: a segment ends
      else len += num;                                            // =, X
    }
    if (len > 0) { if (!any) { first = start; any = true; } last = start + len - 1; }
  }
  // GetFieldZ (alignments.hpp:453-460): value of a Z tag, null when absent (or not a string)
  const char *fieldZ(const char *tag) const {
    size_t i = 0;
    const size_t n = aux.size();
    while (i + 3 <= n) {
      const char t0 = (char)aux[i], t1 = (char)aux[i + 1], ty = (char)aux[i + 2];
      i += 3;
      const bool hit = t0 == tag[0] && t1 == tag[1];
      if (ty == 'Z' || ty == 'H') { const char *v = (const char *)&aux[i]; while (i < n && aux[i]) ++i; ++i; if (hit) return v; }   // bam_aux2Z returns the bytes of Z and H fields
      else if (ty == 'A' || ty == 'c' || ty == 'C') i += 1;
      else if (ty == 's' || ty == 'S') i += 2;
      else if (ty == 'i' || ty == 'I' || ty == 'f') i += 4;
      else if (ty == 'd') i += 8;
      else if (ty == 'B') {
        if (i + 5 > n) break;
        const char sub = (char)aux[i]; uint32_t cnt; memcpy(&cnt, &aux[i + 1], 4);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        i += 5 + (size_t)cnt * w;
      } else break;
      if (hit && ty != 'Z' && ty != 'H') return nullptr;
    }
    return nullptr;
  }
};

struct BamReader {
  gzFile fp = nullptr;
  std::string path;
  std::vector<std::string> refNames;
  std::map<std::string, int> nameToId;
  bool readExact(void *dst, size_t n) { return n == 0 || gzread(fp, dst, (unsigned)n) == (int)n; }
  bool open(const char *file) {
    path = file;
    fp = gzopen(file, "rb");
    if (!fp) return false;
    gzbuffer(fp, 1 << 20);
    return header();
  }
  bool header() {
    char magic[4];
    int32_t ltext = 0, nref = 0;
    if (!readExact(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !readExact(&ltext, 4)) return false;
    std::vector<char> text((size_t)ltext);
    if (!readExact(text.data(), (size_t)ltext) || !readExact(&nref, 4)) return false;
    refNames.clear(); nameToId.clear();
    for (int i = 0; i < nref; ++i) {
      int32_t lname = 0, lref = 0;
      if (!readExact(&lname, 4)) return false;
      std::vector<char> nm((size_t)lname);
      if (!readExact(nm.data(), (size_t)lname) || !readExact(&lref, 4)) return false;
      refNames.push_back(std::string(nm.data()));
      nameToId[refNames.back()] = i;
    }
    return true;
  }
  void rewind() { gzrewind(fp); header(); }
  void close() { if (fp) { gzclose(fp); fp = nullptr; } }
  // false at the end of the file; a record that is cut short or whose lengths do not add up ends the program (a truncated or
  // corrupt BAM must not pass for a shorter one)
  [[noreturn]] void corrupt(const char *what) const {
    fprintf(stderr, "%s: %s -- truncated or corrupt BAM\n", path.c_str(), what);
    exit(1);
  }
  bool next(BamRecord &r) {
    int32_t block = 0;
    const int got = gzread(fp, &block, 4);
    if (got == 0) return false;                        // clean end of the file, at a record boundary
    if (got != 4) corrupt("record header cut short");
    if (block < 32 || block > (1 << 28)) corrupt("implausible record size");
    std::vector<uint8_t> buf((size_t)block);
    if (!readExact(buf.data(), (size_t)block)) corrupt("record cut short");
    const uint8_t *p = buf.data();
    uint8_t lname; uint16_t ncig, flag; int32_t lseq;
    memcpy(&r.tid, p, 4); memcpy(&r.pos, p + 4, 4); lname = p[8]; memcpy(&ncig, p + 12, 2); memcpy(&flag, p + 14, 2);
    memcpy(&lseq, p + 16, 4); memcpy(&r.mtid, p + 20, 4); memcpy(&r.mpos, p + 24, 4);
    r.flag = flag; r.ncigar = ncig; r.lseq = lseq;
    if (lseq < 0 || 32 + (size_t)lname + 4 * (size_t)ncig + ((size_t)lseq + 1) / 2 + (size_t)lseq > (size_t)block) corrupt("field lengths exceed the record");
    size_t o = 32;
    r.name.assign((const char *)p + o, lname ? (size_t)lname - 1 : 0); o += lname;
    r.cigar.resize(ncig);
    if (ncig) memcpy(r.cigar.data(), p + o, 4 * (size_t)ncig);
    o += 4 * (size_t)ncig;
    r.seq.assign(p + o, p + o + ((size_t)lseq + 1) / 2); o += ((size_t)lseq + 1) / 2;
    r.qual.assign(p + o, p + o + (size_t)lseq); o += (size_t)lseq;
    r.aux.assign(p + o, p + (size_t)block);
    return true;
  }
  // Alignments::GetChromIdFromName (alignments.hpp:289-310)
  int chromId(const char *s) const {
    std::string ss(s);
    auto it = nameToId.find(ss);
    if (it != nameToId.end()) return it->second;
    if (strlen(s) >= 4 && (it = nameToId.find(std::string(s + 3))) != nameToId.end()) return it->second;
    if ((it = nameToId.find(std::string("chr") + ss)) != nameToId.end()) return it->second;
    printf("Unknown genome name: %s\n", s);
    exit(1);
  }
};

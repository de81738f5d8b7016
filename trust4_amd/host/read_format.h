// trust4_amd/host/read_format.h -- host-side text handling of the stage-0 extractor: which parts of a read / barcode / UMI
// record are kept (ReadFormatter.hpp), barcode correction against a whitelist (BarcodeCorrector.hpp) and barcode
// translation (BarcodeTranslator.hpp). Plain string work around the GPU candidate test; written against the
// behaviour of those reference classes, including the corners noted below.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <unordered_map>
#include <vector>

enum { FMT_READ1 = 0, FMT_READ2, FMT_BARCODE, FMT_UMI, FMT_COUNT };

struct ReadFormat {
  struct Seg { int start = 0, end = -1, strand = 1; bool inComment = false; int field = 0; std::string prefix; };
  std::vector<Seg> segs[FMT_COUNT];

  // one "xx:...": r1 / r2 / bc / um, optional "hd:<field number or tag prefix>:", then start:end[:strand]
  // (ReadFormatter.hpp:49-143)
  bool parseOne(const char *s, int len) {
    if (len < 3 || s[2] != ':') return false;
    int cat;
    if (s[0] == 'r' && s[1] == '1') cat = FMT_READ1;
    else if (s[0] == 'r' && s[1] == '2') cat = FMT_READ2;
    else if (s[0] == 'b' && s[1] == 'c') cat = FMT_BARCODE;
    else if (s[0] == 'u' && s[1] == 'm') cat = FMT_UMI;
    else return false;
    Seg seg;
    int at = 3;
    if (len >= 6 && s[3] == 'h' && s[4] == 'd' && s[5] == ':') {
      seg.inComment = true;
      int i = 6;
      std::string tok;
      while (i < len && s[i] != ':') tok.push_back(s[i++]);
      bool digits = true;
      for (char ch : tok) if (ch < '0' || ch > '9') digits = false;
      if (digits) seg.field = atoi(tok.c_str());
      else { seg.field = -1; seg.prefix = tok; }
      at = i + 1;
    }
    int part = 0;
    std::string tok;
    for (int i = at; i <= len; ++i) {
      if (i >= len || s[i] == ':') {
        if (part == 0) seg.start = atoi(tok.c_str());
        else if (part == 1) seg.end = atoi(tok.c_str());
        else seg.strand = (!tok.empty() && tok[0] == '+') ? 1 : -1;
        tok.clear();
        if (i < len) ++part;
      } else tok.push_back(s[i]);
    }
    if (part >= 3 || part < 1) return false;
    segs[cat].push_back(seg);
    return true;
  }
  void init(const char *fmt) {   // entries separated by ',' or ';' (ReadFormatter.hpp:212-239)
    for (int i = 0; fmt[i];) {
      int j = i;
      while (fmt[j] && fmt[j] != ';' && fmt[j] != ',') ++j;
      if (!parseOne(fmt + i, j - i)) { fprintf(stderr, "Format description error in %s\n", fmt); exit(1); }
      i = fmt[j] ? j + 1 : j;
    }
  }
  void addSegment(int start, int end, int strand, int cat) { Seg s; s.start = start; s.end = end; s.strand = strand; segs[cat].push_back(s); }
  bool isInComment(int cat) const { return !segs[cat].empty() && segs[cat][0].inComment; }
  bool needExtract(int cat) const {
    if (segs[cat].empty()) return false;
    if (segs[cat].size() == 1) { const Seg &s = segs[cat][0]; if (s.start == 0 && s.end == -1 && s.strand == 1 && !s.inComment) return false; }
    return true;
  }
  // ReadFormatter::Extract (ReadFormatter.hpp:296-403). `text` is the sequence, its quality string, or (categories kept
  // in the header) the comment; a missing comment yields "". Segments are concatenated in the order given; one minus
  // segment reverses the whole result (complemented for bases, non-ACGT -> N).
  std::string extract(const std::string *text, int cat, bool complement) const {
    if (!text) return std::string();
    if (!needExtract(cat)) return *text;
    const std::string &seq = *text;
    const int len = (int)seq.size();
    std::string out;
    int strand = 1;
    for (const Seg &sg : segs[cat]) {
      int start = sg.start, end = sg.end, lenk = len;
      if (isInComment(cat)) {
        int fstart = 0, fend = 0;
        if (sg.field >= 0) {
          int f = 0;
          for (int j = 0; j <= len; ++j) {
            const char ch = j < len ? seq[j] : '\0';
            if (ch == ' ' || ch == '\t' || ch == '\0') {
              ++f;
              if (f == sg.field) fstart = j + 1;
              else if (f == sg.field + 1) { fend = j - 1; break; }
            }
          }
          if (f <= sg.field) { fstart = len; fend = len - 1; }
        } else {
          size_t p = seq.find(sg.prefix);
          if (p != std::string::npos) {
            fstart = (int)p;
            while (p < seq.size() && seq[p] != ' ' && seq[p] != '\t') ++p;
            fend = (int)p - 1;
          } else { fstart = len; fend = len - 1; }
        }
        if (start >= 0) start += fstart;
        if (end >= 0) end += fstart;
        lenk = fend + 1;
      }
      if (start < 0) start = lenk + start;
      if (end >= lenk) end = lenk - 1;
      else if (end < 0) end = lenk + end;
      for (int j = start; j <= end; ++j) if (j >= 0 && j < len) out.push_back(seq[j]);
      if (sg.strand == -1) strand = -1;
    }
    if (strand == -1) {
      std::string r(out.rbegin(), out.rend());
      if (complement) for (char &ch : r) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : 'N';
      out.swap(r);
    }
    return out;
  }
};

// BarcodeCorrector.hpp: whitelist kept in a 4-ary trie with a count per node. As in the reference a lookup does not ask
// whether the node ends a whitelist entry, so a proper prefix of an entry (and the empty string) counts as present.
struct BarcodeWhitelist {
  struct Node { int next[4] = {-1, -1, -1, -1}; int count = 0; };
  std::vector<Node> nodes = std::vector<Node>(1);
  static int num(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
  void insert(const std::string &s, int w) {
    for (char c : s) if (num(c) < 0) return;
    int p = 0;
    for (char c : s) { int t = num(c); if (nodes[p].next[t] < 0) { nodes[p].next[t] = (int)nodes.size(); nodes.emplace_back(); } p = nodes[p].next[t]; }
    nodes[p].count += w;
  }
  int searchAndUpdate(const std::string &s, int w) {   // count after the update, -1: not there
    for (char c : s) if (num(c) < 0) return -1;
    int p = 0;
    for (char c : s) { p = nodes[p].next[num(c)]; if (p < 0) return -1; }
    nodes[p].count += w;
    return nodes[p].count;
  }
  void load(const char *file) {   // one barcode per line, gz or plain (BarcodeCorrector.hpp:117-139)
    gzFile fp = gzopen(file, "r");
    if (!fp) { fprintf(stderr, "Could not open %s\n", file); exit(EXIT_FAILURE); }
    char buf[256];
    while (gzgets(fp, buf, sizeof buf)) {
      size_t l = strlen(buf);
      if (l > 0 && buf[l - 1] == '\n') buf[--l] = 0;
      insert(buf, 1);
    }
    gzclose(fp);
  }
  // -1: cannot be corrected, 0: on the list, 1: one base changed (BarcodeCorrector.hpp:156-217): among the single-base
  // substitutions that are on the list the most frequent one wins, ties go to the one at the lowest-quality position
  int correct(std::string &barcode, const std::string *qual) {
    if (searchAndUpdate(barcode, 0) != -1) return 0;
    struct Rec { int pos, base, cnt; };
    std::vector<Rec> recs;
    std::string t = barcode;
    for (size_t i = 0; i < barcode.size(); ++i)
      for (int j = 0; j < 4; ++j) {
        if ("ACGT"[j] == barcode[i]) continue;
        t[i] = "ACGT"[j];
        int cnt = searchAndUpdate(t, 0);
        t[i] = barcode[i];
        if (cnt != -1) recs.push_back(Rec{(int)i, j, cnt});
      }
    if (recs.empty()) return -1;
    auto q = [&](int pos) -> int { return (size_t)pos < qual->size() ? (int)(*qual)[pos] : 0; };
    int bestCnt = -1, bestTag = -1, bestLowQual = 255;
    for (size_t i = 0; i < recs.size(); ++i) {
      if (recs[i].cnt > bestCnt) { bestCnt = recs[i].cnt; bestTag = (int)i; if (qual) bestLowQual = q(recs[i].pos); }
      else if (recs[i].cnt == bestCnt && qual && q(recs[i].pos) < bestLowQual) { bestLowQual = q(recs[i].pos); bestTag = (int)i; }
    }
    barcode[recs[bestTag].pos] = "ACGT"[recs[bestTag].base];
    return 1;
  }
};

// BarcodeTranslator.hpp: lines "TO<sep>FROM" (sep = ',', tab or space); a barcode is cut into pieces of the FROM length
// (that of the last line read) and every piece replaced, joined by '-'; an unknown piece gives "".
struct BarcodeTranslate {
  std::unordered_map<std::string, std::string> table;
  int fromLen = -1;
  bool set = false;
  void load(const char *file) {
    set = true;
    gzFile fp = gzopen(file, "r");
    if (!fp) { fprintf(stderr, "Could not open %s\n", file); exit(EXIT_FAILURE); }
    char buf[512];
    while (gzgets(fp, buf, sizeof buf)) {
      size_t l = strlen(buf);
      if (l > 0 && buf[l - 1] == '\n') buf[--l] = 0;
      std::string line(buf);
      size_t i = 0;
      while (i < line.size() && line[i] != ',' && line[i] != '\t' && line[i] != ' ') ++i;
      std::string to = line.substr(0, i), from = i + 1 <= line.size() ? line.substr(i + 1) : std::string();
      fromLen = (int)line.size() - (int)i - 1;
      table[from] = to;
    }
    gzclose(fp);
  }
  std::string translate(const std::string &bc) const {
    std::string ret;
    if (fromLen <= 0) return ret;
    for (size_t i = 0; i < bc.size() / (size_t)fromLen; ++i) {
      auto it = table.find(bc.substr(i * (size_t)fromLen, (size_t)fromLen));
      if (it == table.end()) return std::string();
      if (i == 0) ret = it->second; else ret += "-" + it->second;
    }
    return ret;
  }
};

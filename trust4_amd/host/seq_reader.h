// trust4_amd/host/seq_reader.h -- FASTA / FASTQ (optionally gzip) reader shared by the host drivers.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

// ---- FASTA / FASTQ (optionally gzip) reader with kseq's record model ---------------------------------
struct SeqReader {
  std::vector<std::string> files;
  size_t cur = 0;
  gzFile fp = nullptr;
  std::string pending;   // look-ahead line
  bool havePending = false;
  std::string id, seq, qual, comment;   // comment: rest of the header line after the separator that ends the name (kseq)
  bool hasQual = false;
  bool stripMateSuffix = true;          // false: ids as written (what the reference's threaded extractor path prints)
  bool getLine(std::string &out) {
    if (havePending) { out.swap(pending); havePending = false; return true; }
    out.clear();
    char buf[1 << 16];
    bool any = false;
    while (gzgets(fp, buf, sizeof buf)) {
      any = true;
      size_t l = strlen(buf);
      bool eol = l > 0 && buf[l - 1] == '\n';
      while (l > 0 && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) buf[--l] = 0;
      out.append(buf, l);
      if (eol) break;
    }
    return any;
  }
  void rewind() { if (fp) { gzclose(fp); fp = nullptr; } cur = 0; havePending = false; }
  bool next() {
    for (;;) {
      if (!fp) {
        if (cur >= files.size()) return false;
        fp = gzopen(files[cur].c_str(), "rb");
        if (fp) gzbuffer(fp, 1 << 20);
        if (!fp) { fprintf(stderr, "Could not open %s\n", files[cur].c_str()); exit(EXIT_FAILURE); }
      }
      std::string line;
      bool got = false;
      while (getLine(line)) if (!line.empty() && (line[0] == '>' || line[0] == '@')) { got = true; break; }
      if (!got) { gzclose(fp); fp = nullptr; ++cur; havePending = false; continue; }
      size_t e = 1;
      while (e < line.size() && line[e] != ' ' && line[e] != '\t') ++e;
      id.assign(line, 1, e - 1);
      if (e < line.size()) comment.assign(line, e + 1, std::string::npos); else comment.clear();
      size_t n = id.size();   // ReadFiles.hpp:180-185 (Next() drops a /1 or /2; the batch reader NextWithBuffer, 200-241, does not)
      if (stripMateSuffix && n >= 2 && (id[n - 1] == '1' || id[n - 1] == '2') && id[n - 2] == '/') id.resize(n - 2);
      seq.clear(); qual.clear(); hasQual = false;
      bool plus = false;
      while (getLine(line)) {
        if (!line.empty() && (line[0] == '>' || line[0] == '@')) { pending.swap(line); havePending = true; break; }
        if (!line.empty() && line[0] == '+') { plus = true; break; }
        for (char c : line) if (c > ' ' && c < 127) seq.push_back(c);
      }
      if (plus) {
        hasQual = true;
        while (qual.size() < seq.size() && getLine(line)) qual += line;
      }
      return true;
    }
  }
};


// trust4_amd/host/seq_reader.h -- FASTA / FASTQ (optionally gzip) reader shared by the host drivers.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---- FASTA / FASTQ (optionally gzip) reader with kseq's record model ---------------------------------
struct SeqReader {
  std::vector<std::string> files;
  size_t cur = 0;
  gzFile fp = nullptr;
  std::string pending;   // look-ahead line
  std::string lineBuf;
  bool havePending = false;
  std::string id, seq, qual, comment;   // comment: rest of the header line after the separator that ends the name (kseq)
  bool hasQual = false;
  bool stripMateSuffix = true;          // false: ids as written (what the reference's threaded extractor path prints)
  // plain (not gzip) regular files are read in 4 MiB pieces and split at the newlines here; zlib's gzgets, which also passes plain
  // files through, costs several times the parsing itself
  FILE *raw = nullptr;
  std::vector<char> rbuf;
  size_t rpos = 0, rend = 0;
  bool getLineRaw(std::string &out) {
    bool any = false;
    for (;;) {
      if (rpos == rend) {
        if (rbuf.empty()) rbuf.resize((size_t)4 << 20);
        rend = fread(rbuf.data(), 1, rbuf.size(), raw);
        rpos = 0;
        if (rend == 0) break;
      }
      any = true;
      const char *b = rbuf.data() + rpos;
      const char *nl = (const char *)memchr(b, '\n', rend - rpos);
      if (!nl) { out.append(b, rend - rpos); rpos = rend; continue; }
      out.append(b, (size_t)(nl - b));
      rpos = (size_t)(nl - rbuf.data()) + 1;
      break;
    }
    while (!out.empty() && out.back() == '\r') out.pop_back();
    return any;
  }
  bool openNext() {
    const char *path = files[cur].c_str();
    struct stat st;
    if (stat(path, &st) == 0 && S_ISREG(st.st_mode)) {
      FILE *f = fopen(path, "rb");
      if (!f) return false;
      unsigned char magic[2] = {0, 0};
      const size_t got = fread(magic, 1, 2, f);
      if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) && fseek(f, 0, SEEK_SET) == 0) { raw = f; rpos = rend = 0; return true; }
      fclose(f);
    }
    fp = gzopen(path, "rb");
    if (fp) gzbuffer(fp, 1 << 20);
    return fp != nullptr;
  }
  bool isOpen() const { return fp != nullptr || raw != nullptr; }
  void closeCur() { if (fp) { gzclose(fp); fp = nullptr; } if (raw) { fclose(raw); raw = nullptr; } }
  bool getLine(std::string &out) {
    if (havePending) { out.swap(pending); havePending = false; return true; }
    out.clear();
    if (raw) return getLineRaw(out);
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
  void rewind() { closeCur(); cur = 0; havePending = false; }
  SeqReader() = default;
  SeqReader(const SeqReader &) = delete;              // (owns an open file)
  SeqReader &operator=(const SeqReader &) = delete;
  ~SeqReader() { closeCur(); }
  bool next() {
    for (;;) {
      if (!isOpen()) {
        if (cur >= files.size()) return false;
        if (!openNext()) { fprintf(stderr, "Could not open %s\n", files[cur].c_str()); exit(EXIT_FAILURE); }
      }
      std::string &line = lineBuf;   // (a member: its storage serves every line of the file)
      bool got = false;
      while (getLine(line)) if (!line.empty() && (line[0] == '>' || line[0] == '@')) { got = true; break; }
      if (!got) { closeCur(); ++cur; havePending = false; continue; }
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
        bool clean = true;   // (a line of printable characters only -- every line of a well-formed file -- is appended as it is)
        for (char c : line) clean &= (c > ' ' && c < 127);
        if (clean) seq += line;
        else for (char c : line) if (c > ' ' && c < 127) seq.push_back(c);
      }
      if (plus) {
        hasQual = true;
        while (qual.size() < seq.size() && getLine(line)) qual += line;
      }
      return true;
    }
  }
};


// The same reader on a thread of its own: records arrive in blocks through a bounded queue, so the files of one run (mate 1, mate 2,
// barcodes, UMIs) are read and split side by side while the driver's loop consumes them in lock-step. Same members as SeqReader
// for the consumer (files / next() / id / seq / qual / hasQual / comment).
struct ThreadedSeqReader {
  std::vector<std::string> files;
  bool stripMateSuffix = true;
  std::string id, seq, qual, comment;
  bool hasQual = false;
  struct Rec { std::string id, seq, qual, comment; bool hasQual; };
  typedef std::vector<Rec> Block;
  enum { BLOCK = 16384, DEPTH = 4 };
  bool next() {
    if (!started) start();
    if (at >= cur.size()) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !ready.empty() || done; });
      if (ready.empty()) return false;
      cur.swap(ready.front());
      ready.pop_front();
      at = 0;
      lk.unlock();
      cv.notify_all();
      if (cur.empty()) return false;
    }
    Rec &r = cur[at++];
    id.swap(r.id); seq.swap(r.seq); qual.swap(r.qual); comment.swap(r.comment); hasQual = r.hasQual;
    return true;
  }
  // a whole block of records at once (not to be mixed with next()): the consumer's per-record work can then run on its own threads
  bool nextBlock(Block &out) {
    if (!started) start();
    recycle(out);   // (what `out` still holds goes back to the reader)
    Block got;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !ready.empty() || done; });
      if (ready.empty()) { out.clear(); return false; }
      got.swap(ready.front());
      ready.pop_front();
    }
    cv.notify_all();
    out.swap(got);
    return !out.empty();
  }
  // A block the consumer is done with goes back to the reader, records and all: the reader fills the records again in place, their
  // strings keep their storage, and in the steady state nobody allocates or frees a string (the records a consumer swapped its own
  // strings into bring those along).
  void recycle(Block &b) {
    if (b.empty()) return;
    std::lock_guard<std::mutex> lk(mu);
    if (spare.size() < (size_t)DEPTH * 8) { spare.emplace_back(); spare.back().swap(b); }
  }
  ~ThreadedSeqReader() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
 private:
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Block> ready, spare;
  Block cur;
  size_t at = 0;
  bool started = false, done = false, quit = false;
  void start() {
    started = true;
    th = std::thread([this] {
      SeqReader in;
      in.files = files; in.stripMateSuffix = stripMateSuffix;
      for (;;) {
        Block b;
        { std::lock_guard<std::mutex> lk(mu); if (!spare.empty()) { b.swap(spare.back()); spare.pop_back(); } }
        b.reserve(BLOCK);
        size_t nb = 0;
        while (nb < (size_t)BLOCK && in.next()) {
          if (nb == b.size()) b.emplace_back();
          Rec &r = b[nb++];
          r.id.swap(in.id); r.seq.swap(in.seq); r.qual.swap(in.qual); r.comment.swap(in.comment); r.hasQual = in.hasQual;
        }
        b.resize(nb);
        const bool last = b.size() < (size_t)BLOCK;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ready.size() < (size_t)DEPTH || quit; });
        if (quit) return;
        if (!b.empty()) ready.push_back(std::move(b));
        if (last) { done = true; lk.unlock(); cv.notify_all(); return; }
        lk.unlock();
        cv.notify_all();
      }
    });
  }
};

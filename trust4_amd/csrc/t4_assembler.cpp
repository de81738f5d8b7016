// trust4_amd/csrc/t4_assembler.cpp -- host side of the order-dependent contig builder.
//
// The greedy assembly loop of stage 1 (main.cpp:1583-1880) is inherently sequential: every successful
// SeqSet::AddRead mutates the contig set the next read is matched against (hard part 1).
// This file is the ordered-commit half of that design: it owns the mutable contig set and its k-mer
// index on the host and replays the reference's bookkeeping exactly, while everything that is
// expensive and read-only per read -- GetOverlapsFromRead and the ExtendOverlap alignments -- is
// obtained from the GPU (t4_add_query* : GetOverlapsFromRead + the ExtendOverlap of every overlap) against a device
// image of the set, for a window of upcoming reads at a time.
//   * A set whose index is not keyed by barcode (bulk mode) is a LIVE set: the host index stores its posting lists in the
//     layout of the device image and every edit is shipped by position (t4_index_apply_delta); the window slides and an
//     entry is kept exactly as long as no commit can have changed its query (the rules and why they are safe).
//   * A cell of a per-barcode set (t4_cellset) has its image rebuilt after an observable change and its window ends there.
//
// Reference semantics followed: SeqSet::AddRead (SeqSet.hpp:3426-4473), RepeatAddRead (4477-4507),
// InputNovelRead (3028-3073), UpdateConsensus / UpdateAllConsensus (4525-4588), SubstituteConsensusPos
// (11058-11080), IsNameCompatible (3370-3419), Output (10939-10994), and KmerIndex::Insert / Remove /
// BuildIndexFromRead / UpdateIndexFromRead / RemoveIndexFromRead (KmerIndex.hpp:66-201).
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <tuple>
#include <vector>

#include "../../include/trust4_hip.h"
#include "t4_internal.h"

namespace {

// a posWeight column as the kernels see it (== t4PwByte, t4_device.h): bit x = (sum < 3 * count[x]), bit 4 = (sum == 0)
inline unsigned char t4PwByte(int a, int c, int g, int t) {
  const int sum = a + c + g + t;
  return (unsigned char)((sum == 0 ? 16 : 0) | (sum < 3 * a ? 1 : 0) | (sum < 3 * c ? 2 : 0) | (sum < 3 * g ? 4 : 0) | (sum < 3 * t ? 8 : 0));
}
inline int nucNum(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
const char NUM2NUC[4] = {'A', 'C', 'G', 'T'};

struct KCode {  // KmerCode.hpp
  int k, invalidPos;
  uint64_t code, mask;
  explicit KCode(int kl) : k(kl), invalidPos(-1), code(0), mask(kl < 32 ? ((1ull << (2 * kl)) - 1ull) : ~0ull) {}
  void restart() { code = 0; invalidPos = -1; }
  void append(char c) {
    if (invalidPos != -1) ++invalidPos;
    code = ((code << 2) & mask) | (uint64_t)(nucNum(c) & 3);
    if (c == 'N') invalidPos = 0;
    if (invalidPos >= k) invalidPos = -1;
  }
  bool valid() const { return invalidPos == -1; }
};

struct Post { int idx, offset; };
struct Key {
  uint64_t code; int h;
  bool operator==(const Key &o) const { return code == o.code && h == o.h; }
};
struct KeyHash {
  size_t operator()(const Key &k) const {
    uint64_t z = k.code * 1000003ull + (uint64_t)k.h;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return (size_t)(z ^ (z >> 31));
  }
};
inline uint64_t mix64h(uint64_t z) {   // == t4k::mix64 (t4_kernels.h): slot of a code in the device table of a live set
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// The mutable KmerIndex: a posting multiset per (code, bucket) with the reference's edit operations. The order inside a
// list is not kept (nothing downstream of SortHits observes it, SURVEY 7 hard part 4; Remove and UpdateIndexFromRead only
// ever pick among EQUAL postings), so a removal fills the hole with the last posting. Lists live in one flat arena whose
// layout IS the layout of the device image of a live set: `mirror` turns on the bookkeeping of what changed
// since the last t4_index_apply_delta (table slots and postings by position).
struct ListRef { uint32_t start = 0, cnt = 0, cap = 0; int64_t slot = -1; bool dirty = false; };
// (code, bucket) -> ListRef: open addressing over entry numbers, the entries themselves in a deque (their addresses are kept
// by the dirty-key list). The index edits of a commit are tens of thousands of lookups; a node-based map spent most of the
// time of those edits chasing its nodes.
struct KeyMap {
  struct value_type { Key first; ListRef second; bool live = false; };
  std::deque<value_type> store;
  std::vector<int32_t> table;      // entry number, -1 empty, -2 erased
  std::vector<int32_t> freeEntries;
  size_t nLive = 0, nSlotsUsed = 0;
  struct iterator {
    KeyMap *m; size_t i;
    value_type &operator*() const { return m->store[i]; }
    value_type *operator->() const { return &m->store[i]; }
    iterator &operator++() { ++i; while (i < m->store.size() && !m->store[i].live) ++i; return *this; }
    bool operator!=(const iterator &o) const { return i != o.i; }
    bool operator==(const iterator &o) const { return i == o.i; }
  };
  iterator begin() { size_t i = 0; while (i < store.size() && !store[i].live) ++i; return iterator{this, i}; }
  iterator end() { return iterator{this, store.size()}; }
  struct const_iterator {
    const KeyMap *m; size_t i;
    const value_type &operator*() const { return m->store[i]; }
    const value_type *operator->() const { return &m->store[i]; }
    const_iterator &operator++() { ++i; while (i < m->store.size() && !m->store[i].live) ++i; return *this; }
    bool operator!=(const const_iterator &o) const { return i != o.i; }
    bool operator==(const const_iterator &o) const { return i == o.i; }
  };
  const_iterator begin() const { size_t i = 0; while (i < store.size() && !store[i].live) ++i; return const_iterator{this, i}; }
  const_iterator end() const { return const_iterator{this, store.size()}; }
  size_t size() const { return nLive; }
  void clear() { store.clear(); table.clear(); freeEntries.clear(); nLive = nSlotsUsed = 0; }
  int64_t probe(const Key &k) const {   // entry number or -1
    if (table.empty()) return -1;
    const size_t mask = table.size() - 1;
    for (size_t s = KeyHash()(k) & mask;; s = (s + 1) & mask) {
      const int32_t e = table[s];
      if (e == -1) return -1;
      if (e >= 0 && store[(size_t)e].first == k) return e;
    }
  }
  void rehash(size_t n) {
    table.assign(n, -1);
    const size_t mask = n - 1;
    for (size_t e = 0; e < store.size(); ++e) if (store[e].live) {
      size_t s = KeyHash()(store[e].first) & mask;
      while (table[s] != -1) s = (s + 1) & mask;
      table[s] = (int32_t)e;
    }
    nSlotsUsed = nLive;
  }
  iterator find(const Key &k) { const int64_t e = probe(k); return e < 0 ? end() : iterator{this, (size_t)e}; }
  const_iterator find(const Key &k) const { const int64_t e = probe(k); return e < 0 ? end() : const_iterator{this, (size_t)e}; }
  std::pair<iterator, bool> emplace(const Key &k, const ListRef &v) {
    const int64_t e0 = probe(k);
    if (e0 >= 0) return std::make_pair(iterator{this, (size_t)e0}, false);
    if (2 * (nSlotsUsed + 1) > table.size()) rehash(table.empty() ? 1024 : (2 * (nLive + 1) > table.size() / 2 ? table.size() * 2 : table.size()));
    size_t e;
    if (!freeEntries.empty()) { e = (size_t)freeEntries.back(); freeEntries.pop_back(); }
    else { e = store.size(); store.emplace_back(); }
    value_type &x = store[e];
    x.first = k; x.second = v; x.live = true;
    const size_t mask = table.size() - 1;
    size_t s = KeyHash()(k) & mask;
    while (table[s] >= 0) s = (s + 1) & mask;
    if (table[s] == -1) ++nSlotsUsed;
    table[s] = (int32_t)e;
    ++nLive;
    return std::make_pair(iterator{this, e}, true);
  }
  void erase(iterator it) {
    const size_t mask = table.size() - 1;
    for (size_t s = KeyHash()(it->first) & mask;; s = (s + 1) & mask) if (table[s] == (int32_t)it.i) { table[s] = -2; break; }
    store[it.i].live = false;
    freeEntries.push_back((int32_t)it.i);
    --nLive;
  }
};
struct IndexListener {
  virtual void onInsert(uint64_t code, int h, int idx, int off, uint32_t sizeAfter) = 0;
  virtual void onRemove(uint64_t code, int h, int idx, int off, uint32_t sizeAfter) = 0;
  virtual void onMove(uint64_t code, int h, int oldIdx, int oldOff, int idx, int off) = 0;
  virtual ~IndexListener() {}
};
struct HostIndex {
  int k; bool considerBarcode = false;
  IndexListener *hook = nullptr;
  typedef KeyMap Map;
  Map map;
  std::vector<Post> arena;
  size_t arenaUsed = 0, garbage = 0;
  size_t total = 0;
  // ---- device mirror bookkeeping (live sets only)
  bool mirror = false;
  std::vector<unsigned char> tabOcc;          // occupancy of the device table's slots
  int64_t tabSlots = 0, tabKeys = 0;
  bool tabRebuilt = false;                    // every slot goes into the next delta
  std::vector<Map::value_type *> dirtyKeys;
  std::vector<uint32_t> dirtyPost;
  std::vector<unsigned char> dirtyPostFlag;
  explicit HostIndex(int kl) : k(kl) {}
  // ---- where a contig's postings are (round 6). KmerIndex::Remove and UpdateIndexFromRead (KmerIndex.hpp:143-201) walk a posting list
  // until they meet (idx, offset); with k = 9 the lists of a gene segment that thousands of contigs share hold thousands of postings, and
  // a left extension rewrites every posting of its contig: 26 us per extension on config C2, the largest item of the chain's host thread.
  // hint[idx][offset] is the arena position of a posting (idx, offset) and the k-mer code of the list it stands in. It is kept current
  // through every move of a posting (a list that moves to a larger place, the last posting of a list that fills a hole) and cleared when
  // its posting leaves or is rewritten, so a hint that is set names a LIVE posting of the list of `code`. A lookup takes it only if it
  // names the list of the k-mer in hand and holds (idx, offset) -- any such posting is as good as the one the walk would have met first
  // (the reference only ever picks among equal postings of one list) -- and walks the list as before otherwise.
  static constexpr uint32_t NOPOS = 0xFFFFFFFFu;
  struct Hint { uint32_t pos, codeLo, codeHi; };
  std::vector<std::vector<Hint>> hint;
  int64_t hintHits = 0, hintWalks = 0, walkSteps = 0;
  uint32_t hintGet(int idx, int off) const {
    if (idx < 0 || off < 0 || (size_t)idx >= hint.size() || (size_t)off >= hint[(size_t)idx].size()) return NOPOS;
    return hint[(size_t)idx][(size_t)off].pos;
  }
  void hintSet(int idx, int off, uint32_t pos, uint64_t code) {
    if (idx < 0 || off < 0) return;
    if ((size_t)idx >= hint.size()) hint.resize((size_t)idx + 1 + ((size_t)idx >> 3));
    std::vector<Hint> &v = hint[(size_t)idx];
    if ((size_t)off >= v.size()) v.resize((size_t)off + 64, Hint{NOPOS, 0u, 0u});
    v[(size_t)off] = Hint{pos, (uint32_t)code, (uint32_t)(code >> 32)};
  }
  void hintMoved(const Post &p, uint32_t from, uint32_t to) { if (hintGet(p.idx, p.offset) == from) hint[(size_t)p.idx][(size_t)p.offset].pos = to; }
  // the hint of (idx, off) when it names a posting (idx, off) of the list of `code`: no look at the key map at all
  uint32_t hintOf(int idx, int off, uint64_t code) const {
    if (idx < 0 || off < 0 || (size_t)idx >= hint.size() || (size_t)off >= hint[(size_t)idx].size()) return NOPOS;
    const Hint &hp = hint[(size_t)idx][(size_t)off];
    if (hp.pos == NOPOS || hp.codeLo != (uint32_t)code || hp.codeHi != (uint32_t)(code >> 32)) return NOPOS;
    return arena[hp.pos].idx == idx && arena[hp.pos].offset == off ? hp.pos : NOPOS;
  }
  // position in list l of a posting (idx, off), NOPOS when the list holds none
  uint32_t findPost(const ListRef &l, int idx, int off) {
    const uint32_t h = hintGet(idx, off);
    if (h != NOPOS && h - l.start < l.cnt && arena[h].idx == idx && arena[h].offset == off) { ++hintHits; return h; }
    ++hintWalks;
    for (uint32_t i = 0; i < l.cnt; ++i) if (arena[l.start + i].idx == idx && arena[l.start + i].offset == off) { walkSteps += i + 1; return l.start + i; }
    walkSteps += l.cnt;
    return NOPOS;
  }
  int bucket(uint64_t code, int barcode) const { return (int)((code + (uint64_t)(int64_t)(considerBarcode ? barcode + 1 : 0)) % 1000003ull); }
  void markPost(uint32_t at) {
    if (!mirror) return;
    if (dirtyPostFlag.size() < arena.size()) dirtyPostFlag.resize(arena.size(), 0);
    if (!dirtyPostFlag[at]) { dirtyPostFlag[at] = 1; dirtyPost.push_back(at); }
  }
  void markKey(Map::value_type *kv) { if (mirror && !kv->second.dirty) { kv->second.dirty = true; dirtyKeys.push_back(kv); } }
  int64_t probeSlot(uint64_t code) {
    uint64_t s = mix64h(code) & (uint64_t)(tabSlots - 1);
    while (tabOcc[s]) s = (s + 1) & (uint64_t)(tabSlots - 1);
    tabOcc[s] = 1;
    return (int64_t)s;
  }
  void growTable() {   // load <= 1/2; every key gets a new slot
    tabSlots = tabSlots ? tabSlots * 2 : ((int64_t)1 << 16);
    while (tabSlots < 2 * (tabKeys + 1)) tabSlots *= 2;
    tabOcc.assign((size_t)tabSlots, 0);
    for (auto &kv : map) kv.second.slot = probeSlot(kv.first.code);
    tabRebuilt = true;
  }
  Map::value_type *listOf(const Key &key) {
    auto ins = map.emplace(key, ListRef());
    Map::value_type *kv = &*ins.first;
    if (ins.second && mirror) {
      ++tabKeys;
      if (2 * tabKeys > tabSlots) growTable(); else kv->second.slot = probeSlot(key.code);
    }
    return kv;
  }
  void insert(const KCode &kc, int idx, int off, int barcode) {
    if (!kc.valid()) return;
    const int h = bucket(kc.code, barcode);
    Map::value_type *kv = listOf(Key{kc.code, h});
    ListRef &l = kv->second;
    if (l.cnt == l.cap) {   // move the list to the end of the arena with twice the room
      const uint32_t ncap = l.cap ? l.cap * 2 : 2;
      if (arenaUsed + ncap > arena.size()) arena.resize((arenaUsed + ncap) * 2 > 1024 ? (arenaUsed + ncap) * 2 : 1024);
      for (uint32_t i = 0; i < l.cnt; ++i) { arena[arenaUsed + i] = arena[l.start + i]; markPost((uint32_t)arenaUsed + i); hintMoved(arena[l.start + i], l.start + i, (uint32_t)arenaUsed + i); }
      garbage += l.cap;
      l.start = (uint32_t)arenaUsed; l.cap = ncap; arenaUsed += ncap;
    }
    arena[l.start + l.cnt] = Post{idx, off};
    markPost(l.start + l.cnt);
    hintSet(idx, off, l.start + l.cnt, kc.code);
    ++l.cnt; ++total;
    markKey(kv);
    if (hook) hook->onInsert(kc.code, h, idx, off, l.cnt);
  }
  void remove(const KCode &kc, int idx, int off, int barcode) {  // one posting equal to (idx, off)
    if (!kc.valid()) return;
    const int h = bucket(kc.code, barcode);
    auto it = map.find(Key{kc.code, h});
    if (it == map.end()) return;
    ListRef &l = it->second;
    const uint32_t at = findPost(l, idx, off);
    if (at != NOPOS) {
      const uint32_t last = l.start + l.cnt - 1;
      if (hintGet(idx, off) == at) hint[(size_t)idx][(size_t)off].pos = NOPOS;   // (an equal posting elsewhere in the list is found by the walk when it is asked for)
      if (at != last) { arena[at] = arena[last]; markPost(at); hintMoved(arena[at], last, at); }
      --l.cnt; --total;
      markKey(&*it);
      if (hook) hook->onRemove(kc.code, h, idx, off, l.cnt);
    }
    if (l.cnt == 0 && !mirror) { garbage += l.cap; map.erase(it); }   // a live set keeps the key (its table slot) with an empty list
  }
  double secOps = 0;   // seconds in build / update / removeSeq (live sets report it)
  struct OpTimer { double &acc; std::chrono::steady_clock::time_point t0; explicit OpTimer(double &a) : acc(a), t0(std::chrono::steady_clock::now()) {} ~OpTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } };
  void build(const char *s, int len, int id, int barcode, int shift = 0) {  // BuildIndexFromRead
    if (len < k) return;
    OpTimer tm_(secOps);
    KCode kc(k), prev(k);
    int i;
    for (i = 0; i < k - 1; ++i) kc.append(s[i]);
    for (; i < len; ++i) {
      kc.append(s[i]);
      if (kc.valid() && (i == k || kc.code != prev.code)) insert(kc, id, i - k + 1 + shift, barcode);
      prev = kc;
    }
  }
  void update(const char *s, int len, int barcode, int shift, int oldId, int id) {  // UpdateIndexFromRead
    if (len < k) return;
    OpTimer tm_(secOps);
    KCode kc(k);
    int i;
    for (i = 0; i < k - 1; ++i) kc.append(s[i]);
    // (the hints of oldId are in the old coordinates; what BuildIndexFromRead has just inserted for the prepended bases is in the new
    // ones, below `shift`. The rewritten postings' hints are collected apart and take their place afterwards.)
    static thread_local std::vector<uint32_t> moved;
    static thread_local std::vector<uint64_t> movedCode;
    moved.assign((size_t)(len > 0 ? len : 0), NOPOS);
    movedCode.assign(moved.size(), ~0ull);
    // Pass 1: the k-mer of every offset and where its hint says the posting stands -- the postings of one contig lie all over the arena
    // (each in the list of its k-mer), so every one of them is a cache miss, and so is its dirty flag: both are asked for ahead of pass 2
    // (600 independent misses overlap instead of queueing up: a left extension was 25 us of them).
    const std::vector<Hint> *hv = (oldId >= 0 && (size_t)oldId < hint.size()) ? &hint[(size_t)oldId] : nullptr;
    for (; i < len; ++i) {
      kc.append(s[i]);
      if (!kc.valid()) continue;
      const int off = i - k + 1;
      movedCode[(size_t)off] = kc.code;
      if (hv && (size_t)off < hv->size()) {
        const Hint &hp = (*hv)[(size_t)off];
        if (hp.pos != NOPOS && hp.codeLo == (uint32_t)kc.code && hp.codeHi == (uint32_t)(kc.code >> 32)) {
          moved[(size_t)off] = hp.pos;
          __builtin_prefetch(&arena[hp.pos], 1);
          if (mirror && hp.pos < dirtyPostFlag.size()) __builtin_prefetch(&dirtyPostFlag[hp.pos], 1);
        }
      }
    }
    // Pass 2: the rewrite, offset by offset as UpdateIndexFromRead goes
    for (int off = 0; off + k <= len; ++off) {
      const uint64_t code = movedCode[(size_t)off];
      if (code == ~0ull) continue;   // (no valid k-mer here)
      const int h = bucket(code, barcode);
      uint32_t at = moved[(size_t)off];
      moved[(size_t)off] = NOPOS;
      if (at != NOPOS && arena[at].idx == oldId && arena[at].offset == off) ++hintHits;
      else {
        auto it = map.find(Key{code, h});
        if (it == map.end()) continue;
        at = findPost(it->second, oldId, off);
        if (at == NOPOS) continue;
      }
      Post &p = arena[at];
      p.idx = id; p.offset += shift; markPost(at);
      moved[(size_t)off] = at;
      if (hook) hook->onMove(code, h, oldId, off, id, p.offset);
    }
    // hints: those of the old coordinates go (a cell that named a rewritten posting is stale either way), the rewritten ones come in
    // (a cell names a posting (oldId, its offset) only: the cells of the rewritten postings are exactly those at the offsets they had)
    for (int o = 0; o < len; ++o) {
      if (moved[(size_t)o] == NOPOS) continue;
      if (hintGet(oldId, o) == moved[(size_t)o]) hint[(size_t)oldId][(size_t)o].pos = NOPOS;
      hintSet(id, o + shift, moved[(size_t)o], movedCode[(size_t)o]);
    }
  }
  void removeSeq(const char *s, int len, int id, int barcode, int offset) {  // RemoveIndexFromRead
    if (len < k) return;
    OpTimer tm_(secOps);
    KCode kc(k);
    int i;
    for (i = 0; i < k - 1; ++i) kc.append(s[i]);
    for (; i < len; ++i) { kc.append(s[i]); if (kc.valid()) remove(kc, id, i - k + 1 + offset, barcode); }
  }
  const ListRef *find(uint64_t code, int h) const { auto it = map.find(Key{code, h}); return it == map.end() ? nullptr : &it->second; }
};

struct PosWeight { int c[4]; };
struct Seq {
  std::string name, cons;
  std::vector<PosWeight> pw;
  bool released = false;
  // live set: postings (this contig, offset) the index holds, per offset -- shipped as "posting marks" in bits 5-6 of the predicate
  // bytes (t4_device.h T4_PW_MARK_*), so that a restricted re-query reads a contig's postings off the contig itself
  std::vector<uint8_t> postCnt;
  bool marksBad = false;   // an offset held more than three postings once: the marks of this contig are not used
  bool pwTouched = true;   // posWeight or consensus changed since UpdateAllConsensus last looked at this contig (it is idempotent on a contig that did not)
  bool frozen = false;   // ReleaseFinishedBarcodeSeq: out of the index, posWeight final (SeqSet.hpp:10847-10935)
  int minLeftExtAnchor = 0, minRightExtAnchor = 0, barcode = -1, numRead = 0;
  // live set: place in the device arena of consensus chars / predicate bytes, and what of it changed since the last delta
  int64_t baseOff = -1; int baseCap = 0;
  bool devDirty = false; int dLo = 0, dHi = 0;
};

struct Ov {  // struct _overlap fields that the Add path reads or writes
  int seqIdx = -1, readStart = -1, readEnd = -1, seqStart = -1, seqEnd = -1, strand = 1, matchCnt = 0, indelCnt = 0;
  double similarity = 0;
};
bool ovLess(const Ov &a, const Ov &b) {  // _overlap::operator< (SeqSet.hpp:104-128)
  if (a.matchCnt != b.matchCnt) return a.matchCnt > b.matchCnt;
  if (a.similarity != b.similarity) return a.similarity > b.similarity;
  if (a.readEnd - a.readStart != b.readEnd - b.readStart) return a.readEnd - a.readStart > b.readEnd - b.readStart;
  if (a.seqIdx != b.seqIdx) return a.seqIdx < b.seqIdx;
  if (a.strand != b.strand) return a.strand < b.strand;
  if (a.readStart != b.readStart) return a.readStart < b.readStart;
  if (a.readEnd != b.readEnd) return a.readEnd < b.readEnd;
  if (a.seqStart != b.seqStart) return a.seqStart < b.seqStart;
  return a.seqEnd < b.seqEnd;
}
Ov fromT4(const t4_overlap &o) {
  Ov v; v.seqIdx = o.seqIdx; v.readStart = o.readStart; v.readEnd = o.readEnd; v.seqStart = o.seqStart; v.seqEnd = o.seqEnd;
  v.strand = o.strand; v.matchCnt = o.matchCnt; v.indelCnt = o.indelCnt; v.similarity = o.similarity;
  return v;
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
// IsNameCompatible (SeqSet.hpp:3370-3419): b comes after a
bool nameCompatible(const std::string &a, const std::string &b) {
  int maxA = -1, minB = 10;
  auto each = [](const std::string &s, auto fn) {
    size_t i = 0;
    while (i < s.size()) {
      if (s[i] == '+') { ++i; continue; }
      size_t j = i;
      while (j < s.size() && s[j] != '+') ++j;
      std::string part = s.substr(i, j - i);
      part.append(8, '\0');
      fn(geneType(part.c_str()));
      i = j;
    }
  };
  each(a, [&](int gt) { if (gt > maxA) maxA = gt; });
  each(b, [&](int gt) { if (gt < minB && gt != -1) minB = gt; });
  return maxA <= minB;
}

// fn(i) for i in [0, n) on up to `threads` host threads (dynamic chunks); fn must only touch data owned by item i
template <class F> void parallelFor(int n, int threads, F fn) {
  if (threads <= 1 || n <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
  if (threads > n) threads = n;
  std::atomic<int> next(0);
  const int chunk = n / (threads * 8) > 0 ? n / (threads * 8) : 1;
  auto body = [&]() { for (;;) { int b = next.fetch_add(chunk); if (b >= n) break; int e = b + chunk < n ? b + chunk : n; for (int i = b; i < e; ++i) fn(i); } };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(body);
  body();
  for (auto &th : pool) th.join();
}

void reverseComplement(std::string &rc, const std::string &s) {
  size_t n = s.size();
  rc.resize(n);
  for (size_t i = 0; i < n; ++i) { char c = s[n - 1 - i]; rc[i] = c != 'N' ? NUM2NUC[3 - nucNum(c)] : 'N'; }
}

}  // namespace

// ---- speculation window of a live set -----------------------------------------------------------------
// What the cached query result of an upcoming read depends on, tracked exactly enough to be safe:
//   * the posting lists of the read's own k-mers (both strands): every insertion / removal of such a key is examined;
//   * for a contig the read has three or more hits with on one strand (fewer can never form a candidate run,
//     SeqSet.hpp:923-925), the posWeight columns and the contig ends within reach of the read (hull of its hits +- read length).
// Per (strand, contig) the number of hits and the hull of their offsets, from the host replica of the index (a superset
// of what GetHitsFromRead emits: no repeat skipping, no barcode filter).
struct Grp { uint32_t key; uint32_t cnt; int32_t lo, hi; };   // key = contig * 2 + (strand == 1); 0xFFFFFFFF = empty
struct GroupTable {
  std::vector<Grp> t;
  uint32_t mask = 0, n = 0;
  void reset(uint32_t expect) {
    uint32_t sz = 64;
    while (sz < 2 * expect + 2) sz <<= 1;
    if (t.size() != sz) t.resize(sz);
    for (Grp &g : t) g.key = 0xFFFFFFFFu;
    mask = sz - 1; n = 0;
  }
  static uint32_t hashOf(uint32_t k) { k *= 0x9E3779B1u; return k ^ (k >> 15); }
  Grp *find(uint32_t key) {
    if (t.empty()) return nullptr;
    for (uint32_t s = hashOf(key) & mask;; s = (s + 1) & mask) {
      if (t[s].key == key) return &t[s];
      if (t[s].key == 0xFFFFFFFFu) return nullptr;
    }
  }
  Grp &get(uint32_t key) {
    if (t.empty() || 2 * (n + 1) > t.size()) grow();
    for (uint32_t s = hashOf(key) & mask;; s = (s + 1) & mask) {
      if (t[s].key == key) return t[s];
      if (t[s].key == 0xFFFFFFFFu) { t[s].key = key; t[s].cnt = 0; t[s].lo = 0x7FFFFFFF; t[s].hi = -0x7FFFFFFF; ++n; return t[s]; }
    }
  }
  void grow() {
    std::vector<Grp> old; old.swap(t);
    uint32_t sz = old.empty() ? 64 : (uint32_t)old.size() * 2;
    t.resize(sz);
    for (Grp &g : t) g.key = 0xFFFFFFFFu;
    mask = sz - 1; n = 0;
    for (const Grp &g : old) if (g.key != 0xFFFFFFFFu) { Grp &d = get(g.key); d.cnt = g.cnt; d.lo = g.lo; d.hi = g.hi; }
  }
};

// Host threads that help the caller with one job at a time (the dependency sets of a query round are derived while the GPU
// runs the query): started once, parked on a condition variable in between.
struct HelperPool {
  std::mutex mu;
  std::condition_variable cvJob, cvDone;
  std::vector<std::thread> th;
  const std::function<void()> *job = nullptr;
  long long gen = 0;
  int wanted = 0, running = 0;
  bool quit = false;
  ~HelperPool() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; ++gen; }
    cvJob.notify_all();
    for (auto &t : th) t.join();
  }
  void worker(int rank) {
    long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cvJob.wait(lk, [&] { return gen != seen; });
      seen = gen;
      if (quit) return;
      if (rank >= wanted) continue;
      const std::function<void()> *fn = job;
      lk.unlock();
      (*fn)();
      lk.lock();
      if (--running == 0) cvDone.notify_all();
    }
  }
  void start(int n, const std::function<void()> &fn) {   // n helpers run fn; the caller goes on and calls wait() later
    while ((int)th.size() < n) { const int rank = (int)th.size(); th.emplace_back([this, rank] { worker(rank); }); }
    { std::lock_guard<std::mutex> lk(mu); job = &fn; wanted = n; running = n; ++gen; }
    cvJob.notify_all();
  }
  void wait() { std::unique_lock<std::mutex> lk(mu); cvDone.wait(lk, [&] { return running == 0; }); }
};

// Where the one host thread of the ordered chain spends its time (T4_TIMING): time-stamp counter sections, a few nanoseconds each
#if defined(__x86_64__)
#include <x86intrin.h>
static inline uint64_t t4Tick() { return __rdtsc(); }
#else
static inline uint64_t t4Tick() { return (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(); }
#endif
enum { TS_ADD_SERVE, TS_ADD_CANDS, TS_ADD_DECIDE, TS_ADD_MERGE, TS_ADD_EXTEND, TS_ADD_BUMP, TS_REPEAT, TS_NOVEL, TS_EVENTS_STRUCT, TS_EVENTS_INDEX, TS_HARVEST_WAIT, TS_HARVEST_COPY,
       TS_HARVEST_MERGE, TS_ANNOUNCE, TS_PUMP_OTHER, TS_LAUNCH_PACK, TS_LAUNCH_CALL, TS_UPDATE_CONS, TS_DELTA, TS_LAUNCH_GROUPS, TS_EXT_STRINGS, TS_EXT_BUILD, TS_EXT_UPDATE, TS_EXT_PW, TS_EXT_SUBST, TS_N };
static const char *const TS_NAMES[TS_N] = {"addRead: serving the entry", "addRead: candidate list", "addRead: decision loops", "addRead: contig merge", "addRead: extension", "addRead: weights + N fill",
                                           "RepeatAddRead", "InputNovelRead", "events: structural", "events: index", "harvest: waiting for the device", "harvest: whole-query records",
                                           "harvest: restricted merges", "announcement", "pump: choosing what to launch", "launch: packing the items", "launch: the query call", "UpdateAllConsensus", "launch: delta", "launch: dependency sets beside the kernels", "extension: strings", "extension: BuildIndexFromRead of the new ends", "extension: UpdateIndexFromRead", "extension: weights moved", "extension: SubstituteConsensusPos"};
struct TscSections {
  uint64_t acc[TS_N] = {}, t0 = 0, tStart = 0;
  std::chrono::steady_clock::time_point wall0 = std::chrono::steady_clock::now();
  TscSections() { tStart = t4Tick(); }
  void begin() { t0 = t4Tick(); }
  void lap(int sec) { const uint64_t t = t4Tick(); acc[sec] += t - t0; t0 = t; }
  double secondsPerTick() const { const double w = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count(); const uint64_t d = t4Tick() - tStart; return d ? w / (double)d : 0; }
};

#define T4_MAX_READ_KMERS 512   // k-mer positions of a read (reads are at most 384 bases: t4_device.h T4_MAXL)
struct t4_cellset;
struct t4_assembler : IndexListener {
  TscSections ts;
  t4_ctx *ctx;
  t4_index *dev = nullptr;   // device image of the current set
  t4_cellset *owner = nullptr;   // cell of a per-barcode set: the image lives in the owner's arena slot
  int slot = -1, cellBarcode = -1;
  // cell mode: IsBaseEqual flips since the image was staged (most changes between two queries of a cell are single columns)
  struct PwPatch { int seq, pos; unsigned char val; };
  std::vector<PwPatch> patches;
  std::vector<int64_t> imgPwOff;   // byte offset of every contig's posWeight bytes inside the staged image
  int64_t windowStamp = -1;
  bool dirty = true;
  int k, radius = 10, hitLenRequired = 31;
  double novelSim = 0.9;
  std::vector<Seq> seqs;
  HostIndex index;
  Ov prevAdd;
  std::string err;
  int64_t queries = 0, refreshes = 0, cacheHits = 0;
  double secRefresh = 0, secQuery = 0;
  // speculation window: query results of upcoming reads. Cells: valid until any change a query of the cell can observe.
  struct Cached {
    std::string read; int strand, barcode, skip; int32_t cnt; bool valid; std::vector<t4_overlap> ov, ext; std::vector<int32_t> extRet;
    // live sets
    int64_t uid = 0;
    bool registered = false;     // its k-mers are in winKmers (done beside its first query: nothing looks an entry up before it holds a result)
    unsigned char tier = 0;      // the last query of this read ended on the global-scratch tier
    bool hintPredicted = false;  // tier was set (or left) by a count of the read's emitted hits before its first query (launchOn: the reads of the NEXT whole-query round)
    unsigned char lastKill = 0;  // why its last result fell whole (0: it never held one; 1 key, 2 list >= 100, 3 region, 4 shift, 5 contig, 6 tolerance, 7 a restricted re-query fell back): reporting only
    bool fragile = false;        // any change of one of its keys' lists invalidates it
    int slack = 0;               // tolerated hit-set changes left before possibleOverlapCnt could pass 100 (SeqSet.hpp:813-823)
    int lastUs = 0;              // what this read's last query took in its workgroup (microseconds; 0: never queried): a launch lasts as long as its slowest read
    bool statsStable = false;    // the query itself found that edits of small groups cannot move novelMinHitRequired (T4QueryArgs::statsStable): exact, needs no slack
    GroupTable groups;
    // Dependency records that came back from the wide query (csrc/t4_wide.h) instead of being derived here: the groups of the hits
    // the query EMITTED (after the repeat-skip rule), sorted by contig within a strand, minus strand (even keys) first. Counts of
    // groups below three hits only ever go up under later index edits (an edit of a k-mer the rule passed over is counted as if it
    // had been emitted, a removal is not subtracted): a superset, as the table above is.
    std::vector<Grp> devGroups;
    size_t devSplit = 0;          // first plus-strand record
    bool hasDev = false;
    // A read with lists beyond 10000 postings (fragile): per device record the group's hits of shorter lists (capped at 4, bits 0-2) and
    // whether its first hit is one (bit 3) -- what the statistics loop reads of a group for removeOnlyRepeats (SeqSet.hpp:796-806) --,
    // removeOnlyRepeats per strand as the query found it, and the groups this commit's edits touched (processEvents: such an entry stands
    // while the loop, repeated over its groups, gives the same threshold and the same removeOnlyRepeats and no edit lies within the first
    // `largest group` hits of the hit array, which the run test of 934-940 reads)
    std::vector<uint8_t> devInfo;
    bool ror0[2] = {false, false}, rorOk = false;
    std::vector<uint32_t> editedKeys;
    std::vector<Grp> dbgGroups;   // T4_VERIFY_WINDOW: what buildGroups derives for a read the wide query serves
    bool hostRecords = false;     // ... derived on the host (buildGroups: the emitted hits of a read of the LDS tier), not returned by the wide query
    bool expectWide = false;      // the launch expects the wide query to serve this read: no table derived beside the launch
    Grp *findGroup(uint32_t key) {
      if (hasDev) {
        size_t lo = (key & 1u) ? devSplit : 0, hi = (key & 1u) ? devGroups.size() : devSplit;
        while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (devGroups[mid].key < key) lo = mid + 1; else hi = mid; }
        if (lo < ((key & 1u) ? devGroups.size() : devSplit) && devGroups[lo].key == key) return &devGroups[lo];
      }
      return groups.find(key);
    }
    Grp &getGroup(uint32_t key) { Grp *g = findGroup(key); return g ? *g : groups.get(key); }
    // ---- restricted re-query. The full query left: nAll overlaps on the strand of the best one before the similarity
    // cut, nOther on the other strand, that strand, n4 groups of four or more hits. While no overlap exists on the other strand,
    // at most 50 on the best one's (the order-dependent pre-filters of SeqSet.hpp:1705-1794 stay off) and at most 100 groups of four
    // hits (novelMinHitRequired stays 3, SeqSet.hpp:813-823), what the read's result takes from one contig is independent of every
    // other contig: a commit that touches contig c of such an entry leaves the entry PARTIAL -- it keeps its records of the other
    // contigs -- and only the overlaps with c are asked for again (t4_add_query_pool_begin, only_seq).
    int nAll = 0, nOther = 0, n4 = 0, nAllBound = 0, restrictedCount = 0;
    // ---- the candidate store: what lets an entry with MORE than 50 candidate overlaps, or with more than 100 groups of
    // four hits, keep its other contigs when one contig changes. cands: every overlap of the read on the strand of the best one as it
    // stood before the similarity cut, in the order of the scan of SeqSet.hpp:1673-2094, with its pre-score key, its scored fields and
    // whether the pre-filters of 1705-1794 cut it (from the query itself: t4_add_query_last_cands). After a restricted re-query of
    // contig c the host swaps c's candidates, repeats the scan (mergeRestricted) and needs the whole query only when a candidate of
    // ANOTHER contig that was cut now passes (its ExtendOverlap record was never made). Group statistics (784-823): bounds of the
    // true counts per strand -- groups of >= 4 hits, of >= 5 hits, the largest group -- and the novelMinHitRequired T the entry's
    // candidates were made with; a restricted re-query runs with T and is accepted only if the bounds, updated with the contig's old
    // and new group sizes, still certify T (the certificate of overlapsFromKeys: both corners give the same threshold).
    std::vector<t4_cand> cands;
    bool candOk = false;
    int n4lo[2] = {0, 0}, n4hi[2] = {0, 0}, n5lo[2] = {0, 0}, n5hi[2] = {0, 0}, smlo[2] = {0, 0}, smhi[2] = {0, 0}, minT[2] = {3, 3};
    int toleratedSince = 0;            // index edits of small groups this entry has tolerated since its whole query
    // Are the recorded sizes of the entry's groups still EXACT (the hits the query would emit now)? A tolerated index edit is booked
    // exactly when the edited k-mer is certain to be emitted by GetHitsFromRead: its list holds fewer than 100 postings before and after
    // (the repeat-skip rule of SeqSet.hpp:1381-1391 only passes over lists of 100 and more) and the read has no two equal k-mers within
    // k / 2 + 1 positions (the rule's other test: a k-mer equal to the last one looked up is not looked up again). Any other tolerated
    // edit leaves the sizes as bounds -- `inexact` -- until the entry's next whole query. While they are exact, the statistics loop of
    // SeqSet.hpp:784-811 can be repeated on the host at any time (exactStats): an entry whose threshold the query could not certify
    // against edits of small groups (statsStable false) is then CHECKED after such an edit instead of spending a budget and falling.
    bool inexact = false, repeatNear = false, checkPending = false;
    // Which k-mer positions GetHitsFromRead LOOKS UP and does not pass over (SeqSet.hpp:1367-1391), per strand (0: the read as given,
    // 1: its reverse complement), replayed on the host against the index the entry's query ran on (buildGroups / emittedHits). The
    // decisions only move when a list crosses 100 postings, which ends the entry (processEvents); until then an index edit of a key
    // changes this read's hits exactly at the occurrences whose position is in the mask -- and not at all at the others.
    uint64_t emitMask[2][(T4_MAX_READ_KMERS + 63) / 64];
    bool maskOk = false;
    std::vector<uint32_t> exactKeys;   // groups whose recorded hit count is exact because a restricted re-query set it (host-derived tables hold supersets)
    bool strand0Plus = false, auxOk = false;
    bool partial = false, merged = false;
    int pendingContig = -1;
    // (with the candidate store an entry may wait for SEVERAL contigs: one restricted re-query each, in one launch; the further ones)
    std::vector<int> morePending;
    bool isPending(int c) const { return partial && (c == pendingContig || std::find(morePending.begin(), morePending.end(), c) != morePending.end()); }
    std::vector<std::pair<uint64_t, int>> kmerPos;   // the read's k-mers of both strands, (code, position << 1 | plus), sorted by code (built on first use)
    // a (re-)query of this entry is running on a lane: commits since its launch are examined against its dependency sets like
    // those of an entry that holds a result; `killed` = one of them invalidated it (the result is dropped when it arrives),
    // `shifts` = left extensions of contigs it meets, to be applied to the records when they arrive
    bool inflight = false, killed = false;
    std::vector<std::pair<int, int>> shifts;
    bool standing() const { return valid || partial || (inflight && !killed); }
  };
  std::vector<Cached> cache;
  size_t cacheHead = 0;
  int64_t invalidations = 0;
  // ---- live set (index not keyed by barcode): the device image is patched (t4_index_apply_delta), never rebuilt, and the
  // window slides: entries stay valid across commits that cannot change their query (rules above)
  bool live() const { return !owner && !index.considerBarcode; }
  int threads = 1;
  std::unique_ptr<HelperPool> helpers;
  int64_t nextUid = 1;
  struct KOcc { int64_t uid; int slot; unsigned char f, r; short pos; int next; };   // an occurrence of a key in a window read: the k-mer of the read as given that starts at pos (its reverse complement is the k-mer of the reverse strand at len - k - pos)
  // inverted map key -> window reads that hold it: open addressing on (code, bucket), chains of KOcc nodes; references of
  // retired reads stay (their uid no longer matches) until the map is rebuilt
  struct WinKmers {
    struct Bucket { uint64_t code; int h; int head; };
    std::vector<Bucket> tab;
    std::vector<KOcc> nodes;
    size_t used = 0;
    void clear() { tab.clear(); nodes.clear(); used = 0; }
    static size_t hashOf(uint64_t code, int h) { return (size_t)mix64h(code * 1000003ull + (uint64_t)(uint32_t)h); }
    void grow() {
      std::vector<Bucket> old; old.swap(tab);
      tab.assign(old.empty() ? 4096 : old.size() * 2, Bucket{0, 0, -2});
      for (const Bucket &b : old) if (b.head != -2) { size_t s = hashOf(b.code, b.h) & (tab.size() - 1); while (tab[s].head != -2) s = (s + 1) & (tab.size() - 1); tab[s] = b; }
    }
    void add(uint64_t code, int h, const KOcc &o) {
      if (2 * (used + 1) > tab.size()) grow();
      size_t s = hashOf(code, h) & (tab.size() - 1);
      while (tab[s].head != -2 && !(tab[s].code == code && tab[s].h == h)) s = (s + 1) & (tab.size() - 1);
      if (tab[s].head == -2) { tab[s].code = code; tab[s].h = h; tab[s].head = -1; ++used; }
      nodes.push_back(o);
      nodes.back().next = tab[s].head;
      tab[s].head = (int)nodes.size() - 1;
    }
    int find(uint64_t code, int h) const {   // first node of the key's chain, -1 when none
      if (tab.empty()) return -1;
      size_t s = hashOf(code, h) & (tab.size() - 1);
      while (tab[s].head != -2) { if (tab[s].code == code && tab[s].h == h) return tab[s].head; s = (s + 1) & (tab.size() - 1); }
      return -1;
    }
  };
  // (four shards by the key's hash: the reads a round queries for the first time are registered by up to four host threads side by
  // side, each writing its own shard -- the map was filled by one thread, 6 s of config C2 that the round's other dependency work waited for)
  static constexpr int WK_SHARDS = 4;
  WinKmers winShard[WK_SHARDS];
  static int shardOf(uint64_t code, int h) { return (int)((WinKmers::hashOf(code, h) >> 60) & (WK_SHARDS - 1)); }
  size_t winKmerRefs = 0, winKmerLive = 0;
  // the window reads that hold k-mer `code` (of a contig of barcode class h's set): fn(occurrence, onForward)
  template <class F> void forEachOcc(uint64_t code, int h, F fn) {
    { const WinKmers &wkm = winShard[shardOf(code, h)]; for (int nd = wkm.find(code, h); nd >= 0; nd = wkm.nodes[nd].next) fn(wkm.nodes[nd], true); }
    uint64_t rc = 0, x = code;
    for (int i = 0; i < k; ++i) { rc = (rc << 2) | (3ull - (x & 3ull)); x >>= 2; }
    // (the bucket of a window read's key is that of its own barcode class: a live set is not keyed by barcode, so it follows from the code)
    { const int hr = index.bucket(rc, -1); const WinKmers &wkm = winShard[shardOf(rc, hr)]; for (int nd = wkm.find(rc, hr); nd >= 0; nd = wkm.nodes[nd].next) fn(wkm.nodes[nd], false); }
  }
  std::vector<Cached *> pool;      // window entries by slot (stable while the entry lives)
  std::vector<int> freeSlots;
  std::deque<int> order;           // slots of the upcoming reads, head first
  struct IdxEv { uint64_t code; int h, idx, off, delta; uint32_t sizeAfter; };
  std::vector<IdxEv> idxEvents, evNet;
  std::vector<uint32_t> evOrder;
  std::vector<int> evSum;
  struct StructEv { int kind /* 0 region, 1 shift, 2 whole contig */, c, a, b; };
  std::vector<StructEv> structEvents;
  double runEma = 8; int64_t hitsAtLastRound = 0;   // reads served per round, recently
  int64_t baseUsed = 0;            // device arena of consensus chars / posWeight predicate bytes (one offset space)
  std::vector<int> dirtySeqs;
  bool liveReset = true;           // the next delta describes the whole image (first upload, k change)
  int maxSeqLenSeen = 0;
  int64_t toleratedStable = 0, invLongLists = 0;
  int64_t wideServed = 0, wideGroupRecords = 0, wideMispredicted = 0;
  int64_t restrictedMarks = 0, restrictedMerged = 0, restrictedFallbacks = 0, restrictedStale = 0, restrictedMulti = 0;
  bool restrictOn = true, candStore = true;
  int64_t candRecords = 0, candMerges = 0, candFallbackUncut = 0, candFallbackStats = 0, candFallbackStrand = 0, candFallbackOther = 0, candRecut = 0, candSelfChecks = 0, candMergesBig = 0, candMergesStats = 0, candExactStats = 0, candRaised = 0;
  bool mergeRestricted(Cached &c, int pc, int k2, const t4_overlap *ov, const t4_overlap *ex, const int32_t *rets, const t4_cand *nc, int ncnt, const int32_t *s8);
  static void replayScan(const std::vector<t4_cand> &cands, const std::vector<Seq> &seqs, int len, int radius, double repeatSim, std::vector<unsigned char> &cut);
  int64_t whyNot[6] = {0, 0, 0, 0, 0, 0};   // entries that fell whole although one contig changed: lists beyond 10000 postings, overlaps on the other strand, more than 44 candidate overlaps, more than ~100 groups of four hits, no report from the query, other
  // (precondition of every restricted path: what the entry's result takes from one contig is independent of the other contigs. The
  // one step of GetOverlapsFromRead that pairs hits ACROSS sequences, the VJ-junction rescue of SeqSet.hpp:1570-1575, considers
  // reference genes only and a contig set holds none; the query reports a result that came from it as nOther = 32767, which keeps
  // the entry out of here.)
  bool eligibleForRestricted(const Cached &e) {
    if (!restrictOn || !e.valid || e.skip || e.barcode != -1) { ++whyNot[5]; return false; }
    if (!e.auxOk) { ++whyNot[4]; return false; }
    if (e.fragile) { ++whyNot[0]; return false; }
    if (e.nOther != 0) { ++whyNot[1]; return false; }
    if (candStore && e.candOk) return true;   // (the replay of the scan and the certificate of the group statistics decide when the records arrive)
    if (e.nAllBound > 44) { ++whyNot[2]; return false; }
    if (e.n4 + 2 * e.restrictedCount + 2 > 100) { ++whyNot[3]; return false; }
    return true;
  }
  void rebuildGroup(Cached &e, int c);
  bool wideQueries = true; int wideHitLimit = 3072;   // what t4_add_query_pool_begin will do with a read of that many emitted hits (set in ensureLanes)
  int emittedHits(Cached &e);
  int64_t headWholeWhy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t toleranceChecks = 0, toleranceCheckKills = 0, groupSelfChecks = 0, fragileChecks = 0;
  bool exactStats(const Cached &c, int pc, const int32_t *pcSize, int T[2], int e4[2], int e5[2], int big[2], bool *ror = nullptr, int headLen = 0, bool *headTouched = nullptr);
  int64_t deltas = 0, deltaBytes = 0, rounds = 0, readsQueried = 0, invKey = 0, invCross = 0, invRegion = 0, invShift = 0, invContig = 0, invFragile = 0, tolerated = 0;
  double secDelta = 0, secGroups = 0, secEvents = 0, secRegister = 0, secPrefetch = 0, secAddTotal = 0;

  // testing / development aids, read from the environment ONCE per builder (none changes a result; DESIGN 7b lists them)
  struct Knobs {
    bool verifyWindow = false, noStableStats = false, wideQueries = true, candStore = true, restrictOn = true, predictHints = true, useMarks = true, contigKills = false, exactTolerance = true, fragileChecks = false;
    int wideHitLimit = 3072;
    int lanes = 1, queryAhead = 0, minBatch = 4, harvestDelay = 0, lightAhead = 0, maxPending = 8, restrictAhead = 0;
    FILE *roundLog = nullptr;
    Knobs() {
      auto num = [](const char *n, int d) { const char *e = getenv(n); return e ? atoi(e) : d; };
      verifyWindow = getenv("T4_VERIFY_WINDOW") != nullptr;     // every served window entry is queried again and compared
      noStableStats = getenv("T4_NO_STABLE_STATS") != nullptr;  // A/B aid: the budget rule for every entry
      lanes = num("T4_LIVE_LANES", 1); queryAhead = num("T4_QUERY_AHEAD", 0); minBatch = num("T4_LIVE_MIN_BATCH", 4); harvestDelay = num("T4_LIVE_HARVEST_DELAY", 0);
      wideQueries = !getenv("T4_WIDE_OFF") && !getenv("T4_AQ_FORCE_GLOBAL");
      { const int lim = num("T4_AQ_CAP_LIMIT", 0); wideHitLimit = lim > 0 ? lim : num("T4_WIDE_MIN_HITS", 3072); }
      fragileChecks = getenv("T4_FRAGILE_CHECKS") != nullptr;   // entries of reads with lists beyond 10000 postings take booked edits and are checked (threshold, removeOnlyRepeats, head of the hit array) instead of falling to every edit
      exactTolerance = !getenv("T4_NO_EXACT_TOLERANCE");   // A-B aid: the budget rule for every entry whose query did not certify its threshold (until round 6)
      contigKills = getenv("T4_CONTIG_KILLS") != nullptr;   // A-B aid: a merge ends every window entry with a hit on the merged contigs (until round 6)
      useMarks = !getenv("T4_NO_MARKS");        // A-B aid: restricted re-queries walk the read's posting lists as in round 4
      predictHints = !getenv("T4_NO_PREDICT");   // A-B aid: no look at the reads of the next whole-query round
      candStore = !getenv("T4_CANDS_OFF");      // testing / A-B aid: the restricted path as round 4 had it (at most 44 candidates, ~100 groups of four hits)
      restrictOn = !getenv("T4_RESTRICT_OFF");  // testing / A-B aid: every invalidated entry is queried again in full
      restrictAhead = num("T4_RESTRICT_AHEAD", 0);   // restricted re-queries only for entries within this many places of the head (0: as far as whole queries reach; -n: n/2 x the reads a round has recently served + 4)
      maxPending = num("T4_MAX_PENDING", 8);   // contigs a window entry may wait for at a time (1: round 4's rule, a second contig ends the entry)
      lightAhead = num("T4_LIGHT_AHEAD", 0);    // whole queries ride with a head that waits for a restricted re-query only when they are this near the head (0: the head's own); -1: every round carries every entry without a result (round 4)
      if (getenv("T4_ROUND_LOG")) roundLog = fopen(getenv("T4_ROUND_LOG"), "w");   // one line per launch: reads, kernel ms, per read us / overlaps / tier / killed in flight
    }
    ~Knobs() { if (roundLog) fclose(roundLog); }
  } knobs;
  t4_assembler(t4_ctx *c, int kl) : ctx(c), k(kl), index(kl) { prevAdd.readStart = -1; index.hook = this; }
  ~t4_assembler() {
    abandonJobs();
    for (Lane &L : lanes) { if (L.dev) t4_index_destroy(L.dev); if (L.ownCtx && L.ctx) t4_destroy(L.ctx); }
    for (Cached *c : pool) delete c;
  }
  void dropWindow() {
    cache.clear(); cacheHead = 0;
    abandonJobs();
    for (int s : order) { pool[s]->valid = false; pool[s]->partial = false; pool[s]->inflight = false; pool[s]->uid = 0; freeSlots.push_back(s); }
    order.clear(); for (WinKmers &w : winShard) w.clear(); winKmerRefs = winKmerLive = 0;
    idxEvents.clear(); structEvents.clear();
  }
  void invalidateSlot(int slot) { if (slot >= (int)cacheHead && slot < (int)cache.size() && cache[slot].valid) { cache[slot].valid = false; ++invalidations; } }
  void invalidateCell() { for (size_t q = cacheHead; q < cache.size(); ++q) invalidateSlot((int)q); }
  // IndexListener: cells end their window on any index change; live sets examine the change at the end of the commit
  void postMark(int idx, int off, int delta) {   // the posting marks of contig idx (see Seq::postCnt)
    if (idx < 0 || idx >= (int)seqs.size() || off < 0) return;
    Seq &q = seqs[idx];
    if ((int)q.postCnt.size() <= off) q.postCnt.resize((size_t)off + 64, 0);
    uint8_t &v = q.postCnt[(size_t)off];
    if (delta > 0) { if (v >= 3 && !q.marksBad) { q.marksBad = true; markBaseDirty(idx, 0); } if (v < 255) ++v; }
    else if (v > 0) --v;
    markBaseDirty(idx, off);
  }
  void onInsert(uint64_t code, int h, int idx, int off, uint32_t sizeAfter) override {
    if (!live()) { invalidateCell(); return; }
    postMark(idx, off, +1);
    if (!order.empty()) idxEvents.push_back(IdxEv{code, h, idx, off, +1, sizeAfter});
  }
  void onRemove(uint64_t code, int h, int idx, int off, uint32_t sizeAfter) override {
    if (!live()) { invalidateCell(); return; }
    postMark(idx, off, -1);
    if (!order.empty()) idxEvents.push_back(IdxEv{code, h, idx, off, -1, sizeAfter});
  }
  void onMove(uint64_t code, int h, int oldIdx, int oldOff, int idx, int off) override {
    if (!live()) { invalidateCell(); return; }
    postMark(oldIdx, oldOff, -1); postMark(idx, off, +1);
    if (order.empty() || oldIdx == idx) return;   // a shift of one contig's postings is one structural event (evShift)
    idxEvents.push_back(IdxEv{code, h, oldIdx, oldOff, -1, 0xFFFFFFFFu});   // sizes unchanged
    idxEvents.push_back(IdxEv{code, h, idx, off, +1, 0xFFFFFFFFu});
  }
  // ---- what a commit changed, for the window of a live set
  void evRegion(int c, int lo, int hi) { if (live() && !order.empty()) structEvents.push_back(StructEv{0, c, lo, hi}); }   // columns [lo, hi) of contig c (current coordinates)
  void evShift(int c, int shift) { if (live() && !order.empty()) structEvents.push_back(StructEv{1, c, shift, 0}); }       // bases prepended: every offset of c moves
  void evContig(int c) { if (live() && !order.empty()) structEvents.push_back(StructEv{2, c, 0, 0}); }                       // anything about c
  void processEvents();
  void markSeqDirty(int c) { if (live() && !seqs[c].devDirty) { seqs[c].devDirty = true; dirtySeqs.push_back(c); } seqs[c].dLo = 0; seqs[c].dHi = 0x7FFFFFFF; }
  void markBaseDirty(int c, int pos) {
    if (!live()) return;
    Seq &q = seqs[c];
    if (!q.devDirty) { q.devDirty = true; dirtySeqs.push_back(c); q.dLo = pos; q.dHi = pos + 1; return; }
    if (pos < q.dLo) q.dLo = pos;
    if (pos + 1 > q.dHi) q.dHi = pos + 1;
  }
  // contig c changed in a way a query can observe (consensus, length, postings, an IsBaseEqual state of a column).
  // fine == true: the caller described the change to the window itself (evRegion / evShift)
  void structuralChange(int c, bool fine = false) {
    dirty = true; patches.clear();
    if (live()) { markSeqDirty(c); if (!fine) evContig(c); }
    else invalidateCell();
  }
  // ++count[base] of one posWeight column; reports whether AlignAlgo::IsBaseEqual (AlignAlgo.hpp:49-55) can now answer differently
  void bumpWeight(int seqIdx, PosWeight &w, int base) {
    int sum = w.c[0] + w.c[1] + w.c[2] + w.c[3];
    {
      // 150 calls per committed read: the common case -- no predicate bit can flip -- decided without building the two masks. With
      // s = sum: bit x != base flips (1 -> 0) iff 3 c[x] == s + 1; the base's own bit flips (0 -> 1) iff 3 c <= s <= 3 c + 1; the
      // empty-column bit iff s == 0.
      const int s1 = sum + 1;
      bool flip = sum == 0 || (unsigned)(sum - 3 * w.c[base]) < 2u;
      for (int x = 0; x < 4; ++x) flip = flip || (x != base && 3 * w.c[x] == s1);
      if (!flip) { ++w.c[base]; return; }
    }
    unsigned before = sum == 0 ? 16u : 0u, after = 0;
    for (int x = 0; x < 4; ++x) before |= (sum < 3 * w.c[x]) ? (1u << x) : 0u;
    ++w.c[base]; ++sum;
    for (int x = 0; x < 4; ++x) after |= (sum < 3 * w.c[x]) ? (1u << x) : 0u;
    if (before == after) return;   // cannot be observed by a query: the image stays as it is
    const int pos = (int)(&w - seqs[seqIdx].pw.data());
    if (live()) { markBaseDirty(seqIdx, pos); evRegion(seqIdx, pos, pos + 1); return; }
    if (owner && !dirty && slot >= 0) {   // the resident image only needs this byte
      invalidateCell();
      patches.push_back(PwPatch{seqIdx, pos, (unsigned char)after});
      return;
    }
    structuralChange(seqIdx);
  }
  int prefetch(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive);
  int prefetchLive(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive);
  void buildGroups(Cached &e);
  void registerKmers(Cached &e, int slot, int shard = -1);   // shard >= 0: that shard's keys only, no bookkeeping (several threads, one per shard; the caller does the bookkeeping once)
  // ---- asynchronous query lanes of a live set. Every lane owns a ctx (its stream and scratch) and a REPLICA of the device image;
  // a delta is built once (makeDelta) and applied to a lane right before that lane's next launch, so a query always runs against
  // an image nothing else touches while the host goes on committing reads -- and later deltas go to the other replicas.
  struct DeltaRec {
    std::vector<int64_t> slot; std::vector<uint64_t> slotCode; std::vector<uint32_t> slotStart, slotCnt;
    std::vector<int64_t> postAt; std::vector<int32_t> postLen, postData;
    std::vector<int32_t> seqId; std::vector<t4_seq_record> seqRec;
    std::vector<int64_t> baseAt; std::vector<int32_t> baseLen; std::string baseCons; std::vector<uint8_t> basePw;
    t4_index_delta d;
    int64_t version = 0;
    void reset() {
      slot.clear(); slotCode.clear(); slotStart.clear(); slotCnt.clear(); postAt.clear(); postLen.clear(); postData.clear(); seqId.clear(); seqRec.clear();
      baseAt.clear(); baseLen.clear(); baseCons.clear(); basePw.clear(); version = 0;
    }
  };
  std::vector<std::unique_ptr<DeltaRec>> deltaSpare;
  struct Lane {
    t4_ctx *ctx = nullptr; bool ownCtx = false;
    t4_index *dev = nullptr;
    int64_t version = 0;   // deltas applied to this replica
    bool busy = false;
    int polls = 0;
    std::vector<int> slots; std::vector<int64_t> uids; std::vector<unsigned char> hint;
    std::string bases; std::vector<int64_t> offs; std::vector<int32_t> bcs, sts, only, force; std::vector<double> fac;
    int repetitive = 0;
  };
  std::vector<Lane> lanes;
  std::deque<std::unique_ptr<DeltaRec>> deltaLog;
  int64_t deltaVersion = 0;
  int64_t launches = 0, launchesUrgent = 0, headWaits = 0, killedInFlight = 0, lightRounds = 0, restrictedOnlyRounds = 0, wholeQueries = 0, restrictedQueries = 0;
  std::vector<float> roundKernelMs, roundWallMs;   // per launch (t4_assembler_chain_stats)
  std::chrono::steady_clock::time_point laneT0;
  double secHeadWait = 0, secLaunch = 0, secHarvest = 0;
  int ensureLanes();
  int flushLive(Lane **out);
  int makeDelta();
  int bringUpToDate(Lane &L);
  int launchOn(Lane &L, const std::vector<int> &todo, int repetitive);
  int harvest(Lane &L);
  void abandonJobs();
  int pumpLive(bool needHead, int repetitive);
  void announceLive(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive);
  int verifyServed(const Cached &c);
  int64_t verified = 0, verifiedMerged = 0;
  void beginWindow(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive);
  void endWindow(const int32_t *cnts, const t4_overlap *ov, const t4_overlap *ex, const int32_t *rets, int stride);
  int stageImage();   // cell mode: queue this cell's image in the owner's arena
  int releaseFinishedBarcode(int barcode, int contigMinCov);
  bool isContigShallow(int i, int minCov) const;
  void releaseShallowContigs(int minCov);

  void setPrev(int seqIdx, int rs, int re, int ss, int se, int strand) {
    prevAdd.seqIdx = seqIdx; prevAdd.readStart = rs; prevAdd.readEnd = re; prevAdd.seqStart = ss; prevAdd.seqEnd = se; prevAdd.strand = strand;
  }

  // stand-alone set whose index IS keyed by barcode (not what trust4-hip builds; kept for the C ABI): whole image per change
  int refreshDevice() {
    if (!dirty) return T4_OK;
    auto t0_ = std::chrono::steady_clock::now();
    struct Tm { double &acc; std::chrono::steady_clock::time_point t0; ~Tm() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tm_{secRefresh, t0_};
    int r;
    if (!dev) { if ((r = t4_index_create(ctx, k, index.considerBarcode ? 1 : 0, &dev))) return r; }
    if ((r = t4_index_clear(dev))) return r;
    if ((r = t4_index_set_params(dev, hitLenRequired, radius, novelSim))) return r;
    for (const Seq &s : seqs) {
      int id;
      const char *cons = s.released ? "" : s.cons.c_str();
      if ((r = t4_index_add_contig(dev, s.released ? "" : s.name.c_str(), cons, s.barcode,
                                   s.released || s.pw.empty() ? nullptr : (const int32_t *)s.pw.data(), &id))) return r;
    }
    std::vector<uint64_t> code; std::vector<int32_t> bucket, idx, off;
    code.reserve(index.total); bucket.reserve(index.total); idx.reserve(index.total); off.reserve(index.total);
    for (const auto &kv : index.map)
      for (uint32_t t = 0; t < kv.second.cnt; ++t) {
        const Post &p = index.arena[kv.second.start + t];
        code.push_back(kv.first.code); bucket.push_back(kv.first.h); idx.push_back(p.idx); off.push_back(p.offset);
      }
    if ((r = t4_index_commit_postings(dev, (int64_t)code.size(), code.data(), bucket.data(), idx.data(), off.data()))) return r;
    dirty = false; ++refreshes;
    return T4_OK;
  }

  // SeqSet::InputNovelRead (SeqSet.hpp:3028-3073)
  int inputNovelRead(const char *id, const char *read, int strand, int barcode) {
    Seq ns;
    ns.name = id; ns.cons = read;
    if (strand == -1) reverseComplement(ns.cons, std::string(read));
    ns.barcode = barcode; ns.numRead = 1;
    int len = (int)ns.cons.size();
    ns.pw.assign(len, PosWeight{{0, 0, 0, 0}});
    for (int i = 0; i < len; ++i) if (ns.cons[i] != 'N') ns.pw[i].c[nucNum(ns.cons[i])] = 1;
    int seqIdx = (int)seqs.size();
    seqs.push_back(ns);
    index.build(seqs[seqIdx].cons.c_str(), len, seqIdx, barcode);
    setPrev(seqIdx, 0, len - 1, 0, len - 1, strand);
    structuralChange(seqIdx, true);   // a new contig reaches the window through its index insertions only
    return seqIdx;
  }

  // UpdateConsensus (SeqSet.hpp:4537-4588)
  void updateConsensus(int seqIdx, bool updateIndex) {
    Seq &s = seqs[seqIdx];
    if (s.frozen) return;   // posWeightCompressed (SeqSet.hpp:4542-4543)
    std::vector<std::pair<int, int>> changes;
    for (int i = 0; i < (int)s.cons.size(); ++i) {
      int mx = 0, tag = 0;
      for (int j = 0; j < 4; ++j) if (s.pw[i].c[j] > mx) { mx = s.pw[i].c[j]; tag = j; }
      if (mx == 0) continue;
      int cur = s.cons[i] == 'N' ? 0 : nucNum(s.cons[i]);   // nucToNum['N' - 'A'] is 0 in the reference's table (main.cpp:39-44)
      if (cur != tag && s.pw[i].c[cur] < mx) changes.push_back({i, tag});
    }
    if (changes.empty()) return;
    if (updateIndex) index.removeSeq(s.cons.c_str(), (int)s.cons.size(), seqIdx, s.barcode, 0);
    for (auto &c : changes) { s.cons[c.first] = NUM2NUC[c.second]; evRegion(seqIdx, c.first, c.first + 1); }
    if (updateIndex) index.build(s.cons.c_str(), (int)s.cons.size(), seqIdx, s.barcode, 0);
    structuralChange(seqIdx, true);
  }
  // UpdateAllConsensus (SeqSet.hpp:4525-4535). The driver calls it every 10 000 assembled reads (main.cpp:1862-1868); a contig whose
  // counts and consensus are what they were at the last call comes out of UpdateConsensus unchanged (the call before settled every
  // column), so only the contigs a read has touched since are walked -- all 23 k of them 170 times was 3-4 s of config C2's chain.
  void updateAllConsensus() {
    for (int i = 0; i < (int)seqs.size(); ++i) if (!seqs[i].released && seqs[i].pwTouched) { updateConsensus(i, true); seqs[i].pwTouched = false; }
  }

  // SubstituteConsensusPos (SeqSet.hpp:11058-11080), updateIndex == true
  void substituteConsensusPos(int seqIdx, int pos, char c) {
    Seq &s = seqs[seqIdx];
    int clen = (int)s.cons.size();
    if (pos >= clen || s.cons[pos] == c) return;
    int start = pos - k + 1, end = pos + k - 1;
    if (start < 0) start = 0;
    if (end >= clen) end = clen - 1;
    index.removeSeq(s.cons.c_str() + start, end - start + 1, seqIdx, s.barcode, start);
    s.cons[pos] = c;
    index.build(s.cons.c_str() + start, end - start + 1, seqIdx, s.barcode, start);
    evRegion(seqIdx, pos, pos + 1);
    structuralChange(seqIdx, true);
  }

  // RepeatAddRead (SeqSet.hpp:4477-4507)
  int repeatAddRead(const char *read) {
    if (prevAdd.seqIdx < 0) return prevAdd.seqIdx;
    std::string r = read;
    if (prevAdd.strand == -1) reverseComplement(r, std::string(read));
    Seq &s = seqs[prevAdd.seqIdx];
    s.pwTouched = true;
    for (int i = prevAdd.readStart; i <= prevAdd.readEnd; ++i) {
      if (r[i] == 'N') continue;
      bumpWeight(prevAdd.seqIdx, s.pw[i + prevAdd.seqStart], nucNum(r[i]));
    }
    ++s.numRead;
    return prevAdd.seqIdx;
  }

  // SeqSet::ChangeKmerLength -> Clean(false) (SeqSet.hpp:4591-4629): drop released slots (contig ids are renumbered),
  // rebuild the whole index with the new k
  int changeKmerLength(int kl) {
    k = kl;
    bool cb = index.considerBarcode;
    dropWindow();
    index = HostIndex(kl);
    index.considerBarcode = cb;
    index.hook = this;
    index.mirror = live();
    std::vector<Seq> kept;
    for (Seq &s : seqs) if (!s.released) kept.push_back(std::move(s));
    seqs.swap(kept);
    for (Seq &s : seqs) { s.postCnt.clear(); s.marksBad = false; }   // (the builds below mark every posting of the new index)
    for (int i = 0; i < (int)seqs.size(); ++i) index.build(seqs[i].cons.c_str(), (int)seqs[i].cons.size(), i, seqs[i].barcode, 0);
    setPrev(-1, -1, -1, -1, -1, 0);
    if (dev) { t4_index_destroy(dev); dev = nullptr; }   // nomatchGapLimit and the lookup layout depend on k
    if (live()) {   // a fresh image: every contig gets a new place in the arena
      for (Lane &L : lanes) { if (L.dev) { t4_index_destroy(L.dev); L.dev = nullptr; } L.version = deltaVersion; }   // (no job is in flight: dropWindow above)
      deltaLog.clear();
      liveReset = true; baseUsed = 0; dirtySeqs.clear(); maxSeqLenSeen = 0;
      for (int i = 0; i < (int)seqs.size(); ++i) { seqs[i].baseOff = -1; seqs[i].baseCap = 0; seqs[i].devDirty = false; markSeqDirty(i); }
    }
    dirty = true;
    return T4_OK;
  }
  int addRead(const char *read, const char *geneName, int *strandIO, int barcode, int minKmerCount, bool repetitiveData, double similarityThreshold);
  int output(const char *path) const;
};

// SeqSet::AddRead (SeqSet.hpp:3426-4473) for a set that holds novel contigs only (stage 1 keeps the reference genes
// in a separate SeqSet, main.cpp:642, so the isRef branches of the reference are unreachable there).
int t4_assembler::addRead(const char *readC, const char *geneName, int *strandIO, int barcode, int minKmerCount,
                          bool repetitiveData, double similarityThreshold) {
  const std::string read = readC;
  const int len = (int)read.size();
  const int K = this->k;   // k-mer length
  setPrev(-1, -1, -1, -1, -1, 0);
  // GetOverlapsFromRead + the ExtendOverlap of every overlap: from the speculation window when it is still valid,
  // else from a fresh GPU query of this read
  std::vector<t4_overlap> ovBuf, extBuf;
  std::vector<int32_t> extRet;
  int32_t cnt = 0;
  if (live()) {
    ts.lap(TS_ADD_SERVE);
    processEvents();   // a release_* call or UpdateAllConsensus may have changed the set since the last commit was examined
    auto lines = [&](const Cached &c) { return c.read == read && c.strand == *strandIO && c.barcode == barcode && c.skip == (repetitiveData ? 1 : 0); };
    if (!order.empty() && lines(*pool[order.front()])) {
      if (pool[order.front()]->valid) ++cacheHits;
      { const int rc = pumpLive(true, repetitiveData ? 1 : 0); if (rc) return -100 + rc; }   // background re-queries; waits for the head when it has no result yet
    } else {
      const char *one = read.c_str();
      int st = *strandIO, bcOne = barcode;
      int rc = prefetchLive(1, &one, &st, &bcOne, repetitiveData ? 1 : 0);
      if (rc) return -100 + rc;
    }
    const int sl = order.front();
    Cached &c = *pool[sl];
    if (knobs.verifyWindow) { const int rc = verifyServed(c); if (rc) return -100 + rc; }
    cnt = c.cnt; ovBuf.swap(c.ov); extBuf.swap(c.ext); extRet.swap(c.extRet);
    order.pop_front();
    c.valid = false; c.partial = false; c.uid = 0; freeSlots.push_back(sl);   // its winKmers references are stale from here on
    if (c.registered) { --winKmerLive; c.registered = false; }
    ts.lap(TS_ADD_SERVE);
  } else {
    bool served = false;
    if (cacheHead < cache.size()) {
      Cached &c = cache[cacheHead];
      if (c.valid && c.read == read && c.strand == *strandIO && c.barcode == barcode && c.skip == (repetitiveData ? 1 : 0)) {
        cnt = c.cnt; ovBuf.swap(c.ov); extBuf.swap(c.ext); extRet.swap(c.extRet);
        ++cacheHead; ++cacheHits; served = true;
      }
    }
    if (!served) {
      dropWindow();
      const char *one = read.c_str();
      int st = *strandIO, bcOne = barcode;
      int rc = prefetch(1, &one, &st, &bcOne, repetitiveData ? 1 : 0);
      if (rc) return -100 + rc;
      Cached &c = cache[0];
      cnt = c.cnt; ovBuf.swap(c.ov); extBuf.swap(c.ext); extRet.swap(c.extRet);
      cacheHead = 1;
    }
  }
  int overlapCnt = cnt;
  if (overlapCnt <= 0) return -1;

  // (this runs once per served read, between two query rounds: the candidate records are filtered in place and ordered through an
  // index array, and the working arrays keep their storage from call to call)
  struct Cand { Ov ov; Ov ext; int extRet; };
  static thread_local std::vector<Cand> candStore_, cands;
  static thread_local std::vector<int> ordIdx;
  candStore_.clear();
  candStore_.reserve((size_t)overlapCnt);
  for (int i = 0; i < overlapCnt; ++i) {
    if (geneName[0] != '\0') {   // SeqSet.hpp:3462-3473: contigs of another gene family are left out
      const std::string &nm = seqs[ovBuf[i].seqIdx].name;
      int j = 3;
      if (!nm.empty() && nm[0] >= 'A' && nm[0] <= 'Z') {
        for (j = 0; j < 3; ++j) if ((j < (int)nm.size() ? nm[j] : '\0') != geneName[j]) break;
      }
      if (!(j == 3 || nm == "Novel")) continue;
    }
    candStore_.push_back(Cand{fromT4(ovBuf[i]), fromT4(extBuf[i]), extRet[i]});
  }
  overlapCnt = (int)candStore_.size();
  if (overlapCnt <= 0) return -1;
  ordIdx.resize((size_t)overlapCnt);
  for (int i = 0; i < overlapCnt; ++i) ordIdx[(size_t)i] = i;
  // (the order of std::stable_sort by _overlap::operator<, without its temporary buffer per call: ties go by position)
  if (overlapCnt > 1)
    std::sort(ordIdx.begin(), ordIdx.end(), [&](int a, int b) {
      if (ovLess(candStore_[(size_t)a].ov, candStore_[(size_t)b].ov)) return true;
      if (ovLess(candStore_[(size_t)b].ov, candStore_[(size_t)a].ov)) return false;
      return a < b;
    });
  cands.resize((size_t)overlapCnt);
  for (int i = 0; i < overlapCnt; ++i) cands[(size_t)i] = candStore_[(size_t)ordIdx[(size_t)i]];
  static thread_local std::vector<Ov> overlaps, ext, failed;
  static thread_local std::vector<std::pair<int, int>> oldMinExtAnchor;
  overlaps.resize((size_t)overlapCnt);
  for (int i = 0; i < overlapCnt; ++i) overlaps[i] = cands[i].ov;
  ext.assign((size_t)overlapCnt, Ov()); failed.assign((size_t)overlapCnt, Ov());
  oldMinExtAnchor.assign((size_t)overlapCnt, std::pair<int, int>(0, 0));
  int ne = 0, ret = -1, failedCnt = 0, tag = 0, jMerge = 0;
  bool sortExtended = true;
  Ov good;
  std::string rcRead;
  reverseComplement(rcRead, read);
  const std::string &r = overlaps[0].strand == 1 ? read : rcRead;
  int readInConsensusOffset = 0, seqIdx = -1;
  bool addNew = true;
  int i, j;
  ts.lap(TS_ADD_CANDS);

  auto clen = [&](int s) { return (int)seqs[s].cons.size(); };
  for (i = 0; i < overlapCnt; ++i) {
    oldMinExtAnchor[i] = {seqs[overlaps[i].seqIdx].minLeftExtAnchor, seqs[overlaps[i].seqIdx].minRightExtAnchor};
    for (j = 0; j < ne; ++j) {
      int leftRadius = radius, rightRadius = radius;
      if (ext[j].seqStart == 0) leftRadius = 0;
      if (ext[j].seqEnd == clen(ext[j].seqIdx) - 1) rightRadius = 0;
      if (overlaps[i].readStart >= ext[j].readStart - leftRadius && overlaps[i].readEnd <= ext[j].readEnd + rightRadius &&
          (overlaps[i].seqStart >= radius || overlaps[i].seqEnd <= clen(overlaps[i].seqIdx) - radius - 1)) break;
      leftRadius = radius; rightRadius = radius;
      if (overlaps[i].seqStart == 0) leftRadius = 0;
      if (overlaps[i].seqEnd == clen(overlaps[i].seqIdx) - 1) rightRadius = 0;
      if (ext[j].readStart >= overlaps[i].readStart - leftRadius && ext[j].readEnd <= overlaps[i].readEnd + rightRadius) break;
    }
    if (j < ne) continue;
    ext[ne] = cands[i].ext;                       // what ExtendOverlap leaves in extendedOverlaps[ne]
    if (cands[i].extRet == 1) {
      Seq &es = seqs[ext[ne].seqIdx];
      if (ext[ne].similarity < similarityThreshold) {
        if ((minKmerCount <= 1 || ext[ne].similarity + 0.01 >= similarityThreshold) && ext[ne].readStart == 0 && ext[ne].readEnd == len - 1) good = ext[ne];
        continue;
      }
      for (j = 0; j < ne; ++j) {
        int leftRadius = radius, rightRadius = radius;
        if (ext[j].seqStart == 0) leftRadius = 0;
        if (ext[j].seqEnd == clen(ext[j].seqIdx) - 1) rightRadius = 0;
        if (ext[ne].readStart >= ext[j].readStart - leftRadius && ext[ne].readEnd <= ext[j].readEnd + rightRadius &&
            (overlaps[i].seqStart > 0 || overlaps[i].seqEnd < clen(overlaps[i].seqIdx) - 1)) break;
        if (ext[j].readStart >= ext[ne].readStart - radius && ext[j].readEnd <= ext[ne].readEnd + radius) break;
      }
      if (j < ne) continue;
      const int span = ext[ne].readEnd - ext[ne].readStart + 1;
      auto raiseAnchors = [&]() {
        if (ext[ne].readStart > 0 && es.minLeftExtAnchor < span) es.minLeftExtAnchor = span;
        if (ext[ne].readEnd < len - 1 && es.minRightExtAnchor < span) es.minRightExtAnchor = span;
      };
      for (j = 0; j < i; ++j) {
        if (ext[ne].seqStart == 0 && ext[ne].seqEnd == clen(ext[ne].seqIdx) - 1) continue;
        if (ext[ne].readStart >= overlaps[j].readStart && ext[ne].readEnd <= overlaps[j].readEnd &&
            (overlaps[j].readEnd - overlaps[j].readStart >= ext[ne].readEnd - ext[ne].readStart + 10 ||
             overlaps[j].similarity + 0.02 >= ext[ne].similarity)) { raiseAnchors(); break; }
      }
      if (j < i) continue;
      for (j = 0; j < failedCnt; ++j) {
        if (ext[ne].seqStart == 0 && ext[ne].seqEnd == clen(ext[ne].seqIdx) - 1) continue;
        if (ext[ne].readStart >= failed[j].readStart && ext[ne].readEnd <= failed[j].readEnd) { raiseAnchors(); break; }
      }
      if (j < failedCnt) continue;
      if (ext[ne].readStart > 0 && es.minLeftExtAnchor >= span) continue;
      if (ext[ne].readEnd < len - 1 && es.minRightExtAnchor >= span) continue;
      tag = i;
      ++ne;
    } else failed[failedCnt++] = ext[ne];
  }

  if (ne == 1 && ext[0].readStart <= radius && ext[0].readEnd >= len - radius) {
    // could the read bridge two contigs?
    for (i = 0; i < overlapCnt; ++i) {
      if (tag == i) continue;
      ext[ne] = cands[i].ext;
      if (cands[i].extRet == 1) { jMerge = i; ++ne; }
    }
    if (ne > 2) ne = 1;
    else if (ne == 2) {
      const int span1 = ext[1].readEnd - ext[1].readStart + 1;
      if (ext[1].readStart > 0 && oldMinExtAnchor[jMerge].first >= span1) ne = 1;
      if (ext[1].readEnd < len - 1 && oldMinExtAnchor[jMerge].second >= span1) ne = 1;
      if (ne == 2) {
        if (ext[0].seqEnd == clen(ext[0].seqIdx) - 1 && ext[1].seqStart == 0) sortExtended = false;
        else if (ext[0].seqStart == 0 && ext[1].seqEnd == clen(ext[1].seqIdx) - 1) { sortExtended = false; std::swap(ext[0], ext[1]); }
        else ne = 1;
      }
    }
  }
  if (similarityThreshold > novelSim) {
    int c = 0;
    for (i = 0; i < ne; ++i) if (ext[i].similarity >= similarityThreshold) ext[c++] = ext[i];
    ne = c;
  }
  if (ne == 0 && good.seqIdx != -1) { ext[0] = good; ne = 1; }
  if (ne > 1) {
    for (i = 0; i < ne; ++i) if (ext[i].similarity >= 0.95) break;
    if (i >= ne) {
      int maxtag = 0;
      for (i = 1; i < ne; ++i) if (ovLess(ext[i], ext[maxtag])) maxtag = i;
      ext[0] = ext[maxtag];
      ne = 1;
    }
  }
  if (ne > 1) {
    bool dup = false;
    for (i = 0; i < ne - 1 && !dup; ++i) for (j = i + 1; j < ne; ++j) if (ext[i].seqIdx == ext[j].seqIdx) { dup = true; break; }
    if (dup) ne = 0;
  }

  ts.lap(TS_ADD_DECIDE);
  if (ne > 1) {
    // ---- merge contigs through the read (SeqSet.hpp:3878-4130)
    const int eCnt = ne;
    addNew = false;
    if (sortExtended) std::stable_sort(ext.begin(), ext.begin() + eCnt, [](const Ov &a, const Ov &b) { return a.readStart < b.readStart; });
    for (i = 0; i < eCnt; ++i) for (j = i + 1; j < eCnt; ++j) if (!nameCompatible(seqs[ext[i].seqIdx].name, seqs[ext[j].seqIdx].name)) return -1;
    int sum = 0;
    for (i = 0; i < eCnt; ++i) sum += clen(ext[i].seqIdx);
    std::string newCons(sum + len + 1, '\0');
    std::vector<int> seqOffset(eCnt);
    if (ext[0].readStart > 0) { for (i = 0; i < eCnt; ++i) seqOffset[i] = ext[i].readStart; }
    else {
      seqOffset[0] = 0;
      for (i = 1; i < eCnt; ++i) seqOffset[i] = seqOffset[i - 1] + clen(ext[i - 1].seqIdx) - 1 + (ext[i].readStart - ext[i - 1].readEnd);
    }
    if (ext[0].readStart > 0) memcpy(&newCons[0], r.data(), len); else memcpy(&newCons[ext[0].seqStart], r.data(), len);
    for (i = eCnt - 1; i >= 0; --i) memcpy(&newCons[seqOffset[i]], seqs[ext[i].seqIdx].cons.data(), clen(ext[i].seqIdx));
    int newLen = 0, lastEnd = eCnt - 1, kk = 0;
    for (i = 0; i < eCnt; ++i) if (seqOffset[i] + clen(ext[i].seqIdx) > kk) { kk = seqOffset[i] + clen(ext[i].seqIdx); lastEnd = i; }
    if (ext[lastEnd].readEnd < len) newLen = kk + (len - ext[lastEnd].readEnd - 1); else newLen = kk;
    newCons.resize(newLen);
    int newSeqIdx = ext[0].seqIdx, kmin = 0;
    for (i = 1; i < eCnt; ++i) if (ext[i].seqIdx < newSeqIdx) { newSeqIdx = ext[i].seqIdx; kmin = i; }
    {
      std::vector<PosWeight> &pw = seqs[newSeqIdx].pw;
      int oldLen = clen(newSeqIdx);
      std::vector<PosWeight> npw(newLen, PosWeight{{0, 0, 0, 0}});
      for (int t = 0; t < oldLen && seqOffset[kmin] + t < newLen; ++t) npw[seqOffset[kmin] + t] = pw[t];
      pw.swap(npw);
    }
    for (i = 0; i < eCnt; ++i) {
      int sIdx = ext[i].seqIdx;
      if (sIdx == newSeqIdx) continue;
      seqs[newSeqIdx].numRead += seqs[sIdx].numRead;
      for (j = 0; j < clen(sIdx); ++j) for (int c = 0; c < 4; ++c) seqs[newSeqIdx].pw[seqOffset[i] + j].c[c] += seqs[sIdx].pw[j].c[c];
    }
    for (i = 0; i < eCnt; ++i) index.removeSeq(seqs[ext[i].seqIdx].cons.c_str(), clen(ext[i].seqIdx), ext[i].seqIdx, barcode, 0);
    {
      int nameIdx;
      for (nameIdx = 0; nameIdx < eCnt; ++nameIdx) if (seqs[ext[nameIdx].seqIdx].name != "Novel") break;
      if (nameIdx >= eCnt) nameIdx = 0;
      std::string nb = seqs[ext[nameIdx].seqIdx].name;
      for (i = 0; i < eCnt; ++i) {
        if (i == nameIdx) continue;
        if (i > 0 && seqs[ext[i].seqIdx].name != seqs[ext[i - 1].seqIdx].name) nb += "+" + seqs[ext[i].seqIdx].name;
      }
      // the reference frees/overwrites name strings in place; names of merged-away seqs are read before release
      seqs[newSeqIdx].name = nb;
    }
    // anchors are read from the (pre-merge) owners before they are released
    const int newMinLeft = seqs[ext[0].seqIdx].minLeftExtAnchor, newMinRight = seqs[ext[lastEnd].seqIdx].minRightExtAnchor;
    for (i = 0; i < eCnt; ++i) {
      int sIdx = ext[i].seqIdx;
      if (sIdx == newSeqIdx) continue;
      Seq &d = seqs[sIdx];
      d.released = true; d.name.clear(); d.cons.clear(); d.pw.clear();
    }
    seqs[newSeqIdx].cons = newCons;
    updateConsensus(newSeqIdx, false);
    index.build(seqs[newSeqIdx].cons.c_str(), newLen, newSeqIdx, barcode);
    seqs[newSeqIdx].minLeftExtAnchor = newMinLeft;
    seqs[newSeqIdx].minRightExtAnchor = newMinRight;
    readInConsensusOffset = ext[0].seqStart > 0 ? ext[0].seqStart : 0;
    seqIdx = newSeqIdx;
    for (i = 0; i < eCnt; ++i) structuralChange(ext[i].seqIdx);
    ts.lap(TS_ADD_MERGE);
  } else if (ne == 1) {
    // ---- extend one contig, or place the read inside it (SeqSet.hpp:4131-4316)
    addNew = false;
    seqIdx = ext[0].seqIdx;
    Seq &seq = seqs[seqIdx];
    ++seq.numRead;
    if (ext[0].readStart > 0 || ext[0].readEnd < len - 1) {
      std::vector<std::pair<int, int>> replacement;
      const int oldLen = clen(seqIdx);
      std::string newCons;
      if (ext[0].readStart > 0) newCons.assign(r.data(), ext[0].readStart);
      newCons += seq.cons;
      if (ext[0].readEnd < len - 1) newCons.append(r.data() + ext[0].readEnd + 1, len - 1 - ext[0].readEnd);
      const int newLen = (int)newCons.size();
      const int shift = ext[0].readStart;
      ts.lap(TS_EXT_STRINGS);
      if (shift > 0) {
        index.build(newCons.c_str(), ext[0].readStart + K - 1, seqIdx, barcode);
        ts.lap(TS_EXT_BUILD);
        index.update(seq.cons.c_str(), oldLen, barcode, shift, seqIdx, seqIdx);
        ts.lap(TS_EXT_UPDATE);
      }
      if (ext[0].readEnd < len - 1) {
        int start = ext[0].readStart + ext[0].seqEnd - K + 2;
        index.build(newCons.c_str() + start, newLen - start, seqIdx, barcode, start);
        ts.lap(TS_EXT_BUILD);
      }
      const int expandSize = ext[0].readStart + (len - 1 - ext[0].readEnd);
      seq.pw.resize(oldLen + expandSize, PosWeight{{0, 0, 0, 0}});
      if (shift > 0) {
        for (i = oldLen - 1; i >= 0; --i) seq.pw[i + shift] = seq.pw[i];
        if (barcode == -1 || minKmerCount > 1) {
          for (i = 0; i < 2; ++i) {
            if (i + shift >= len || r[i + shift] == 'N') continue;
            if (r[i + shift] != newCons[i + shift] && newCons[i + shift] != 'N' && seq.pw[i + shift].c[nucNum(newCons[i + shift])] == 1)
              replacement.push_back({i + shift, (int)r[i + shift]});
            for (j = 0; j < 4; ++j) if (r[i + shift] != NUM2NUC[j] && seq.pw[i + shift].c[j] > 1) --seq.pw[i + shift].c[j];
          }
        }
        for (i = 0; i < shift; ++i) seq.pw[i] = PosWeight{{0, 0, 0, 0}};
      }
      if (ext[0].readEnd < len - 1) {
        int start = ext[0].readStart + oldLen;
        for (i = 0; i < len - ext[0].readEnd - 1; ++i) seq.pw[start + i] = PosWeight{{0, 0, 0, 0}};
        if (barcode == -1 || minKmerCount > 1) {
          for (i = oldLen - 2; i < oldLen; ++i) {
            int pos = i - ext[0].seqStart, seqPos = i + shift;
            if (pos < 0 || r[pos] == 'N') continue;
            if (r[pos] != newCons[seqPos] && newCons[seqPos] != 'N' && seq.pw[seqPos].c[nucNum(newCons[seqPos])] == 1)
              replacement.push_back({seqPos, (int)r[pos]});
            for (j = 0; j < 4; ++j) if (r[pos] != NUM2NUC[j] && seq.pw[seqPos].c[j] > 1) --seq.pw[seqPos].c[j];
          }
        }
      }
      if (shift > 0) seq.minLeftExtAnchor = 0;
      if (ext[0].readEnd < len - 1) seq.minRightExtAnchor = 0;
      readInConsensusOffset = ext[0].seqStart > 0 ? ext[0].seqStart : 0;
      seq.cons = newCons;
      // what the window sees: bases prepended (every offset moves; the two lowered end columns are inside the new reach of
      // the left end), bases appended from two columns before the old end
      if (shift > 0) evShift(seqIdx, shift);
      if (ext[0].readEnd < len - 1) evRegion(seqIdx, shift + oldLen - 2, 0x7FFFFFFF);
      structuralChange(seqIdx, true);
      ts.lap(TS_EXT_PW);
      for (auto &p : replacement) substituteConsensusPos(seqIdx, p.first, (char)p.second);
      ts.lap(TS_EXT_SUBST);
    } else readInConsensusOffset = ext[0].seqStart;
    ts.lap(TS_ADD_EXTEND);
  }

  if (!addNew) {
    Seq &seq = seqs[seqIdx];
    std::vector<int> nPos;
    for (i = 0; i < len; ++i) {
      if (r[i] == 'N') continue;
      bumpWeight(seqIdx, seq.pw[i + readInConsensusOffset], nucNum(r[i]));
      if (seq.cons[i + readInConsensusOffset] == 'N') nPos.push_back(i);
    }
    setPrev(seqIdx, 0, len - 1, readInConsensusOffset, readInConsensusOffset + len - 1, overlaps[0].strand);
    int size = (int)nPos.size();
    for (i = 0; i < size;) {
      for (j = i + 1; j < size; ++j) if (nPos[j] > nPos[j - 1] + K - 1) break;
      for (int l = i; l < j; ++l) seq.cons[nPos[l] + readInConsensusOffset] = r[nPos[l]];
      int start = nPos[i] - K + 1 + readInConsensusOffset;
      if (start < 0) start = 0;
      int end = nPos[j - 1] + K - 1 + readInConsensusOffset;
      if (end >= (int)seq.cons.size()) end = (int)seq.cons.size() - 1;
      index.build(seq.cons.c_str() + start, end - start + 1, seqIdx, barcode, start);
      evRegion(seqIdx, start, end + 1);
      structuralChange(seqIdx, true);
      i = j;
    }
    ret = seqIdx;
    seq.pwTouched = true;   // (extension / merge / placement: counts or consensus of this contig changed)
    ts.lap(TS_ADD_BUMP);
  }
  // a set of novel contigs has no reference sequence to anchor a new contig on (SeqSet.hpp:4373-4384)
  if (ret == -1) { setPrev(-2, -1, -1, -1, -1, 0); ret = -2; }
  if (ret >= 0 && *strandIO == 0) *strandIO = overlaps[0].strand;
  return ret;
}

// ---- per-barcode sets -------------------------------------------------------------------------------
// With barcodes the index key carries the barcode and hits are filtered to the read's barcode (main.cpp:1556-1559,
// SeqSet.hpp:1418): the contigs of different cells never interact, a cell is an independent SeqSet. t4_cellset keeps one
// t4_assembler per barcode (contig ids local to the cell) and serves the AddRead queries of many cells in one launch.
struct t4_cellset {
  t4_ctx *ctx = nullptr;
  t4_cellstore *store = nullptr;
  int k = 9, hitLenRequired = 31, radius = 10;
  double novelSim = 0.9;
  std::map<int, t4_assembler *> cells;   // by barcode id == the reference's processing order of the cells
  int64_t queries = 0, readsQueried = 0, batchSeq = 0;
  std::atomic<int64_t> stagedImages{0};
  double secQuery = 0, secStage = 0;
  int threads = 1;   // host threads for image builds and window bookkeeping (cells are independent)
  std::unique_ptr<t4_overlap[]> resOv, resEx;
  std::unique_ptr<int32_t[]> resRet;
  size_t resCap = 0;
  std::string err;
};

int t4_assembler::stageImage() {   // thread-safe across cells once the owner has run t4_cellstore_prepare
  if (!dirty) return T4_OK;
  int r;
  if (slot < 0 || !owner) return T4_ERR_STATE;
  const int n = (int)seqs.size();
  std::vector<const char *> names(n), cons(n);
  std::vector<const int32_t *> pw(n);
  for (int i = 0; i < n; ++i) {
    const Seq &q = seqs[i];
    const bool gone = q.released;
    names[i] = gone ? "" : q.name.c_str();
    cons[i] = gone ? "" : q.cons.c_str();
    pw[i] = (gone || q.pw.empty()) ? nullptr : (const int32_t *)q.pw.data();
  }
  // one list per (code, bucket) key, postings in the replica's order; the order of the keys does not matter to a hash table
  static thread_local std::vector<uint64_t> keyCode; static thread_local std::vector<int32_t> keyBucket, keyCnt, post;
  keyCode.clear(); keyBucket.clear(); keyCnt.clear(); post.clear();
  for (const auto &kv : index.map) {
    keyCode.push_back(kv.first.code); keyBucket.push_back(kv.first.h); keyCnt.push_back((int32_t)kv.second.cnt);
    for (uint32_t t = 0; t < kv.second.cnt; ++t) { const Post &p = index.arena[kv.second.start + t]; post.push_back(p.idx); post.push_back(p.offset); }
  }
  int64_t oPw = 0;
  r = t4_cellstore_stage(owner->store, slot, cellBarcode, n, names.data(), cons.data(), pw.data(), (int64_t)keyCode.size(), keyCode.data(),
                         keyBucket.data(), keyCnt.data(), post.data(), &oPw, nullptr);
  if (r) return r;
  imgPwOff.resize(n);
  for (int i = 0; i < n; ++i) { imgPwOff[i] = oPw; oPw += (int64_t)strlen(cons[i]) + 1; }
  patches.clear();
  dirty = false; ++refreshes;
  ++owner->stagedImages;
  return T4_OK;
}

// ---- live set: device image by deltas, sliding speculation window ---------------------------------------------------

// Bring the device image up to date: everything that changed since the last call, each destination once with its final value.
// What changed in the set since the last delta, described by position in the image's four arrays (the host replica stores its lists
// in the device layout). Built once; every lane applies it to its own replica before its next launch (bringUpToDate).
int t4_assembler::makeDelta() {
  auto t0_ = std::chrono::steady_clock::now();
  struct Tm { double &acc; std::chrono::steady_clock::time_point t0; ~Tm() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tm_{secDelta, t0_};
  if (!liveReset && !index.tabRebuilt && index.dirtyKeys.empty() && index.dirtyPost.empty() && dirtySeqs.empty()) return T4_OK;   // the replicas are current
  if (liveReset) {   // everything is new to the device
    index.tabRebuilt = true;
    index.dirtyPost.clear();
    for (uint32_t at = 0; at < (uint32_t)index.arenaUsed; ++at) index.dirtyPost.push_back(at);
    for (int i = 0; i < (int)seqs.size(); ++i) markSeqDirty(i);
  }
  std::unique_ptr<DeltaRec> recp;
  if (!deltaSpare.empty()) { recp = std::move(deltaSpare.back()); deltaSpare.pop_back(); recp->reset(); }   // (a dozen vectors allocated afresh per round otherwise)
  else recp.reset(new DeltaRec());
  DeltaRec &R = *recp;
  t4_index_delta &d = R.d;
  memset(&d, 0, sizeof d);
  // table slots
  auto putKey = [&](const HostIndex::Map::value_type &kv) {
    R.slot.push_back(kv.second.slot); R.slotCode.push_back(kv.first.code); R.slotStart.push_back(kv.second.start); R.slotCnt.push_back(kv.second.cnt);
  };
  if (index.tabRebuilt) { for (auto &kv : index.map) { putKey(kv); kv.second.dirty = false; } }
  else for (auto *kv : index.dirtyKeys) { putKey(*kv); kv->second.dirty = false; }
  index.dirtyKeys.clear();
  // postings: runs of consecutive dirty positions
  std::sort(index.dirtyPost.begin(), index.dirtyPost.end());
  for (size_t i = 0; i < index.dirtyPost.size();) {
    size_t j = i + 1;
    while (j < index.dirtyPost.size() && index.dirtyPost[j] == index.dirtyPost[j - 1] + 1) ++j;
    R.postAt.push_back(index.dirtyPost[i]); R.postLen.push_back((int32_t)(j - i));
    for (size_t t = i; t < j; ++t) { const Post &p = index.arena[index.dirtyPost[t]]; R.postData.push_back(p.idx); R.postData.push_back(p.offset); if (index.dirtyPost[t] < index.dirtyPostFlag.size()) index.dirtyPostFlag[index.dirtyPost[t]] = 0; }
    i = j;
  }
  index.dirtyPost.clear();
  // sequences: record + the changed stretch of consensus chars / predicate bytes (one terminator column past the end)
  int maxLen = 0;
  for (int c : dirtySeqs) {
    Seq &q = seqs[c];
    q.devDirty = false;
    const int len = q.released ? 0 : (int)q.cons.size();
    if (len > maxLen) maxLen = len;
    int lo = q.dLo, hi = q.dHi;
    if (len + 1 > q.baseCap) {   // a new place with room to grow
      q.baseCap = len + 1 + (len < 512 ? 256 : len / 2);
      q.baseOff = baseUsed; baseUsed += q.baseCap;
      lo = 0; hi = len + 1;
    }
    if (hi > len + 1) hi = len + 1;
    if (lo < 0) lo = 0;
    t4_seq_record rec;
    memset(&rec, 0, sizeof rec);
    rec.base_off = q.baseOff; rec.len = len; rec.barcode = q.barcode;
    for (int t = 0; t < 8 && t < (int)q.name.size() && !q.released; ++t) rec.name[t] = q.name[t];
    R.seqId.push_back(c); R.seqRec.push_back(rec);
    if (hi > lo) {
      R.baseAt.push_back(q.baseOff + lo); R.baseLen.push_back(hi - lo);
      const size_t at0 = R.baseCons.size();
      R.baseCons.resize(at0 + (size_t)(hi - lo)); R.basePw.resize(at0 + (size_t)(hi - lo));
      char *bc = &R.baseCons[at0];
      uint8_t *bp = &R.basePw[at0];
      const int nMarks = (int)q.postCnt.size();
      for (int t = lo; t < hi; ++t) {
        unsigned char mark = 0;   // posting marks (t4_device.h): postings at this offset, and on the first byte whether the contig's marks count
        if (t < nMarks) mark = (unsigned char)((q.postCnt[(size_t)t] > 3 ? 3 : q.postCnt[(size_t)t]) << 5);
        if (t == 0 && q.marksBad) mark |= 128;
        if (t < len) { bc[t - lo] = q.cons[t]; bp[t - lo] = (uint8_t)(t4PwByte(q.pw[t].c[0], q.pw[t].c[1], q.pw[t].c[2], q.pw[t].c[3]) | mark); }
        else { bc[t - lo] = '\0'; bp[t - lo] = (uint8_t)(t4PwByte(0, 0, 0, 0) | mark); }
      }
    }
  }
  dirtySeqs.clear();
  // (the longest contig so far: every contig whose length changes is in this delta, so the maximum over the deltas is an upper bound
  // of the current maximum -- all the image needs it for, t4Key32Bits; walking every contig per delta cost tens of microseconds a round)
  if (maxLen > maxSeqLenSeen) maxSeqLenSeen = maxLen;
  maxLen = maxSeqLenSeen;
  d.table_slots = index.tabSlots; d.table_rebuilt = index.tabRebuilt ? 1 : 0;
  d.post_cap = (int64_t)index.arena.size(); d.base_cap = baseUsed + 1024; d.seq_cap = (int32_t)seqs.size() + 64;
  d.nseq = (int32_t)seqs.size(); d.max_seq_len = maxLen;
  d.n_slots = (int64_t)R.slot.size(); d.slot = R.slot.data(); d.slot_code = R.slotCode.data(); d.slot_start = R.slotStart.data(); d.slot_cnt = R.slotCnt.data();
  d.n_post_runs = (int64_t)R.postAt.size(); d.post_at = R.postAt.data(); d.post_len = R.postLen.data(); d.post_data = R.postData.data();
  d.n_seqs = (int32_t)R.seqId.size(); d.seq_id = R.seqId.data(); d.seq = R.seqRec.data();
  d.n_base_runs = (int64_t)R.baseAt.size(); d.base_at = R.baseAt.data(); d.base_len = R.baseLen.data(); d.base_cons = R.baseCons.data(); d.base_pw = R.basePw.data();
  index.tabRebuilt = false; liveReset = false;
  R.version = ++deltaVersion;
  ++deltas; deltaBytes += (int64_t)(R.slot.size() * 16 + R.postData.size() * 4 + R.seqRec.size() * sizeof(t4_seq_record) + R.baseCons.size() * 2);
  deltaLog.push_back(std::move(recp));
  return T4_OK;
}

int t4_assembler::ensureLanes() {
  if (!lanes.empty()) return T4_OK;
  // One lane by default: measured on an MI355X (profiles/r03b_lanes_sweep.txt), re-querying invalidated entries in the background on
  // further lanes does not shorten the chain -- the entry a commit invalidates is nearly always the next one to be served, so the
  // head waits for a whole query either way (17.9 k waits at 100 k pairs with 3 lanes against 17.8 k with one) -- while every
  // extra launch costs the one host thread its packing, delta and dependency-set time. The machinery stays for T4_LIVE_LANES > 1.
  const int nLanes = knobs.lanes;
  lanes.resize(nLanes < 1 ? 1 : (nLanes > 8 ? 8 : nLanes));
  lanes[0].ctx = ctx;
  for (size_t i = 1; i < lanes.size(); ++i) {
    const int r = t4_init(t4_ctx_device(ctx), &lanes[i].ctx);
    if (r) { err = "a query lane could not be created"; lanes.resize(i); return lanes.size() > 0 ? T4_OK : r; }
    lanes[i].ownCtx = true;
  }
  for (Lane &L : lanes) L.version = deltaVersion - (int64_t)deltaLog.size();
  return T4_OK;
}

// the lane's replica takes every delta it has not seen yet (in order); deltas every lane has applied leave the log
int t4_assembler::bringUpToDate(Lane &L) {
  int r;
  if (!L.dev) { if ((r = t4_index_create(L.ctx, k, 0, &L.dev))) return r; }
  if ((r = t4_index_set_params(L.dev, hitLenRequired, radius, novelSim))) return r;
  for (const auto &rec : deltaLog) {
    if (rec->version <= L.version) continue;
    if ((r = t4_index_apply_delta(L.dev, &rec->d))) return r;
    L.version = rec->version;
  }
  int64_t minV = deltaVersion;
  for (const Lane &x : lanes) if (x.version < minV) minV = x.version;
  while (!deltaLog.empty() && deltaLog.front()->version <= minV) { if (deltaSpare.size() < 8) deltaSpare.push_back(std::move(deltaLog.front())); deltaLog.pop_front(); }
  return T4_OK;
}

// results of jobs nobody waits for any more (the window is dropped): let the kernels finish, forget what they return
void t4_assembler::abandonJobs() {
  for (Lane &L : lanes) {
    if (!L.busy) continue;
    const int32_t *a = nullptr, *b = nullptr, *e = nullptr; const t4_overlap *c = nullptr, *d2 = nullptr;
    (void)t4_add_query_pool_end(L.ctx, &a, &b, &c, &d2, &e);
    L.busy = false;
    for (size_t i = 0; i < L.slots.size(); ++i) { Cached &en = *pool[L.slots[i]]; if (en.uid == L.uids[i]) { en.inflight = false; en.killed = false; en.shifts.clear(); } }
  }
}

int t4_assembler::flushLive(Lane **out) {   // synchronous users (T4_VERIFY_WINDOW): an idle lane's replica, current
  int r;
  if ((r = ensureLanes())) return r;
  Lane *L = nullptr;
  for (Lane &x : lanes) if (!x.busy) { L = &x; break; }
  if (!L) { L = &lanes.back(); if ((r = harvest(*L))) return r; }
  if ((r = makeDelta())) return r;
  if ((r = bringUpToDate(*L))) return r;
  *out = L;
  return T4_OK;
}

// T4_VERIFY_WINDOW=1 (testing aid): the entry about to be served is queried again, alone, against the set as it is NOW, and
// must equal the cached result record for record -- a direct test of the rules that let a cached result survive commits
// (processEvents), instead of relying on the outputs coming out equal.
int t4_assembler::verifyServed(const Cached &c) {
  if (index.total == 0) return c.cnt == 0 ? T4_OK : T4_ERR_STATE;
  int rc;
  Lane *L = nullptr;
  if ((rc = flushLive(&L))) return rc;
  const char *bases = c.read.empty() ? "A" : c.read.c_str();
  const int64_t offs[2] = {0, (int64_t)c.read.size()};
  const int32_t bc = c.barcode, st = c.strand;
  const double fac = (c.barcode == -1 && !c.skip) ? 1.0 : 2.0;
  const int32_t *cnts = nullptr, *bas = nullptr, *rets = nullptr;
  const t4_overlap *ov = nullptr, *ex = nullptr;
  unsigned char hint = c.tier;
  if ((rc = t4_add_query_pool(L->dev, 1, bases, offs, &bc, &st, c.skip, &fac, &cnts, &bas, &ov, &ex, &rets, &hint))) return rc;
  ++verified;
  auto same = [](const t4_overlap &a, const t4_overlap &b) {
    return a.seqIdx == b.seqIdx && a.readStart == b.readStart && a.readEnd == b.readEnd && a.seqStart == b.seqStart && a.seqEnd == b.seqEnd &&
           a.strand == b.strand && a.matchCnt == b.matchCnt && a.indelCnt == b.indelCnt && a.similarity == b.similarity;
  };
  const int n = c.cnt > 0 ? c.cnt : 0;
  bool ok = c.merged ? (cnts[0] > 0 ? cnts[0] : 0) == n : cnts[0] == c.cnt;
  if (ok && c.merged) {
    // an entry that was put together from restricted re-queries holds the records of the fresh query in another order (SeqSet::AddRead
    // sorts them itself, SeqSet.hpp:3474): compared as sets
    std::vector<int> pa(n), pb(n);
    for (int i = 0; i < n; ++i) pa[i] = pb[i] = i;
    auto key = [](const t4_overlap &o) { return std::make_tuple(o.seqIdx, o.strand, o.readStart, o.readEnd, o.seqStart, o.seqEnd); };
    std::sort(pa.begin(), pa.end(), [&](int x, int y) { return key(c.ov[x]) < key(c.ov[y]); });
    std::sort(pb.begin(), pb.end(), [&](int x, int y) { return key(ov[bas[0] + x]) < key(ov[bas[0] + y]); });
    for (int i = 0; ok && i < n; ++i) ok = same(ov[bas[0] + pb[i]], c.ov[pa[i]]) && same(ex[bas[0] + pb[i]], c.ext[pa[i]]) && rets[bas[0] + pb[i]] == c.extRet[pa[i]];
    if (ok) { ++verifiedMerged; return T4_OK; }
  }
  for (int i = 0; ok && i < n; ++i) ok = same(ov[bas[0] + i], c.ov[i]) && same(ex[bas[0] + i], c.ext[i]) && rets[bas[0] + i] == c.extRet[i];
  if (ok) return T4_OK;
  fprintf(stderr, "T4_VERIFY_WINDOW: the cached query of a served read differs from a fresh one (entry %lld, read %s, strand %d): cached %d overlaps, fresh %d\n",
          (long long)c.uid, c.read.c_str(), c.strand, c.cnt, cnts[0]);
  for (int i = 0; i < n && i < (cnts[0] > 0 ? cnts[0] : 0) && i < 6; ++i) {
    const t4_overlap &a = c.ov[i], &b = ov[bas[0] + i], &ea = c.ext[i], &eb = ex[bas[0] + i];
    fprintf(stderr, "  #%d cached seq %d read %d-%d seq %d-%d m %d sim %.6f | ext %d-%d %d-%d ret %d   fresh seq %d read %d-%d seq %d-%d m %d sim %.6f | ext %d-%d %d-%d ret %d\n", i,
            a.seqIdx, a.readStart, a.readEnd, a.seqStart, a.seqEnd, a.matchCnt, a.similarity, ea.readStart, ea.readEnd, ea.seqStart, ea.seqEnd, c.extRet[i],
            b.seqIdx, b.readStart, b.readEnd, b.seqStart, b.seqEnd, b.matchCnt, b.similarity, eb.readStart, eb.readEnd, eb.seqStart, eb.seqEnd, rets[bas[0] + i]);
  }
  err = "T4_VERIFY_WINDOW mismatch";
  return T4_ERR_STATE;
}

// the read's keys (every valid k-mer of both strands) join the window's inverted map
void t4_assembler::registerKmers(Cached &e, int slotId, int shard) {
  auto t0_ = std::chrono::steady_clock::now();
  double secLocal = 0;
  struct Tm { double &acc; std::chrono::steady_clock::time_point t0; ~Tm() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tm_{shard <= 0 ? secRegister : secLocal, t0_};
  // one reference per occurrence (a k-mer that occurs twice in the read is examined twice, each time for one occurrence: the
  // same verdicts, a tolerance budget spent a little faster). Only the codes of the read AS GIVEN are stored: the reverse strand's codes
  // are their reverse complements position by position, so whoever asks for the readers of a code X asks for X (occurrences on the
  // forward strand) and for the reverse complement of X (occurrences on the reverse strand): forEachOcc. Half the insertions -- 284 per
  // read were 6 s of one host thread per config C2 run -- and a map half the size.
  const int len = (int)e.read.size();
  if (len >= k) {
    KCode kc(k);
    uint64_t recent[20]; int nRecent = 0;   // codes of the last k / 2 + 1 valid positions (k <= 31)
    const int back = k / 2 + 1;
    bool near = false;
    for (int i = 0; i < len; ++i) {
      kc.append(e.read[i]);
      if (i < k - 1 || !kc.valid()) { if (i >= k - 1) near = true; continue; }   // (a read with an N: the rule's `last code` runs on over the window of the N -- never booked exactly)
      if (shard <= 0) {   // (one of the registering threads looks for repeats at short distance: Cached::repeatNear)
        for (int t = 0; t < nRecent; ++t) if (recent[t] == kc.code) near = true;
        if (nRecent < back) recent[nRecent++] = kc.code; else { for (int t = 1; t < back; ++t) recent[t - 1] = recent[t]; recent[back - 1] = kc.code; }
      }
      const int h = index.bucket(kc.code, e.barcode);
      const int sh = shardOf(kc.code, h);
      if (shard >= 0 && sh != shard) continue;
      winShard[sh].add(kc.code, h, KOcc{e.uid, slotId, 1, 0, (short)(i - k + 1), -1});
      if (shard < 0) winKmerRefs += 1;
    }
    if (shard <= 0) e.repeatNear = near;
  }
  if (shard >= 0) return;
  ++winKmerLive;
  e.registered = true;
}

// The dependency record of ONE contig for a window entry, after a restricted re-query: hits of the read's k-mers with every
// k-mer of the contig's consensus (a superset of the contig's postings: KmerIndex::BuildIndexFromRead leaves repeated k-mers out)
// per strand, and the hull of the diagonals with three or more of them -- what buildGroups derives from the posting lists.
void t4_assembler::rebuildGroup(Cached &e, int c) {
  const int len = (int)e.read.size();
  if (e.kmerPos.empty() && len >= k) {
    std::string rcs;
    reverseComplement(rcs, e.read);
    for (int st = 0; st < 2; ++st) {
      const std::string &r = st ? rcs : e.read;
      KCode kc(k);
      for (int i = 0; i < len; ++i) {
        kc.append(r[i]);
        if (i < k - 1 || !kc.valid()) continue;
        e.kmerPos.push_back({kc.code, ((i - k + 1) << 1) | (st ? 0 : 1)});
      }
    }
    std::sort(e.kmerPos.begin(), e.kmerPos.end());
  }
  for (uint32_t plus = 0; plus < 2; ++plus) { Grp *g = e.findGroup((uint32_t)c * 2u + plus); if (g) { g->cnt = 0; g->lo = 0x7FFFFFFF; g->hi = -0x7FFFFFFF; } }
  if (c < 0 || c >= (int)seqs.size() || seqs[c].released) return;
  const std::string &cons = seqs[c].cons;
  static thread_local std::vector<int64_t> diag;   // (plus << 40) | (start of the read on the contig + 2^30), one per hit
  diag.clear();
  KCode kc(k);
  for (int i = 0; i < (int)cons.size(); ++i) {
    kc.append(cons[i]);
    if (i < k - 1 || !kc.valid()) continue;
    const int off = i - k + 1;
    auto it = std::lower_bound(e.kmerPos.begin(), e.kmerPos.end(), std::make_pair(kc.code, 0));
    for (; it != e.kmerPos.end() && it->first == kc.code; ++it) diag.push_back(((int64_t)(it->second & 1) << 40) | (int64_t)(off - (it->second >> 1) + (1 << 30)));
  }
  std::sort(diag.begin(), diag.end());
  for (size_t i = 0; i < diag.size();) {
    size_t j = i + 1;
    while (j < diag.size() && diag[j] == diag[i]) ++j;
    const uint32_t plus = (uint32_t)(diag[i] >> 40);
    Grp &g = e.getGroup((uint32_t)c * 2u + plus);
    g.cnt += (uint32_t)(j - i);
    if (j - i >= 3) {
      const int at = (int)(diag[i] & ((1ll << 40) - 1)) - (1 << 30);
      if (at < g.lo) g.lo = at;
      if (at + len - 1 > g.hi) g.hi = at + len - 1;
    }
    i = j;
  }
}

// _hit records GetHitsFromRead (SeqSet.hpp:1341-1501, allowTotalSkip false) emits for the entry's read against the current index:
// the list sizes of its k-mers under the repeat-skip rule (a list of 100+ postings is passed over while fewer than k / 2 were since
// the last emitted one; a k-mer equal to the last one not passed over is not looked up again). What the query kernel's seed stage
// will find, so that the launch knows which reads the wide query will serve.
int t4_assembler::emittedHits(Cached &e) {
  std::string rcs;
  reverseComplement(rcs, e.read);
  const int len = (int)e.read.size(), skipLimit = k / 2;
  memset(e.emitMask, 0, sizeof e.emitMask);
  e.maskOk = len - k + 1 <= T4_MAX_READ_KMERS;
  if (len < k) return 0;
  int64_t H = 0;
  bool huge = false;
  uint64_t prev = 0;
  for (int st = 0; st < 2; ++st) {
    if (st == 0 ? e.strand == -1 : e.strand == 1) continue;
    const std::string &r = st ? rcs : e.read;
    KCode kc(k);
    int skipCnt = 0;
    for (int i = 0; i < len; ++i) {
      kc.append(r[i]);
      if (i < k - 1) continue;
      if (i == k - 1 || prev != kc.code) {
        const ListRef *l = kc.valid() ? index.find(kc.code, index.bucket(kc.code, e.barcode)) : nullptr;
        const uint32_t size = l ? l->cnt : 0;
        if (size >= 100 && i != k - 1 && i != len - 1 && skipCnt < skipLimit) { ++skipCnt; continue; }
        if (size >= 100 && e.skip) continue;   // allowTotalSkip (repetitive data, --trimLevel 2: SeqSet.hpp:1392-1393)
        skipCnt = 0;
        if (e.maskOk) e.emitMask[st][(i - k + 1) >> 6] |= 1ull << ((i - k + 1) & 63);   // looked up, not passed over (Cached::emitMask)
        H += size;
        if (size > 10000) huge = true;   // a list beyond 10000 postings: the wide query whatever the total (removeOnlyRepeats)
      }
      prev = kc.code;
    }
  }
  return huge || H > 0x7FFFFFFF ? 0x7FFFFFFF : (int)H;
}

// Hits of the read per (strand, contig) against the current index (host replica; read-only here): the number of hits, and the
// stretch of the contig the read lies on along every diagonal that holds three or more hits -- a candidate run of a novel
// contig is a run of hits on ONE diagonal (adjustRadius 0, SeqSet.hpp:906-919) of at least minHitRequired >= 3 hits, and
// everything the query then reads of the contig (gap DPs, ExtendOverlap, the distance tests to the contig ends) lies within
// the read's projection along that diagonal +- radius.
void t4_assembler::buildGroups(Cached &e) {
  // Round 6: the hits GetHitsFromRead EMITS (SeqSet.hpp:1341-1501 with its repeat-skip rule replayed: a list of 100+ postings is passed
  // over while fewer than k / 2 were since the last emitted one, a k-mer equal to the last one looked up is not looked up again; the
  // strands the query searches; the barcode filter of a set that holds several barcodes) -- the records the wide query returns for the
  // reads it serves, derived here for a read of the LDS tier: same content, same order (devGroups), so that every rule that holds for
  // device records holds for these (exact sizes: Cached::inexact, exactStats). Until round 6 this table counted every posting of every
  // k-mer of both strands -- a superset that cost thousands of postings per k-mer of a shared gene segment and pinned nothing down.
  std::string rcs;
  reverseComplement(rcs, e.read);
  const int len = (int)e.read.size(), skipLimit = k / 2;
  struct Slot { uint64_t key; uint32_t cnt; };
  static thread_local std::vector<Slot> tab;
  static thread_local std::vector<uint32_t> usedSlots;
  struct Emit { const ListRef *l; int a; uint32_t plus; };
  static thread_local std::vector<Emit> emits;
  emits.clear();
  size_t nPost = 0;
  uint32_t maxList = 0;
  uint64_t prev = 0;
  memset(e.emitMask, 0, sizeof e.emitMask);
  e.maskOk = len - k + 1 <= T4_MAX_READ_KMERS;
  for (int st = 0; st < 2 && len >= k; ++st) {
    if (st == 0 ? e.strand == -1 : e.strand == 1) continue;
    const std::string &r = st ? rcs : e.read;
    KCode kc(k);
    int skipCnt = 0;
    for (int i = 0; i < len; ++i) {
      kc.append(r[i]);
      if (i < k - 1) continue;
      if (i == k - 1 || prev != kc.code) {
        const ListRef *l = kc.valid() ? index.find(kc.code, index.bucket(kc.code, e.barcode)) : nullptr;
        const uint32_t size = l ? l->cnt : 0;
        if (size > maxList) maxList = size;
        if (size >= 100 && i != k - 1 && i != len - 1 && skipCnt < skipLimit) { ++skipCnt; continue; }
        if (size >= 100 && e.skip) continue;   // allowTotalSkip (repetitive data, --trimLevel 2: SeqSet.hpp:1392-1393)
        skipCnt = 0;
        if (e.maskOk) e.emitMask[st][(i - k + 1) >> 6] |= 1ull << ((i - k + 1) & 63);   // looked up, not passed over (Cached::emitMask)
        if (size) { emits.push_back(Emit{l, i - k + 1, st ? 0u : 1u}); nPost += size; }
      }
      prev = kc.code;
    }
  }
  {
    size_t sz = 1024;
    while (sz < 2 * nPost + 2) sz <<= 1;
    if (tab.size() < sz) tab.assign(sz, Slot{~0ull, 0});
    usedSlots.clear();
  }
  const size_t mask = tab.size() - 1;
  for (const Emit &em : emits)
    for (uint32_t t = 0; t < em.l->cnt; ++t) {
      const Post &p = index.arena[em.l->start + t];
      if (e.barcode != -1 && seqs[(size_t)p.idx].barcode != e.barcode) continue;   // (SeqSet.hpp:1418)
      const uint64_t key = (((uint64_t)(uint32_t)p.idx * 2u + em.plus) << 32) | (uint32_t)(p.offset - em.a + (1 << 30));
      size_t s2 = (size_t)mix64h(key) & mask;
      while (tab[s2].key != key && tab[s2].key != ~0ull) s2 = (s2 + 1) & mask;
      if (tab[s2].key == ~0ull) { tab[s2].key = key; tab[s2].cnt = 0; usedSlots.push_back((uint32_t)s2); }
      ++tab[s2].cnt;
    }
  e.groups.reset((uint32_t)(usedSlots.size() < 64 ? 64 : usedSlots.size()));
  for (uint32_t s2 : usedSlots) {
    const Slot &x = tab[s2];
    Grp &g = e.groups.get((uint32_t)(x.key >> 32));
    g.cnt += x.cnt;
    if (x.cnt >= 3) {
      const int at = (int)(uint32_t)x.key - (1 << 30);
      if (at < g.lo) g.lo = at;
      if (at + len - 1 > g.hi) g.hi = at + len - 1;
    }
  }
  for (uint32_t s2 : usedSlots) tab[s2].key = ~0ull;   // leave the scratch table empty
  // the records in the order of the device's: minus strand (even keys) by contig, then plus strand
  e.devGroups.clear();
  int u4 = 0;
  for (const Grp &g : e.groups.t) if (g.key != 0xFFFFFFFFu) { e.devGroups.push_back(g); if (g.cnt >= 4) ++u4; }
  std::sort(e.devGroups.begin(), e.devGroups.end(), [](const Grp &a, const Grp &b) { return (a.key & 1u) != (b.key & 1u) ? (a.key & 1u) < (b.key & 1u) : a.key < b.key; });
  e.devSplit = 0;
  while (e.devSplit < e.devGroups.size() && !(e.devGroups[e.devSplit].key & 1u)) ++e.devSplit;
  e.hasDev = true; e.hostRecords = true;
  e.groups.reset(16);
  // possibleOverlapCnt counts groups measured at more than 3 hits (SeqSet.hpp:784-810); while it cannot pass 100 the
  // group statistics of GetOverlapsFromHits leave novelMinHitRequired at 3 whatever small groups come and go
  e.slack = 99 - u4;               // negative: no tolerated edit at all (the statistics are live for this read)
  e.fragile = maxList > 10000;     // lists beyond 10000 postings drive removeOnlyRepeats (SeqSet.hpp:802)
}

// Examine what the commit(s) since the last call changed for every window entry that is still valid.
void t4_assembler::processEvents() {
  if (!live()) return;
  if (order.empty() || (idxEvents.empty() && structEvents.empty())) { idxEvents.clear(); structEvents.clear(); return; }
  auto t0_ = std::chrono::steady_clock::now();
  struct Tm { double &acc; std::chrono::steady_clock::time_point t0; ~Tm() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tm_{secEvents, t0_};
  // (an entry whose query is still running is examined like one that holds a result: its dependency sets were derived at launch,
  // from the state its lane's replica holds)
  // the whole result of the entry falls
  auto kill = [&](Cached &e, int64_t &why) {
    e.lastKill = (unsigned char)(&why == &invKey ? 1 : &why == &invCross ? 2 : &why == &invRegion ? 3 : &why == &invShift ? 4 : &why == &invContig ? 5 : 6);
    if (e.partial) { e.partial = false; e.pendingContig = -1; e.morePending.clear(); if (e.inflight) e.killed = true; ++invalidations; ++why; return; }
    if (e.valid) { e.valid = false; ++invalidations; ++why; }
    else if (e.inflight && !e.killed) { e.killed = true; ++invalidations; ++why; }
  };
  // what the entry's result takes from contig c falls: when the entry qualifies (Cached: restricted re-query) it keeps the rest
  auto touch = [&](Cached &e, int c, int64_t &why) {
    if (e.partial) {
      if (!e.isPending(c)) {
        // another contig of an entry that already waits for one: with the candidate store it waits for both (the re-queries are
        // independent of each other; one that is on its way stays good), else the whole entry falls
        if (candStore && e.candOk && (int)e.morePending.size() < knobs.maxPending - 1) { e.morePending.push_back(c); ++invalidations; ++why; ++restrictedMarks; ++restrictedMulti; return; }
        kill(e, why); return;
      }
      if (e.inflight && !e.killed) { e.killed = true; ++restrictedStale; }   // the restricted query on its way saw the contig before this change
      return;
    }
    if (eligibleForRestricted(e)) { e.valid = false; e.partial = true; e.pendingContig = c; ++invalidations; ++why; ++restrictedMarks; return; }
    kill(e, why);
  };
  // structural events, in the order they happened
  for (const StructEv &ev : structEvents) {
    for (int sl : order) {
      Cached &e = *pool[sl];
      if (!e.standing()) continue;
      if (e.inflight && e.expectWide && !e.hasDev) { kill(e, invContig); continue; }   // its dependency records are still on their way: nothing to examine the change against
      if (e.isPending(ev.c)) {   // (its records and its group of this contig are stale already: any change of it counts)
        if (ev.kind == 2 && knobs.contigKills) kill(e, invContig); else touch(e, ev.c, ev.kind == 0 ? invRegion : ev.kind == 1 ? invShift : invContig);
        continue;
      }
      const int margin = radius + 2;
      for (uint32_t plus = 0; plus < 2; ++plus) {
        Grp *g = e.findGroup((uint32_t)ev.c * 2u + plus);
        if (!g || g->cnt == 0) continue;
        if (ev.kind == 2) {
          // A contig that took part in a merge (SeqSet.hpp:3878-4130: the survivor carries the joined consensus, the others are
          // released). Until round 6 every entry with a hit on it fell whole -- 35 k entries of config C2, each a whole-query round when it
          // comes to the head. What such an entry's result takes from the contig is what a restricted re-query asks for again: nothing
          // of a released contig (it has no posting left), the overlaps with the new consensus of the survivor -- and the removals and
          // insertions of the merge reach the entry's small groups as index events like any other edit. A group below three hits holds
          // no candidate (SeqSet.hpp:923-925) and is left to those events.
          if (knobs.contigKills) { kill(e, invContig); break; }
          if (g->cnt >= 3 || g->lo <= g->hi) { touch(e, ev.c, invContig); break; }
          continue;
        }
        if (g->lo > g->hi) continue;   // no diagonal with three hits: nothing of the contig is read
        if (ev.kind == 0) {
          if (ev.a <= g->hi + margin && ev.b > g->lo - margin) { touch(e, ev.c, invRegion); break; }
        } else {   // shift: the old columns 0 and 1 changed and the left end moved away
          if (g->lo - margin < 2) { touch(e, ev.c, invShift); break; }
          g->lo += ev.a; g->hi += ev.a;
        }
      }
      if (ev.kind == 1 && e.isPending(ev.c)) {
        // (the contig's other-strand group, if the loop stopped at the first: its hull moves all the same -- it is rebuilt with the re-query)
      } else if (ev.kind == 1 && (e.valid || e.partial)) {
        for (t4_overlap &o : e.ov) if (o.seqIdx == ev.c) { o.seqStart += ev.a; o.seqEnd += ev.a; }
        for (t4_overlap &o : e.ext) if (o.seqIdx == ev.c) { o.seqStart += ev.a; o.seqEnd += ev.a; }
        for (t4_cand &o : e.cands) if (o.seqIdx == ev.c) { o.ss += ev.a; o.se += ev.a; }
      } else if (ev.kind == 1 && e.standing()) e.shifts.push_back({ev.c, ev.a});   // its records are still to come
    }
  }
  structEvents.clear();
  ts.lap(TS_EVENTS_STRUCT);
  // index events: net effect per (key, posting)
  if (!idxEvents.empty()) {
    bool ins = false, rem = false;
    for (const IdxEv &ev : idxEvents) { if (ev.delta > 0) ins = true; else rem = true; }
    // (both by sorting event numbers in scratch vectors that live with the builder: this runs after every commit, and hash
    // maps built and torn down each time were most of its cost)
    std::vector<IdxEv> &net = evNet;
    net.clear();
    std::vector<uint32_t> &ordv = evOrder;
    const uint32_t nEv = (uint32_t)idxEvents.size();
    if (ins && rem) {
      ordv.resize(nEv);
      for (uint32_t i = 0; i < nEv; ++i) ordv[i] = i;
      std::sort(ordv.begin(), ordv.end(), [&](uint32_t x, uint32_t y) {
        const IdxEv &p = idxEvents[x], &q = idxEvents[y];
        if (p.code != q.code) return p.code < q.code;
        if (p.h != q.h) return p.h < q.h;
        if (p.idx != q.idx) return p.idx < q.idx;
        if (p.off != q.off) return p.off < q.off;
        return x < y;
      });
      evSum.assign(nEv, 0);   // net delta, kept at the first event of every (key, posting)
      for (uint32_t g = 0; g < nEv;) {
        uint32_t e2 = g; int sum = 0;
        const IdxEv &p = idxEvents[ordv[g]];
        while (e2 < nEv) { const IdxEv &q = idxEvents[ordv[e2]]; if (q.code != p.code || q.h != p.h || q.idx != p.idx || q.off != p.off) break; sum += q.delta; ++e2; }
        evSum[ordv[g]] = sum;
        g = e2;
      }
      for (uint32_t i = 0; i < nEv; ++i) if (evSum[i] != 0) { IdxEv x = idxEvents[i]; x.delta = evSum[i]; net.push_back(x); }
    } else net = idxEvents;
    // list sizes before the first and after the last event of every key: a size that crosses 100 changes which k-mers
    // GetHitsFromRead skips (SeqSet.hpp:1381-1391)
    ordv.clear();
    for (uint32_t i = 0; i < nEv; ++i) if (idxEvents[i].sizeAfter != 0xFFFFFFFFu) ordv.push_back(i);
    std::sort(ordv.begin(), ordv.end(), [&](uint32_t x, uint32_t y) {
      const IdxEv &p = idxEvents[x], &q = idxEvents[y];
      if (p.code != q.code) return p.code < q.code;
      if (p.h != q.h) return p.h < q.h;
      return x < y;
    });
    static thread_local std::vector<std::pair<uint64_t, int>> longBefore;   // keys whose list held 100 postings or more before this commit (sorted: ordv is)
    longBefore.clear();
    for (size_t g = 0; g < ordv.size();) {
      size_t e2 = g;
      const IdxEv &first = idxEvents[ordv[g]];
      while (e2 < ordv.size() && idxEvents[ordv[e2]].code == first.code && idxEvents[ordv[e2]].h == first.h) ++e2;
      const uint32_t before = first.delta > 0 ? first.sizeAfter - 1 : first.sizeAfter + 1, after = idxEvents[ordv[e2 - 1]].sizeAfter;
      g = e2;
      if (before >= 100) longBefore.push_back({first.code, first.h});
      if ((before >= 100) == (after >= 100) && (before > 10000) == (after > 10000)) continue;   // 10000: removeOnlyRepeats / the repeat test of a run (SeqSet.hpp:802, 876, 936)
      forEachOcc(first.code, first.h, [&](const KOcc &o, bool) {
        Cached &e = *pool[o.slot];
        if (e.uid == o.uid) kill(e, invCross);
      });
    }
    static thread_local std::vector<int> checkSlots;   // entries whose threshold is checked by repeating the statistics loop once this commit's edits are booked
    checkSlots.clear();
    for (const IdxEv &ev : net) {
      // is the edited k-mer certain to be emitted by GetHitsFromRead (Cached::inexact)? its list below 100 postings before the commit
      // and now (an edit that crosses 100 has ended the entries above; a move leaves the size as it is) -- looked up when first asked for
      int shortList = -1;
      const auto listIsShort = [&]() {
        if (shortList < 0) {
          const ListRef *l = index.find(ev.code, ev.h);
          shortList = (!l || l->cnt < 100u) && !std::binary_search(longBefore.begin(), longBefore.end(), std::make_pair(ev.code, ev.h)) ? 1 : 0;
        }
        return shortList == 1;
      };
      forEachOcc(ev.code, ev.h, [&](const KOcc &o, bool onForward) {
        Cached &e = *pool[o.slot];
        if (e.uid != o.uid || !e.standing()) return;
        if ((e.strand == 1 && !onForward) || (e.strand == -1 && onForward)) return;   // (a strand the query of this read does not search: SeqSet.hpp:1352-1356)
        if (e.maskOk) {   // (a position GetHitsFromRead passes over, or does not look up again: no hit of this read changes with the list)
          const int lenk = (int)e.read.size() - k, sp = onForward ? (int)o.pos : lenk - (int)o.pos;
          if (sp < 0 || sp > lenk || !((e.emitMask[onForward ? 0 : 1][sp >> 6] >> (sp & 63)) & 1ull)) return;
        }
        if (e.inflight && e.expectWide && !e.hasDev) { kill(e, invKey); return; }
        // (a read with lists beyond 10000 postings: removeOnlyRepeats and the run test of SeqSet.hpp:934-940 look across its groups. With
        // the query's own records -- exact sizes, the hits of shorter lists per group -- the edit is booked and the entry checked below;
        // without them every edit of one of its keys ends it, as until round 6)
        // Measured on the first 2 M pairs of C3 (profiles/r06p_*): 14 745 such checks, 17 entries fell, the 320 k kills of this kind gone,
        // outputs identical -- and the run no faster: the entries fall a little later to a change of one of their groups of three or more
        // hits, because no restricted re-query exists for them. Off unless T4_FRAGILE_CHECKS is set, until that path exists
        // and T4_VERIFY_WINDOW has been run at a depth where such reads occur.
        if (e.fragile && !(knobs.fragileChecks && knobs.exactTolerance && e.rorOk && e.maskOk && !e.inexact && e.hasDev && e.candOk)) { kill(e, invFragile); ++invLongLists; return; }
        if (e.isPending(ev.idx)) { touch(e, ev.idx, invKey); return; }
        for (uint32_t plus = 0; plus < 2 && e.standing(); ++plus) {
          const int n = ((plus != 0) == onForward ? 1 : 0) * (ev.delta > 0 ? ev.delta : -ev.delta);
          if (!n) continue;
          if (ev.delta > 0) {
            Grp &g = e.getGroup((uint32_t)ev.idx * 2u + plus);
            if (g.cnt + (uint32_t)n >= 3) { touch(e, ev.idx, invKey); break; }   // (the recorded size of a group of three or more stays what its last query found: mergeRestricted reads it)
            g.cnt += (uint32_t)n;
          } else {
            Grp *g = e.findGroup((uint32_t)ev.idx * 2u + plus);
            if (g && g->cnt >= 3) { touch(e, ev.idx, invKey); break; }
            // (device records count emitted hits, see Cached::devGroups: a removal is subtracted from them only when the k-mer is certain to have been emitted)
            if (g && (!e.hasDev || (!e.inexact && (e.maskOk || (!e.repeatNear && listIsShort()))))) g->cnt = g->cnt > (uint32_t)n ? g->cnt - (uint32_t)n : 0;
          }
          ++tolerated; ++e.toleratedSince;
          if (e.fragile) e.editedKeys.push_back((uint32_t)ev.idx * 2u + plus);
          if (e.hasDev && !e.inexact && !e.maskOk && (e.repeatNear || !listIsShort())) e.inexact = true;
          if (e.statsStable) { ++toleratedStable; continue; }   // exact: the statistics of this read's query cannot move (overlapsFromKeys)
          if (e.fragile && e.inexact) { kill(e, invFragile); ++invLongLists; break; }
          if (knobs.exactTolerance && e.hasDev && !e.inexact && e.candOk) {   // the statistics loop decides, below
            if (!e.checkPending) { e.checkPending = true; checkSlots.push_back(o.slot); }
            continue;
          }
          if (--e.slack < 0) { kill(e, invFragile); break; }
        }
      });
    }
    // The entries that took an edit of a small group which their query could not certify harmless, and whose group sizes are exact: the
    // statistics loop of SeqSet.hpp:784-811 over the sizes as they are now. The same novelMinHitRequired on both strands: every
    // candidate run is judged as before (866, 923) and the result stands. (An entry that waits for a restricted re-query is settled
    // when its records arrive: mergeRestricted.)
    for (int sl : checkSlots) {
      Cached &e = *pool[sl];
      e.checkPending = false;
      if (!e.valid) { e.editedKeys.clear(); continue; }
      int T[2], e4[2], e5[2], big[2];
      bool ror[2] = {false, false}, headTouched = false;
      ++toleranceChecks;
      const int headLen = e.smhi[0] > e.smhi[1] ? e.smhi[0] : e.smhi[1];
      bool ok = !e.inexact && exactStats(e, -1, nullptr, T, e4, e5, big, e.fragile ? ror : nullptr, headLen, e.fragile ? &headTouched : nullptr) && T[0] == e.minT[0] && T[1] == e.minT[1];
      if (ok && e.fragile) { ++fragileChecks; ok = e.rorOk && ror[0] == e.ror0[0] && ror[1] == e.ror0[1] && !headTouched; if (!ok) ++invLongLists; }
      e.editedKeys.clear();
      if (!ok) { kill(e, invFragile); ++toleranceCheckKills; }
    }
    idxEvents.clear();
    ts.lap(TS_EVENTS_INDEX);
  }
}

// The driver announces the next n reads it will offer to AddRead, in order, with the arguments it will offer them with.
// Entries already in the window that line up are kept (with their results, or with the query that is running for them);
// what does not line up is dropped, the rest of the announcement becomes new entries without a result.
void t4_assembler::announceLive(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive) {
  size_t keep = 0;
  for (; keep < order.size() && keep < (size_t)n; ++keep) {
    const Cached &c = *pool[order[keep]];
    if (c.read != reads[keep] || c.strand != strands[keep] || c.barcode != (barcodes ? barcodes[keep] : -1) || c.skip != repetitive) break;
  }
  while (order.size() > keep && keep < (size_t)n) {   // (entries beyond the announced ones stay when every announced read lined up)
    const int sl = order.back(); order.pop_back();
    pool[sl]->valid = false; pool[sl]->partial = false; pool[sl]->inflight = false; pool[sl]->uid = 0; freeSlots.push_back(sl);   // a running query of it is ignored when it returns (uid)
    if (pool[sl]->registered) { --winKmerLive; pool[sl]->registered = false; }
  }
  if (winKmerRefs > 64 * 142 && winKmerRefs > 4 * (winKmerLive + 1) * 142) {   // mostly references of retired entries: rebuild
    for (WinKmers &w : winShard) w.clear(); winKmerRefs = 0; winKmerLive = 0;
    for (int sl : order) if (pool[sl]->registered) registerKmers(*pool[sl], sl);
  }
  for (size_t i = keep; i < (size_t)n; ++i) {
    int sl;
    if (!freeSlots.empty()) { sl = freeSlots.back(); freeSlots.pop_back(); }
    else { sl = (int)pool.size(); pool.push_back(new Cached()); }
    Cached &c = *pool[sl];
    c.read = reads[i]; c.strand = strands[i]; c.barcode = barcodes ? barcodes[i] : -1; c.skip = repetitive; c.cnt = 0; c.valid = false;
    c.inflight = false; c.killed = false; c.shifts.clear();
    c.ov.clear(); c.ext.clear(); c.extRet.clear();
    c.uid = nextUid++; c.tier = 0; c.hintPredicted = false; c.lastUs = 0; c.registered = false; c.lastKill = 0;
    c.hasDev = false; c.hostRecords = false; c.expectWide = false; c.devGroups.clear(); c.dbgGroups.clear(); c.devInfo.clear(); c.rorOk = false; c.editedKeys.clear();
    c.partial = false; c.pendingContig = -1; c.merged = false; c.auxOk = false; c.restrictedCount = 0; c.kmerPos.clear();
    c.cands.clear(); c.candOk = false; c.exactKeys.clear(); c.morePending.clear(); c.inexact = false; c.repeatNear = false; c.checkPending = false; c.maskOk = false;
    order.push_back(sl);
  }
}

// One query launch: the lane's replica is brought up to date, the reads go out, and while the kernels run the dependency sets of the
// queried reads are derived from the host replica (which IS the state the lane's replica now holds). Returns without waiting.
int t4_assembler::launchOn(Lane &L, const std::vector<int> &todo, int repetitive) {
  auto tl0_ = std::chrono::steady_clock::now();
  struct Tl { double &acc; std::chrono::steady_clock::time_point t0; ~Tl() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tl_{secLaunch, tl0_};
  int rc;
  ts.lap(TS_PUMP_OTHER);
  if ((rc = makeDelta())) return rc;
  { auto t0 = std::chrono::steady_clock::now(); rc = bringUpToDate(L); secDelta += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); if (rc) return rc; }
  const int m = (int)todo.size();
  ts.lap(TS_DELTA);
  wideQueries = knobs.wideQueries; wideHitLimit = knobs.wideHitLimit;   // what t4_add_query_pool_begin does with a heavy read under the testing aids of csrc/t4_api.hip
  // one item per whole query, one per contig a partial entry waits for (the items of an entry are adjacent)
  L.slots.clear(); L.uids.clear(); L.hint.clear(); L.bcs.clear(); L.sts.clear(); L.fac.clear(); L.only.clear(); L.force.clear();
  L.bases.clear(); L.offs.assign(1, 0); L.repetitive = repetitive;
  bool anyOnly = false;
  for (int i = 0; i < m; ++i) {
    Cached &c = *pool[todo[i]];
    const int nItems = c.partial ? 1 + (int)c.morePending.size() : 1;
    for (int t = 0; t < nItems; ++t) {
      L.slots.push_back(todo[i]); L.uids.push_back(c.uid); L.hint.push_back(c.tier);
      L.bases += c.read; L.offs.push_back((int64_t)L.bases.size()); L.bcs.push_back(c.barcode); L.sts.push_back(c.strand);
      L.fac.push_back((c.barcode == -1 && !repetitive) ? 1.0 : 2.0);   // ExtendOverlap's mismatch factor (SeqSet.hpp:3597-3598)
      L.only.push_back(c.partial ? (t == 0 ? c.pendingContig : c.morePending[(size_t)t - 1]) : -1);
      L.force.push_back((c.partial && c.candOk) ? ((c.minT[0] & 0xFFFF) | ((c.minT[1] & 0xFFFF) << 16)) : 0);   // the threshold the entry's other candidates were made with
    }
    c.inflight = true; c.killed = false; c.shifts.clear();
    if (c.partial) { anyOnly = true; continue; }   // restricted re-query: the entry keeps what it holds of the other contigs
    c.statsStable = false;
    c.hasDev = false; c.hostRecords = false; c.expectWide = false; c.devGroups.clear(); c.dbgGroups.clear(); c.devInfo.clear(); c.rorOk = false; c.editedKeys.clear();
    c.auxOk = false; c.merged = false; c.restrictedCount = 0;
    c.cands.clear(); c.candOk = false; c.exactKeys.clear();
  }
  const int mi = (int)L.slots.size();
  bool anyOnlyAll = true;
  for (int v : L.only) if (v < 0) { anyOnlyAll = false; break; }
  { int nOnly = 0; for (int v : L.only) nOnly += v >= 0 ? 1 : 0; restrictedQueries += nOnly; wholeQueries += mi - nOnly; if (nOnly == mi) ++restrictedOnlyRounds; }
  laneT0 = tl0_;
  if (L.bases.empty()) L.bases.push_back('A');
  ts.lap(TS_LAUNCH_PACK);
  {
    auto tq0 = std::chrono::steady_clock::now();
    candStore = knobs.candStore;
    rc = t4_add_query_pool_begin2(L.dev, mi, L.bases.data(), L.offs.data(), L.bcs.data(), L.sts.data(), repetitive, L.fac.data(), L.hint.data(), anyOnly ? L.only.data() : nullptr,
                                  anyOnly ? L.force.data() : nullptr, (candStore ? 1 : 0) | (knobs.useMarks ? 2 : 0));
    secQuery += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count();
  }
  ts.lap(TS_LAUNCH_CALL);
  if (rc) { for (int sl : todo) pool[sl]->inflight = false; return rc; }
  L.busy = true; L.polls = 0;
  ++queries; ++rounds; ++launches; readsQueried += m;
  // the hit groups of the queried reads come from the host replica of the index while the GPU runs the query
  // ... and so do the window's inverted k-mer map entries of the reads queried for the first time (one thread: the map has one
  // writer and, until this returns, no reader)
  auto tg1 = std::chrono::steady_clock::now();
  std::atomic<int> nextG(0);
  std::atomic<int> regShard(0);
  restrictOn = knobs.restrictOn;
  std::vector<int> regList;
  for (int sl : todo) if (!pool[sl]->registered) regList.push_back(sl);
  const auto registerNew = [&]() {   // a shard of the map per taker (the first WK_SHARDS threads that come by)
    for (;;) {
      const int sh = regShard.fetch_add(1);
      if (sh >= WK_SHARDS) break;
      for (int sl : regList) registerKmers(*pool[sl], sl, sh);
    }
  };
  const std::function<void()> groupWorker = [&]() {
    if (!regList.empty() && regShard.load() < WK_SHARDS) registerNew();
    for (;;) {
      int i = nextG.fetch_add(1);
      if (i >= m) break;
      Cached &e = *pool[todo[i]];
      if (e.partial) continue;   // (its group of the one contig is rebuilt when the records arrive)
      // a read whose emitted hits outgrow the LDS tier is served by the wide query, which returns its dependency records itself
      // (deriving them here would cost several times the query: every posting of every k-mer, the ones the repeat-skip rule passes over included)
      if (wideQueries && !repetitive && e.barcode == -1 && emittedHits(e) > wideHitLimit) {
        // (T4_VERIFY_WINDOW: the host's replay of the emitted hits is derived all the same and held against the wide query's records when they arrive)
        if (knobs.verifyWindow) { buildGroups(e); e.dbgGroups.swap(e.devGroups); e.devGroups.clear(); e.hasDev = false; e.hostRecords = false; }
        e.expectWide = true; e.groups.reset(16); e.slack = -1; e.fragile = true;
      }
      else buildGroups(e);
    }
  };
  // While this round's whole queries run, the reads of the NEXT such round are looked at: a read that has never been queried and
  // whose seed stage will emit more hits than the LDS tier takes starts on the wide pipeline at once, on the second stream beside the
  // round's query kernel (Cached::tier is that hint), instead of being deferred by the query kernel first -- a fresh heavy read then
  // costs its round the wide kernels, not the query kernel AND the wide kernels one after the other. A count against the replica as
  // it is now; it is a hint: either path serves any read.
  std::vector<int> predict;
  std::atomic<int> nextP(0);
  if (wideQueries && !repetitive && mi > 0 && !anyOnlyAll && knobs.predictHints) {
    size_t seen = 0;
    for (size_t i = 0; i < order.size() && seen < 48; ++i) {
      Cached &e = *pool[order[i]];
      if (e.valid || e.inflight || e.partial || e.hintPredicted || e.tier || e.lastUs || e.barcode != -1) continue;
      predict.push_back(order[i]); ++seen;
    }
  }
  const std::function<void()> worker2 = [&]() {
    groupWorker();
    for (;;) {
      const int t = nextP.fetch_add(1);
      if (t >= (int)predict.size()) break;
      Cached &e = *pool[predict[(size_t)t]];
      if (emittedHits(e) > wideHitLimit) e.tier = 1;
      e.hintPredicted = true;
    }
  };
  const int work = m + (int)predict.size();
  const int nHelp = work >= 2 ? (threads - 1 < work - 1 ? threads - 1 : work - 1) : 0;
  if (nHelp > 0) { if (!helpers) helpers.reset(new HelperPool()); helpers->start(nHelp, worker2); }
  worker2();
  if (nHelp > 0) helpers->wait();
  for (int sl : regList) {   // (the shards are filled: the bookkeeping of the registrations, once)
    Cached &e = *pool[sl];
    const int len = (int)e.read.size();
    if (len >= k) winKmerRefs += (size_t)(len - k + 1);   // (what the rebuild of a map full of retired references is decided on: an N less does not matter)
    ++winKmerLive; e.registered = true;
  }
  secGroups += std::chrono::duration<double>(std::chrono::steady_clock::now() - tg1).count();
  ts.lap(TS_LAUNCH_GROUPS);
  return T4_OK;
}

// The scan of SeqSet.hpp:1673-2094 over a read's candidate overlaps (all on one strand, in scan order), as far as it looks across
// contigs: which candidates the pre-filters of 1705-1794 cut. Every candidate carries its scored fields (the kernels score all of
// them and replay the filters afterwards -- prefilterNovel, wideMergeKernel -- as does this function); bestNovelOverlap is the best
// SCORED overlap so far by _overlap::operator< (2025-2027).
void t4_assembler::replayScan(const std::vector<t4_cand> &cands, const std::vector<Seq> &seqs, int len, int radius, double repeatSim, std::vector<unsigned char> &cut) {
  const int n = (int)cands.size();
  cut.assign((size_t)n, 0);
  if (n <= 50) return;   // (1705: the pre-filters are off)
  auto simOf = [](const t4_cand &o) { return (o.flags & 2) ? 0.0 : (double)o.matchCnt / (double)(o.se - o.ss + 1 + o.re - o.rs + 1); };
  int best = -1; double bs = 0;
  for (int i = 0; i < n; ++i) {
    const t4_cand &o = cands[i];
    if (best != -1) {
      const t4_cand &bn = cands[best];
      const int m0 = o.m0;
      bool c = false;
      if (bn.rs == 0 && bn.re == len - 1) {
        if (bs == 1) c = true;
        else if (bs > repeatSim && m0 < 0.9 * bn.matchCnt) c = true;
      }
      if (!c && bn.rs + len - 1 - bn.re < radius) {
        if (bs == 1 && m0 < 0.9 * bn.matchCnt) c = true;
        else if (bs > repeatSim && m0 < 0.8 * bn.matchCnt) c = true;
      }
      if (!c && o.ss - o.rs >= radius && o.se + (len - 1 - o.re) + radius < (int)seqs[o.seqIdx].cons.size() &&
          bn.matchCnt > 0.97 * (2 * len) && bs > repeatSim && m0 < 0.9 * bn.matchCnt) c = true;
      if (!c && m0 < 0.4 * bn.matchCnt) c = true;
      if (!c && n > 1000 && m0 < 0.9 * bn.matchCnt) c = true;
      if (c) { cut[(size_t)i] = 1; continue; }
    }
    const double so = simOf(o);
    if (!(so > 0)) continue;
    bool better = best == -1;
    if (!better) {   // _overlap::operator< with the scored fields (SeqSet.hpp:104-128)
      const t4_cand &b = cands[best];
      if (o.matchCnt != b.matchCnt) better = o.matchCnt > b.matchCnt;
      else if (so != bs) better = so > bs;
      else if (o.re - o.rs != b.re - b.rs) better = o.re - o.rs > b.re - b.rs;
      else if (o.seqIdx != b.seqIdx) better = o.seqIdx < b.seqIdx;
      else if ((o.flags & 1) != (b.flags & 1)) better = (o.flags & 1) < (b.flags & 1);
      else if (o.rs != b.rs) better = o.rs < b.rs;
      else if (o.re != b.re) better = o.re < b.re;
      else if (o.ss != b.ss) better = o.ss < b.ss;
      else better = o.se < b.se;
    }
    if (better) { best = i; bs = so; }
  }
}

// The statistics loop of SeqSet.hpp:784-811 repeated over an entry's groups as they stand: the query's own dependency records (every
// group with its emitted hits, in the reference's order -- minus strand by contig, then plus strand), the groups restricted re-queries
// and exactly booked index edits have added or resized since (the host's table), and -- pc >= 0 -- contig pc's two groups at the sizes
// pcSize[0 / 1] a re-query has just reported. Including the loop's `i = j; ++i` stepping: a group that follows a measured one is measured
// from its second hit, a one-hit group there vanishes. False when the entry's sizes are not exact (Cached::inexact, no device records).
// Out: novelMinHitRequired per strand (813-823), groups of >= 4 / >= 5 hits and the largest group (true sizes).
// ror (optional, entries with lists beyond 10000 postings): removeOnlyRepeats per strand -- some group holds three hits of shorter lists in
// its measured part (796-806). headTouched: one of c.editedKeys lies within the first headLen hits of the array in the reference's order.
bool t4_assembler::exactStats(const Cached &c, int pc, const int32_t *pcSize, int T[2], int e4[2], int e5[2], int big[2], bool *ror, int headLen, bool *headTouched) {
  if (!c.hasDev || c.inexact) return false;
  if (ror) { if (c.devInfo.size() != c.devGroups.size()) return false; ror[0] = ror[1] = false; }
  if (headTouched) *headTouched = false;
  struct Extra { uint32_t key; int cnt; };
  static thread_local std::vector<Extra> extra;   // groups that are not among the device records
  extra.clear();
  const Grp *d0 = c.devGroups.data(), *d1 = d0 + c.devGroups.size();
  for (const Grp &g : c.groups.t) if (g.key != 0xFFFFFFFFu && (pc < 0 || (g.key >> 1) != (uint32_t)pc)) extra.push_back(Extra{g.key, (int)g.cnt});
  if (pc >= 0)
    for (int t = 0; t < 2; ++t) {
      const uint32_t key = (uint32_t)pc * 2u + (uint32_t)t;
      const Grp *g = const_cast<Cached &>(c).findGroup(key);
      if (!(g && g >= d0 && g < d1)) extra.push_back(Extra{key, pcSize[t]});
    }
  std::sort(extra.begin(), extra.end(), [](const Extra &a, const Extra &b) { return (a.key & 1u) != (b.key & 1u) ? (a.key & 1u) < (b.key & 1u) : a.key < b.key; });
  int possible[2] = {0, 0}, longest[2] = {0, 0};
  e4[0] = e4[1] = e5[0] = e5[1] = big[0] = big[1] = 0;
  bool skip = false;
  size_t xe = 0;
  long long hitsBefore = 0;   // hits of the groups visited so far: the position of the next group's first hit in the array
  auto visit = [&](uint32_t key, int n, int info) {
    if (headTouched && hitsBefore < headLen && std::find(c.editedKeys.begin(), c.editedKeys.end(), key) != c.editedKeys.end()) *headTouched = true;
    if (n <= 0) return;
    const int plus = (int)(key & 1u);
    const int m = n - (skip ? 1 : 0);
    if (m > 0) {
      if (m > 3) ++possible[plus];
      if (m > longest[plus]) longest[plus] = m;
      if (ror && (info & 7) - ((skip && (info & 8)) ? 1 : 0) >= 3) ror[plus] = true;
    }
    skip = !(skip && n == 1);
    if (n >= 4) ++e4[plus];
    if (n >= 5) ++e5[plus];
    if (n > big[plus]) big[plus] = n;
    hitsBefore += n;
  };
  auto before = [](uint32_t a, uint32_t b) { return (a & 1u) != (b & 1u) ? (a & 1u) < (b & 1u) : a < b; };
  for (const Grp *g = d0; g < d1; ++g) {
    // (a group the host has added since holds fewer than three hits: no three of shorter lists, whatever they are)
    while (xe < extra.size() && before(extra[xe].key, g->key)) { visit(extra[xe].key, extra[xe].cnt, 0); ++xe; }
    visit(g->key, pc >= 0 && (g->key >> 1) == (uint32_t)pc ? pcSize[(int)(g->key & 1u)] : (int)g->cnt, ror ? (int)c.devInfo[(size_t)(g - d0)] : 0);
  }
  while (xe < extra.size()) { visit(extra[xe].key, extra[xe].cnt, 0); ++xe; }
  for (int t = 0; t < 2; ++t) {
    T[t] = 3;
    if (possible[t] > 100000) T[t] = (int)(longest[t] * 0.75);
    else if (possible[t] > 10000) T[t] = longest[t] / 2;
    else if (possible[t] > 1000) T[t] = longest[t] / 3;
    else if (possible[t] > 100) T[t] = longest[t] / 4;
  }
  return true;
}

// A restricted re-query of contig pc came back for an entry that holds its candidate list: nc[0 .. ncnt) are ALL overlaps of the read
// with pc (scored; ov / ex / rets are their result records, same order), s8 the true sizes of pc's two hit groups. Returns false
// when the entry needs its whole query again. Three things are settled here, in this order: the threshold novelMinHitRequired under
// the new group sizes (certified by bounds, or the statistics loop repeated exactly; equal to the entry's or HIGHER -- a lower one
// falls back), the candidate list (pc's candidates swapped, candidates of runs shorter than a raised threshold dropped), and the
// scan of the pre-filters over the new list (replayScan).
bool t4_assembler::mergeRestricted(Cached &c, int pc, int k2, const t4_overlap *ov, const t4_overlap *ex, const int32_t *rets, const t4_cand *nc, int ncnt, const int32_t *s8) {
  const int len = (int)c.read.size();
  for (int t = 0; t < ncnt; ++t) if (((nc[t].flags & 1) != 0) != c.strand0Plus) { ++candFallbackStrand; return false; }   // an overlap on the other strand: which strand is the best one's is open again
  // ---- group statistics (SeqSet.hpp:784-823): pc's groups went from g0 to g1 hits; every other group of three or more is as it was
  int n4lo[2], n4hi[2], n5lo[2], n5hi[2], smlo[2], smhi[2];
  bool needExact = false, exactStable = true;
  int newT[2] = {c.minT[0], c.minT[1]};
  for (int t = 0; t < 2; ++t) {
    const uint32_t key = (uint32_t)pc * 2u + (uint32_t)t;
    const Grp *g = c.findGroup(key);
    const int rec = g ? (int)g->cnt : 0;   // what the last query of this group found -- exactly (device records, or set by a restricted re-query) or as a superset (host-derived tables)
    const bool exact = (c.hasDev && g && g >= c.devGroups.data() && g < c.devGroups.data() + c.devGroups.size()) || std::find(c.exactKeys.begin(), c.exactKeys.end(), key) != c.exactKeys.end();
    const int g0hi = rec, g0lo = exact ? rec : 0, g1 = s8[4 + t];
    n4lo[t] = c.n4lo[t] - (g0hi >= 4 ? 1 : 0) + (g1 >= 4 ? 1 : 0); n4hi[t] = c.n4hi[t] - (g0lo >= 4 ? 1 : 0) + (g1 >= 4 ? 1 : 0);
    n5lo[t] = c.n5lo[t] - (g0hi >= 5 ? 1 : 0) + (g1 >= 5 ? 1 : 0); n5hi[t] = c.n5hi[t] - (g0lo >= 5 ? 1 : 0) + (g1 >= 5 ? 1 : 0);
    if (n4lo[t] < 0) n4lo[t] = 0;
    if (n5lo[t] < 0) n5lo[t] = 0;
    smhi[t] = c.smhi[t] > g1 ? c.smhi[t] : g1;
    if (g1 >= c.smhi[t]) smlo[t] = g1;                       // the new group is the largest
    else if (g0hi < c.smlo[t]) smlo[t] = c.smlo[t];          // pc never was the largest: the largest of the others stands
    else {                                                   // pc may have been the largest: the largest of the others, from the records when they are exact
      int others = 0;
      if (c.hasDev) { const size_t b = t ? c.devSplit : 0, e = t ? c.devGroups.size() : c.devSplit; for (size_t q = b; q < e; ++q) if (c.devGroups[q].key != key && (int)c.devGroups[q].cnt > others) others = (int)c.devGroups[q].cnt; }
      smlo[t] = others > g1 ? others : g1;
      if (smlo[t] > smhi[t]) smlo[t] = smhi[t];
    }
    // the certificate of overlapsFromKeys over the bounds: both corners give the threshold the entry's candidates were made with
    const int lo = n5lo[t], hi = n4hi[t];
    const int cLo = lo > 100000 ? 4 : lo > 10000 ? 3 : lo > 1000 ? 2 : lo > 100 ? 1 : 0;
    const int cHi = hi > 100000 ? 4 : hi > 10000 ? 3 : hi > 1000 ? 2 : hi > 100 ? 1 : 0;
    bool ok = cLo == cHi;
    int T = 3;
    if (ok && cLo > 0) {
      const int a = smlo[t] - 1 > 0 ? smlo[t] - 1 : 0, big = smhi[t];
      const int fa = cLo == 4 ? (int)(a * 0.75) : cLo == 3 ? a / 2 : cLo == 2 ? a / 3 : a / 4;
      const int fb = cLo == 4 ? (int)(big * 0.75) : cLo == 3 ? big / 2 : cLo == 2 ? big / 3 : big / 4;
      ok = fa == fb && fa >= 3;
      T = fb;
    }
    if (!ok || T < c.minT[t]) needExact = true;
    newT[t] = T;   // (certified; a RAISED threshold is served below: candidates chained from shorter runs leave the list)
  }
  if (needExact) {
    // The bounds do not pin the threshold (e.g. longestHits / 4 changes between the largest group and the largest group - 1). When the
    // entry's dependency records are the query's own (every group with its emitted hits, in the reference's order: minus strand by
    // contig, then plus strand) and no index edit of a small group has been tolerated since, the statistics loop of
    // SeqSet.hpp:784-811 is repeated exactly over them, with pc's groups at their new sizes -- including its `i = j; ++i` stepping
    // (a group that follows a measured one is measured from its second hit; a one-hit group there vanishes).
    int Tx[2], e4[2], e5[2], big[2];
    if (!exactStats(c, pc, s8 + 4, Tx, e4, e5, big)) { ++candFallbackStats; return false; }
    for (int t = 0; t < 2; ++t) {
      if (Tx[t] < c.minT[t]) { ++candFallbackStats; return false; }   // a LOWER threshold lets runs in that no record of this entry describes
      newT[t] = Tx[t];
      n4lo[t] = n4hi[t] = e4[t]; n5lo[t] = n5hi[t] = e5[t]; smlo[t] = smhi[t] = big[t];
    }
    ++candExactStats;
    // (is the threshold safe from edits of small groups from here on? the certificate again, over the exact numbers)
    for (int t = 0; t < 2 && exactStable; ++t) {
      const int lo = n5lo[t], hi = n4hi[t];
      const int cLo = lo > 100000 ? 4 : lo > 10000 ? 3 : lo > 1000 ? 2 : lo > 100 ? 1 : 0;
      const int cHi = hi > 100000 ? 4 : hi > 10000 ? 3 : hi > 1000 ? 2 : hi > 100 ? 1 : 0;
      bool ok = cLo == cHi;
      if (ok && cLo > 0) {
        const int a = smlo[t] - 1 > 0 ? smlo[t] - 1 : 0, bg = smhi[t];
        const int fa = cLo == 4 ? (int)(a * 0.75) : cLo == 3 ? a / 2 : cLo == 2 ? a / 3 : a / 4;
        const int fb = cLo == 4 ? (int)(bg * 0.75) : cLo == 3 ? bg / 2 : cLo == 2 ? bg / 3 : bg / 4;
        ok = fa == fb && fa >= 3;
      }
      exactStable = ok;
    }
  }
  // ---- the candidate list with pc's candidates swapped, in scan order: m0 desc, read span desc, contig, strand, geometry (operator< before scoring)
  struct Item { t4_cand o; int src; unsigned char wasCut; };   // src: index into nc, -1 for a kept candidate
  static thread_local std::vector<Item> items, fresh;
  items.clear(); fresh.clear();
  auto before = [](const t4_cand &a, const t4_cand &b) {
    if (a.m0 != b.m0) return a.m0 > b.m0;
    if (a.re - a.rs != b.re - b.rs) return a.re - a.rs > b.re - b.rs;
    if (a.seqIdx != b.seqIdx) return a.seqIdx < b.seqIdx;
    if ((a.flags & 1) != (b.flags & 1)) return (a.flags & 1) < (b.flags & 1);
    if (a.rs != b.rs) return a.rs < b.rs;
    if (a.re != b.re) return a.re < b.re;
    if (a.ss != b.ss) return a.ss < b.ss;
    return a.se < b.se;
  };
  // A raised novelMinHitRequired (the contig's group grew: a read that hangs over a contig end the commit before it extended) drops
  // exactly the candidates whose run holds fewer hits than the new threshold (SeqSet.hpp:923-925; every other test of a run is its
  // own): those of pc that the re-query -- run with the OLD threshold -- still returned, and those of the other contigs, whose
  // records leave the entry's result. Run sizes travel with the candidate records (bits 3-12 of the flags).
  const int s0 = c.strand0Plus ? 1 : 0;
  const int raisedT = newT[s0] > c.minT[s0] ? newT[s0] : 0;
  auto runOf = [](const t4_cand &o) { return (int)((o.flags >> 3) & 1023u); };
  static thread_local std::vector<t4_cand> droppedKept;
  droppedKept.clear();
  if (newT[s0] > 1023) { ++candFallbackStats; return false; }   // (the field saturates there)
  if (raisedT) for (const t4_cand &o : c.cands) if (runOf(o) == 0) { ++candFallbackStats; return false; }   // (a record without a run size: never from this engine's kernels)
  for (int t = 0; t < ncnt; ++t) if (runOf(nc[t]) == 0) { ++candFallbackStats; return false; }
  for (int t = 0; t < ncnt; ++t) {
    if (runOf(nc[t]) < newT[s0]) continue;   // (the re-query ran with the threshold of its launch, which an earlier merge of this entry may have raised since)
    Item it; it.o = nc[t]; it.o.flags &= (unsigned short)~4u; it.src = t; it.wasCut = 0; fresh.push_back(it);
  }
  std::sort(fresh.begin(), fresh.end(), [&](const Item &a, const Item &b) { return before(a.o, b.o); });
  {
    size_t f = 0;
    for (const t4_cand &o : c.cands) {
      if (o.seqIdx == pc) continue;
      if (raisedT && runOf(o) < raisedT) { if (!(o.flags & 4)) droppedKept.push_back(o); continue; }
      while (f < fresh.size() && before(fresh[f].o, o)) items.push_back(fresh[f++]);
      Item it; it.o = o; it.src = -1; it.wasCut = (o.flags & 4) ? 1 : 0; items.push_back(it);
    }
    while (f < fresh.size()) items.push_back(fresh[f++]);
  }
  static thread_local std::vector<t4_cand> list;
  static thread_local std::vector<unsigned char> cut;
  list.resize(items.size());
  for (size_t t = 0; t < items.size(); ++t) list[t] = items[t].o;
  replayScan(list, seqs, len, radius, 0.95, cut);
  auto simOf = [](const t4_cand &o) { return (o.flags & 2) ? 0.0 : (double)o.matchCnt / (double)(o.se - o.ss + 1 + o.re - o.rs + 1); };
  bool anyRecut = false;
  for (size_t t = 0; t < items.size(); ++t) {
    if (items[t].src >= 0) continue;
    if (items[t].wasCut && !cut[t] && !(simOf(items[t].o) < novelSim)) { ++candFallbackUncut; return false; }   // it passes now, and its ExtendOverlap record was never made
    if (!items[t].wasCut && cut[t]) anyRecut = true;
  }
  // ---- the entry's result records: those of pc and of candidates that are cut now leave, pc's survivors come in
  auto sameGeom = [](const t4_overlap &r, const t4_cand &o) { return r.seqIdx == o.seqIdx && r.readStart == o.rs && r.readEnd == o.re && r.seqStart == o.ss && r.seqEnd == o.se; };
  size_t w = 0;
  for (size_t t = 0; t < c.ov.size(); ++t) {
    bool keep = c.ov[t].seqIdx != pc;
    if (keep && anyRecut) for (size_t q = 0; q < items.size() && keep; ++q) if (items[q].src < 0 && !items[q].wasCut && cut[q] && sameGeom(c.ov[t], items[q].o)) keep = false;
    if (keep) for (size_t q = 0; q < droppedKept.size() && keep; ++q) if (sameGeom(c.ov[t], droppedKept[q])) keep = false;
    if (!keep) continue;
    if (w != t) { c.ov[w] = c.ov[t]; c.ext[w] = c.ext[t]; c.extRet[w] = c.extRet[t]; }
    ++w;
  }
  c.ov.resize(w); c.ext.resize(w); c.extRet.resize(w);
  for (size_t t = 0; t < items.size(); ++t) {
    const int j = items[t].src;
    if (j < 0 || cut[t]) continue;
    if (ov[j].similarity < novelSim) continue;   // the similarity cut (SeqSet.hpp:2105-2119)
    c.ov.push_back(ov[j]); c.ext.push_back(ex[j]); c.extRet.push_back(rets[j]);
  }
  (void)k2;
  if (anyRecut) ++candRecut;
  if (items.size() > 50) ++candMergesBig;
  if (n4hi[0] > 100 || n4hi[1] > 100) ++candMergesStats;
  c.cnt = (int32_t)c.ov.size();
  c.cands.resize(items.size());
  for (size_t t = 0; t < items.size(); ++t) { c.cands[t] = items[t].o; if (cut[t]) c.cands[t].flags |= 4; else c.cands[t].flags &= (unsigned short)~4u; }
  for (int t = 0; t < 2; ++t) { c.n4lo[t] = n4lo[t]; c.n4hi[t] = n4hi[t]; c.n5lo[t] = n5lo[t]; c.n5hi[t] = n5hi[t]; c.smlo[t] = smlo[t]; c.smhi[t] = smhi[t]; c.minT[t] = newT[t]; }
  if (raisedT) ++candRaised;
  c.nAll = c.nAllBound = (int)c.cands.size();
  // the dependency record of pc as the query found it: emitted hits per strand, hull of the diagonals with three or more of them
  for (int t = 0; t < 2; ++t) {
    const uint32_t key = (uint32_t)pc * 2u + (uint32_t)t;
    if (s8[4 + t] > 0 || c.findGroup(key)) { Grp &g = c.getGroup(key); g.cnt = (uint32_t)s8[4 + t]; g.lo = s8[8 + t]; g.hi = s8[10 + t]; }
    if (std::find(c.exactKeys.begin(), c.exactKeys.end(), key) == c.exactKeys.end()) c.exactKeys.push_back(key);
  }
  c.statsStable = exactStable && !knobs.noStableStats; c.n4 = c.n4hi[0] + c.n4hi[1]; c.slack = 99 - c.n4;
  ++candMerges;
  return true;
}

// The lane's job is over (waits for it if need be): entries that still stand take their records, with the left extensions that
// happened meanwhile applied; entries a commit killed in flight stay without a result.
int t4_assembler::harvest(Lane &L) {
  if (!L.busy) return T4_OK;
  auto th0_ = std::chrono::steady_clock::now();
  struct Th { double &acc; std::chrono::steady_clock::time_point t0; ~Th() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } th_{secHarvest, th0_};
  const int32_t *cnts = nullptr, *bas = nullptr, *rets = nullptr;
  const t4_overlap *ov = nullptr, *ex = nullptr;
  int rc;
  ts.lap(TS_PUMP_OTHER);
  {
    auto tq0 = std::chrono::steady_clock::now();
    rc = t4_add_query_pool_end(L.ctx, &cnts, &bas, &ov, &ex, &rets);
    secQuery += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count();
  }
  ts.lap(TS_HARVEST_WAIT);
  L.busy = false;
  const int m = (int)L.slots.size();
  if (!rc && lanes.size() == 1) {
    double ms_ = 0; (void)t4_add_query_last_call(L.ctx, &ms_, nullptr, nullptr);
    roundKernelMs.push_back((float)ms_);
    roundWallMs.push_back((float)(std::chrono::duration<double>(std::chrono::steady_clock::now() - laneT0).count() * 1e3));
  }
  if (rc) {
    if (L.ctx != ctx) err = t4_last_error(L.ctx);
    for (int i = 0; i < m; ++i) { Cached &c = *pool[L.slots[i]]; if (c.uid == L.uids[i]) { c.inflight = false; c.killed = false; } }
    dropWindow();
    return rc;
  }
  FILE *roundLog = knobs.roundLog;
  if (roundLog) {
    double ms = 0; const int32_t *ticks = nullptr; int nn = 0;
    t4_add_query_last_call(L.ctx, &ms, &ticks, &nn);
    fprintf(roundLog, "%lld %d %.4f %d |", (long long)rounds, m, ms, (int)seqs.size());
    for (int i = 0; i < m && i < nn; ++i) fprintf(roundLog, " %d/%d/%d/%d", ticks ? ticks[i] / 100 : -1, cnts[i], (int)L.hint[i], pool[L.slots[i]]->uid == L.uids[i] ? (int)pool[L.slots[i]]->killed : 2);
    fputc('\n', roundLog);
  }
  const int32_t *ticks10ns = nullptr; int nTicks = 0;
  { double ms_ = 0; (void)t4_add_query_last_call(L.ctx, &ms_, &ticks10ns, &nTicks); }
  const int32_t *stable = nullptr; int nStable = 0;
  const bool noStable = knobs.noStableStats;
  if (!noStable) (void)t4_add_query_last_stable(L.ctx, &stable, &nStable);
  const int32_t *aux = nullptr, *n4s = nullptr, *qstatus = nullptr; int nAux = 0;
  (void)t4_add_query_last_aux(L.ctx, &aux, &n4s, &qstatus, &nAux);
  const t4_cand *candPool = nullptr; const int32_t *candBase = nullptr, *candCnt = nullptr, *stats8 = nullptr; int nCand = 0;
  (void)t4_add_query_last_cands(L.ctx, &candPool, &candBase, &candCnt, &stats8, &nCand);
  // The dependency records of the reads the wide query served are the bulk of what a round hands back (thousands of 16-byte records
  // per read; 17 G of them over the first 2 M pairs of C3, where copying them was a third of the pass): when a round brings many,
  // the host threads copy them side by side before the entries are gone through one after the other.
  std::vector<unsigned char> groupsCopied((size_t)m, 0);
  if (threads > 1 && m >= 2) {
    std::vector<int> big; size_t total = 0;
    for (int i = 0; i < m; ++i) {
      if (L.only[i] >= 0) continue;
      const Cached &c = *pool[L.slots[i]];
      if (c.uid != L.uids[i] || !c.inflight || c.killed) continue;
      const t4_grp *dg = nullptr; int ng = 0;
      if (t4_add_query_groups(L.ctx, i, &dg, &ng, nullptr, nullptr) == 1 && ng >= 1024) { big.push_back(i); total += (size_t)ng; }
    }
    if (big.size() >= 2 && total >= 16384) {
      std::atomic<int> next(0);
      const std::function<void()> copier = [&]() {
        for (;;) {
          const int t = next.fetch_add(1);
          if (t >= (int)big.size()) break;
          const int i = big[(size_t)t];
          Cached &c = *pool[L.slots[i]];
          const t4_grp *dg = nullptr; int ng = 0;
          if (t4_add_query_groups(L.ctx, i, &dg, &ng, nullptr, nullptr) == 1) { c.devGroups.assign((const Grp *)dg, (const Grp *)dg + ng); groupsCopied[(size_t)i] = 1; }
        }
      };
      const int nHelp = (int)big.size() - 1 < threads - 1 ? (int)big.size() - 1 : threads - 1;
      if (!helpers) helpers.reset(new HelperPool());
      helpers->start(nHelp, copier);
      copier();
      helpers->wait();
    }
  }
  for (int i = 0; i < m; ++i) {
    Cached &c = *pool[L.slots[i]];
    if (c.uid != L.uids[i] || !c.inflight) continue;   // the entry was retired (or re-announced) meanwhile
    c.inflight = false;
    if (L.only[i] >= 0) {
      // ---- restricted re-queries come back: every overlap of the read with one contig, scored and extended -- one item per contig
      // the entry waited for when the launch went out (adjacent items)
      int j = i + 1;
      while (j < m && L.slots[j] == L.slots[i] && L.uids[j] == L.uids[i] && L.only[j] >= 0) ++j;
      const int i0 = i;
      i = j - 1;
      if (c.killed) { c.killed = false; ++killedInFlight; continue; }   // a contig it waits for changed again meanwhile (the entry stays partial), or the whole entry fell
      if (!c.partial) continue;
      bool fell = false;
      for (int q = i0; q < j && !fell; ++q) {
        const int pc = L.only[q];
        if (!c.isPending(pc)) continue;
        const int k2 = cnts[q] > 0 ? cnts[q] : 0;
        if (c.candOk) {   // the candidate store: swap the contig's candidates, repeat the scan, check the group statistics
          const bool fits = !(qstatus && q < nAux && qstatus[q] == 5) && candPool && candCnt && q < nCand && candCnt[q] == k2;
          if (fits && mergeRestricted(c, pc, k2, ov + bas[q], ex + bas[q], rets + bas[q], candPool + (k2 ? candBase[q] : 0), k2, stats8 + T4_QUERY_STATS * (size_t)q)) {
            ++c.restrictedCount; ++restrictedMerged;
          } else { fell = true; c.candOk = false; if (!fits) ++candFallbackOther; }
        } else {
          bool ok = !(qstatus && q < nAux && qstatus[q] == 5);
          for (int t = 0; ok && t < k2; ++t) if ((ov[bas[q] + t].strand == 1) != c.strand0Plus) ok = false;   // an overlap on the other strand: which strand is the best one's is open again
          c.nAllBound += k2;
          if (c.nAllBound > 50) ok = false;   // (the pre-filters of SeqSet.hpp:1705 may be on now)
          if (!ok) { fell = true; break; }
          size_t w = 0;
          for (size_t t = 0; t < c.ov.size(); ++t) if (c.ov[t].seqIdx != pc) { if (w != t) { c.ov[w] = c.ov[t]; c.ext[w] = c.ext[t]; c.extRet[w] = c.extRet[t]; } ++w; }
          c.ov.resize(w); c.ext.resize(w); c.extRet.resize(w);
          for (int t = 0; t < k2; ++t) {
            const t4_overlap &o = ov[bas[q] + t];
            if (o.similarity < novelSim) continue;   // the similarity cut (SeqSet.hpp:2105-2119)
            c.ov.push_back(o); c.ext.push_back(ex[bas[q] + t]); c.extRet.push_back(rets[bas[q] + t]);
          }
          c.cnt = (int32_t)c.ov.size();
          rebuildGroup(c, pc);
          ++c.restrictedCount; ++restrictedMerged;
        }
        if (!fell) {   // this contig is served
          if (c.pendingContig == pc) { if (c.morePending.empty()) c.pendingContig = -1; else { c.pendingContig = c.morePending.back(); c.morePending.pop_back(); } }
          else c.morePending.erase(std::find(c.morePending.begin(), c.morePending.end(), pc));
        }
      }
      ts.lap(TS_HARVEST_MERGE);
      if (fell) { c.partial = false; c.pendingContig = -1; c.morePending.clear(); c.lastKill = 7; ++restrictedFallbacks; continue; }   // neither valid nor partial: the whole query, next launch
      c.merged = true;
      if (c.pendingContig < 0) { c.partial = false; c.valid = true; }   // (else a contig joined while the launch was out: the entry goes on waiting for that one)
      continue;
    }
    c.tier = L.hint[i];
    if (ticks10ns && i < nTicks) c.lastUs = ticks10ns[i] / 100;
    c.statsStable = stable && i < nStable && stable[i] == 1;
    if (c.killed) { c.killed = false; c.shifts.clear(); ++killedInFlight; continue; }
    c.cnt = cnts[i];
    const int k2 = c.cnt > 0 ? c.cnt : 0;
    c.ov.assign(ov + bas[i], ov + bas[i] + k2);
    c.ext.assign(ex + bas[i], ex + bas[i] + k2);
    c.extRet.assign(rets + bas[i], rets + bas[i] + k2);
    for (const auto &sh : c.shifts) {
      for (t4_overlap &o : c.ov) if (o.seqIdx == sh.first) { o.seqStart += sh.second; o.seqEnd += sh.second; }
      for (t4_overlap &o : c.ext) if (o.seqIdx == sh.first) { o.seqStart += sh.second; o.seqEnd += sh.second; }
    }
    std::vector<std::pair<int, int>> shiftsOfCall; shiftsOfCall.swap(c.shifts);   // (the candidate records below take them too)
    if (aux && i < nAux && aux[i] >= 0) { c.nAll = aux[i] & 32767; c.nOther = (aux[i] >> 15) & 32767; c.strand0Plus = ((aux[i] >> 30) & 1) != 0; c.n4 = n4s[i]; c.nAllBound = c.nAll; c.auxOk = true; }
    else c.auxOk = false;
    c.cands.clear(); c.candOk = false; c.exactKeys.clear();
    if (candStore && c.auxOk && candPool && candCnt && i < nCand && candCnt[i] == c.nAll && c.nAll < 32767 && c.nOther == 0) {
      c.cands.assign(candPool + (c.nAll ? candBase[i] : 0), candPool + (c.nAll ? candBase[i] : 0) + c.nAll);
      for (const auto &sh : shiftsOfCall) for (t4_cand &o : c.cands) if (o.seqIdx == sh.first) { o.ss += sh.second; o.se += sh.second; }
      const int32_t *s8 = stats8 + T4_QUERY_STATS * (size_t)i;
      for (int t = 0; t < 2; ++t) { c.n4lo[t] = c.n4hi[t] = s8[t]; c.n5lo[t] = c.n5hi[t] = s8[2 + t]; c.smlo[t] = c.smhi[t] = s8[4 + t]; c.minT[t] = s8[6 + t] > 0 ? s8[6 + t] : 3; }
      c.candOk = true; c.toleratedSince = 0; c.inexact = false; candRecords += c.nAll;
      if (knobs.verifyWindow && c.cnt >= 0) {   // the host's scan against the kernel's: the same cuts, the same survivors of the similarity cut
        std::vector<unsigned char> cut;
        replayScan(c.cands, seqs, (int)c.read.size(), radius, 0.95, cut);
        int survivors = 0; bool same = true;
        for (size_t t = 0; t < c.cands.size(); ++t) {
          const t4_cand &o = c.cands[t];
          if ((cut[t] != 0) != ((o.flags & 4) != 0)) same = false;
          if (!cut[t] && !(o.flags & 2) && !((double)o.matchCnt / (double)(o.se - o.ss + 1 + o.re - o.rs + 1) < novelSim)) ++survivors;
        }
        ++candSelfChecks;
        if (!same || survivors != (c.cnt > 0 ? c.cnt : 0)) {
          fprintf(stderr, "T4_VERIFY_WINDOW: the host's replay of the pre-filter scan differs from the query's (entry %lld, %d candidates, %d survivors against %d records)\n", (long long)c.uid, c.nAll, survivors, c.cnt);
          err = "candidate scan mismatch"; return T4_ERR_STATE;
        }
      }
    }
    {
      const t4_grp *dg = nullptr; int ng = 0, huge = 0, n4 = 0;
      if (t4_add_query_groups(L.ctx, i, &dg, &ng, &huge, &n4) == 1) {
        static_assert(sizeof(t4_grp) == sizeof(Grp), "dependency record layout");
        if (!groupsCopied[(size_t)i]) c.devGroups.assign((const Grp *)dg, (const Grp *)dg + ng);
        c.devSplit = 0;
        c.devInfo.clear(); c.rorOk = false;
        if (huge) {   // (bits 24-27 of a record's count: T4_GRP_INFO_SHIFT in t4_wide.h)
          c.devInfo.resize(c.devGroups.size());
          for (size_t q = 0; q < c.devGroups.size(); ++q) { c.devInfo[q] = (uint8_t)((c.devGroups[q].cnt >> 24) & 15u); c.devGroups[q].cnt &= 0xFFFFFFu; }
        }
        { size_t lo = 0, hi = c.devGroups.size(); while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (c.devGroups[mid].key & 1u) hi = mid; else lo = mid + 1; } c.devSplit = lo; }
        if (knobs.verifyWindow && !c.dbgGroups.empty()) {   // the host's replay of GetHitsFromRead against the kernels' (both describe the same image)
          bool same = c.dbgGroups.size() == c.devGroups.size();
          for (size_t q = 0; same && q < c.devGroups.size(); ++q) {
            const Grp &a = c.dbgGroups[q], &b = c.devGroups[q];
            same = a.key == b.key && a.cnt == b.cnt && (a.lo > a.hi ? b.lo > b.hi : (a.lo == b.lo && a.hi == b.hi));
          }
          ++groupSelfChecks;
          if (!same) {
            fprintf(stderr, "T4_VERIFY_WINDOW: the host's dependency records of a read differ from the wide query's (entry %lld, %zu against %zu records)\n", (long long)c.uid, c.dbgGroups.size(), c.devGroups.size());
            for (size_t q = 0, shown = 0; q < c.devGroups.size() && q < c.dbgGroups.size() && shown < 6; ++q) {
              const Grp &a = c.dbgGroups[q], &b = c.devGroups[q];
              if (a.key != b.key || a.cnt != b.cnt || a.lo != b.lo || a.hi != b.hi) { fprintf(stderr, "  #%zu host key %u cnt %u hull %d..%d   device key %u cnt %u hull %d..%d\n", q, a.key, a.cnt, a.lo, a.hi, b.key, b.cnt, b.lo, b.hi); ++shown; }
            }
            err = "dependency records mismatch"; return T4_ERR_STATE;
          }
          c.dbgGroups.clear();
        }
        c.hasDev = true; c.hostRecords = false; c.groups.reset(16);
        c.slack = 99 - n4; c.fragile = huge != 0;
        c.inexact = false; c.editedKeys.clear();
        if (huge && c.candOk) {   // removeOnlyRepeats as this query found it, from its own records (exactStats repeats the loop the kernel ran)
          int T[2], e4[2], e5[2], big[2];
          c.rorOk = exactStats(c, -1, nullptr, T, e4, e5, big, c.ror0) && T[0] == c.minT[0] && T[1] == c.minT[1];
        }
        ++wideServed; wideGroupRecords += ng;
      } else if (c.expectWide) { buildGroups(c); ++wideMispredicted; }   // (the LDS tier served it after all)
      c.expectWide = false;
    }
    c.valid = true;
  }
  ts.lap(TS_HARVEST_COPY);
  return T4_OK;
}

// Keep the window's queries going. Finished lanes are harvested; entries near the head that have no result and no running query
// are sent out -- at once when the head itself needs one (it is what the caller waits for), else when a few have gathered and a
// lane beside the one kept for the head is idle. needHead: return only when the head entry holds a result.
int t4_assembler::pumpLive(bool needHead, int repetitive) {
  int rc;
  if ((rc = ensureLanes())) return rc;
  const int fixedAhead = knobs.queryAhead, minBatch = knobs.minBatch;
  // Entries far behind the head rarely survive until they are consumed: (re-)query only as far ahead as a few times what a launch
  // for the head has recently served (every read queried adds to the latency of the launch: it ends with its slowest read)
  const size_t ahead = fixedAhead > 0 ? (size_t)fixedAhead : (lanes.size() > 1 ? 24 : (size_t)(3.0 * runEma) + 12);   // (factors 2 and 4 measured in round 5 with light rounds: 66.4 / 64.1 s against 61.3 s on C2, profiles/r05f)
  const size_t restrictAhead = knobs.restrictAhead > 0 ? (size_t)knobs.restrictAhead : knobs.restrictAhead < 0 ? (size_t)(-knobs.restrictAhead * 0.5 * runEma) + 4 : 0;
  for (;;) {
    // T4_LIVE_HARVEST_DELAY=n (testing aid): a finished launch is only noticed n calls later, so that commits pile up against queries in flight
    const int harvestDelay = knobs.harvestDelay;
    for (Lane &L : lanes) if (L.busy && ++L.polls > harvestDelay && t4_add_query_pool_done(L.ctx)) { if ((rc = harvest(L))) return rc; }
    if (order.empty()) return T4_OK;
    if (needHead && lanes.size() == 1 && pool[order.front()]->valid) return T4_OK;   // (one lane: nothing is launched beside a head that holds its result)
    if (index.total == 0) {   // an empty set has no hit for anybody
      for (size_t i = 0; i < order.size() && i < ahead; ++i) {
        Cached &c = *pool[order[i]];
        if (c.valid || c.inflight) continue;
        c.cnt = 0; c.valid = true; c.groups.reset(16); c.slack = 99; c.fragile = false;
        if (!c.registered) registerKmers(c, order[i]);
      }
      return T4_OK;
    }
    Cached &head = *pool[order.front()];
    std::vector<int> light, heavy;
    for (size_t i = 0; i < order.size() && i < ahead; ++i) {
      Cached &c = *pool[order[i]];
      if (c.valid || c.inflight) continue;
      // two classes of work: restricted re-queries (one contig of an entry that keeps the rest: tens of microseconds) and whole
      // queries (hundreds; a read the wide query serves, more). With two lanes the head's restricted re-query does not wait for
      // the whole queries of the entries behind it.
      // (an entry that waits for a contig far behind the head is likely to meet another change of that contig before it is served -- the
      // reads of a clone extend the same contig end one after the other --: its re-query waits until it comes within restrictAhead places)
      if (c.partial && restrictAhead > 0 && i >= restrictAhead) continue;
      (c.partial ? light : heavy).push_back(order[i]);
    }
    const bool headWaits_ = !head.valid && !head.inflight;
    int idle = 0; Lane *free1 = nullptr;
    for (Lane &L : lanes) if (!L.busy) { ++idle; if (!free1) free1 = &L; }
    bool launched = false;
    if (headWaits_) {
      if (!free1) {   // every lane is busy: the first to finish serves the head
        for (Lane &L : lanes) if (L.busy) { if ((rc = harvest(L))) return rc; break; }
        continue;
      }
      // the head's launch carries the entries of its own weight class; the other class goes beside it when a lane is free
      std::vector<int> &mine = head.partial ? light : heavy, &other = head.partial ? heavy : light;
      if (lanes.size() == 1) {   // one launch: the heavy ones run beside the others on the ctx's second stream
        // A round lasts as long as its slowest read. When the head only waits for a restricted re-query (tens of microseconds), whole
        // queries of entries further back than `lightAhead` places stay out of its round: they go with the next round whose head needs
        // a whole query itself (or when they come within reach of the head).
        if (head.partial && knobs.lightAhead >= 0) {
          std::vector<int> near;
          for (size_t i = 0; i < order.size() && i < ahead && (int)i <= knobs.lightAhead; ++i) { Cached &c = *pool[order[i]]; if (!c.valid && !c.inflight && !c.partial) near.push_back(order[i]); }
          other.swap(near);
          ++lightRounds;
        }
        mine.insert(mine.end(), other.begin(), other.end()); other.clear();
      }
      {
        const double served = (double)(cacheHits - hitsAtLastRound);
        hitsAtLastRound = cacheHits;
        if (launchesUrgent > 0) runEma = 0.8 * runEma + 0.2 * served;
      }
      if (!head.partial) ++headWholeWhy[head.lastKill & 7];   // (what stalls the chain for a whole-query round)
      if ((rc = launchOn(*free1, mine, repetitive))) return rc;
      ++launchesUrgent; launched = true;
      if (!other.empty() && idle >= 2) for (Lane &L : lanes) if (!L.busy) { if ((rc = launchOn(L, other, repetitive))) return rc; break; }
    } else if ((int)(light.size() + heavy.size()) >= minBatch && idle >= 2) {   // one lane stays free for the head
      std::vector<int> all = light; all.insert(all.end(), heavy.begin(), heavy.end());
      if ((rc = launchOn(*free1, all, repetitive))) return rc;
      launched = true;
    }
    (void)launched;
    if (!needHead || pool[order.front()]->valid) return T4_OK;
    if (pool[order.front()]->inflight) {   // wait for the lane that carries the head
      auto tw0 = std::chrono::steady_clock::now();
      const int hs = order.front();
      for (Lane &L : lanes) {
        if (!L.busy) continue;
        bool has = false;
        for (size_t i = 0; i < L.slots.size(); ++i) if (L.slots[i] == hs && L.uids[i] == pool[hs]->uid) { has = true; break; }
        if (has) { if ((rc = harvest(L))) return rc; break; }
      }
      ++headWaits; secHeadWait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
    }
  }
}

int t4_assembler::prefetchLive(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive) {
  auto tp0_ = std::chrono::steady_clock::now();
  struct Tp { double &acc; std::chrono::steady_clock::time_point t0; ~Tp() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tp_{secPrefetch, tp0_};
  ts.begin();
  processEvents();
  // The driver announces before every round (whenever the head holds no result); going through up to 192 window entries to see that
  // they still line up was 12 us of every round. While the head lines up and the window is nearly full, the entries behind it are what
  // the calls before announced -- should the caller ever offer another read than it announced, the entry is caught when it is served
  // (addRead compares the head with the read in hand) -- so the window is topped up only when 32 or more places are free.
  bool skipAnnounce = false;
  if (n > 0 && !order.empty() && (int)order.size() + 32 >= n && (int)order.size() <= n) {
    const Cached &h = *pool[order.front()];
    skipAnnounce = h.strand == strands[0] && h.barcode == (barcodes ? barcodes[0] : -1) && h.skip == repetitive && h.read == reads[0];
  }
  if (!skipAnnounce) announceLive(n, reads, strands, barcodes, repetitive);
  ts.lap(TS_ANNOUNCE);
  const int rcPump = pumpLive(true, repetitive);
  ts.lap(TS_PUMP_OTHER);
  return rcPump;
}

void t4_assembler::beginWindow(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive) {
  dropWindow();
  cache.resize(n);
  for (int i = 0; i < n; ++i) {
    Cached &c = cache[i];
    c.read = reads[i]; c.strand = strands[i]; c.barcode = barcodes ? barcodes[i] : -1; c.skip = repetitive; c.cnt = 0; c.valid = true;
    c.ov.clear(); c.ext.clear(); c.extRet.clear();
  }
}

// results of the window's query (stride records per read)
void t4_assembler::endWindow(const int32_t *cnts, const t4_overlap *ov, const t4_overlap *ex, const int32_t *rets, int stride) {
  const int n = (int)cache.size();
  for (int q = 0; q < n; ++q) {
    Cached &c = cache[q];
    c.cnt = cnts ? cnts[q] : 0;
    int k2 = c.cnt > 0 ? c.cnt : 0;
    if (k2 > stride) k2 = stride;
    if (k2 > 0) {
      c.ov.assign(ov + (size_t)q * stride, ov + (size_t)q * stride + k2);
      c.ext.assign(ex + (size_t)q * stride, ex + (size_t)q * stride + k2);
      c.extRet.assign(rets + (size_t)q * stride, rets + (size_t)q * stride + k2);
    }
  }
}

// Query the GPU for n upcoming reads against the current set (one batch) and keep the results as the speculation window.
int t4_assembler::prefetch(int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive) {
  const int MAXOV = 128;
  int rc;
  if (owner) {   // a cell: its own reads only, through the owner's arena
    std::vector<t4_assembler *> who(n, this);
    return t4_cellset_prefetch(owner, n, who.data(), reads, strands, repetitive);
  }
  if (live()) return n > 0 ? prefetchLive(n, reads, strands, barcodes, repetitive) : T4_OK;
  if ((rc = refreshDevice())) return rc;
  auto tq0_ = std::chrono::steady_clock::now();
  struct Tq { double &acc; std::chrono::steady_clock::time_point t0; ~Tq() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tq_{secQuery, tq0_};
  beginWindow(n, reads, strands, barcodes, repetitive);
  std::string bases; std::vector<int64_t> offs(1, 0); std::vector<int32_t> bcs(n), sts(n); std::vector<double> fac(n);
  for (int i = 0; i < n; ++i) {
    const Cached &c = cache[i];
    bases += c.read; offs.push_back((int64_t)bases.size()); bcs[i] = c.barcode; sts[i] = c.strand;
    fac[i] = (c.barcode == -1 && !repetitive) ? 1.0 : 2.0;   // ExtendOverlap's mismatch factor (SeqSet.hpp:3597-3598)
  }
  if (bases.empty()) bases.push_back('A');
  const size_t m = (size_t)n * MAXOV;
  std::vector<t4_overlap> ov(m), ex(m);
  std::vector<int32_t> cnts(n), rets(m);
  rc = index.total == 0 ? T4_OK   // an empty set has no hit for anybody
                        : t4_add_query(dev, n, bases.data(), offs.data(), bcs.data(), sts.data(), repetitive, fac.data(), MAXOV, cnts.data(), ov.data(), ex.data(), rets.data());
  ++queries;
  if (rc) { dropWindow(); return rc; }
  endWindow(cnts.data(), ov.data(), ex.data(), rets.data(), MAXOV);
  return T4_OK;
}

// SeqSet::IsContigShallow (SeqSet.hpp:2512-2553): a base inside the covered stretch has fewer than minCov reads
bool t4_assembler::isContigShallow(int i, int minCov) const {
  const Seq &s = seqs[i];
  if (s.released) return false;
  const int len = (int)s.cons.size();
  auto sum = [&](int j) { return s.pw[j].c[0] + s.pw[j].c[1] + s.pw[j].c[2] + s.pw[j].c[3]; };
  int start, end, j;
  for (j = 0; j < len; ++j) if (sum(j) >= minCov) break;
  start = j;
  for (j = len - 1; j >= start; --j) if (sum(j) >= minCov) break;
  end = j;
  for (j = start; j <= end; ++j) if (sum(j) < minCov) break;
  return j <= end || end < start;
}
// SeqSet::ReleaseShallowContigs (SeqSet.hpp:10926-10936): at the end of the run, the index is not touched any more
void t4_assembler::releaseShallowContigs(int minCov) {
  for (int i = 0; i < (int)seqs.size(); ++i) if (isContigShallow(i, minCov)) { seqs[i].released = true; dirty = true; }
}

// SeqSet::ReleaseFinishedBarcodeSeq({barcode}, removeFromIndex = true, contigMinCov, earlyStop = true)
// (SeqSet.hpp:10815-10935). The posWeight "compression" there is lossless for Output, so the counts are kept as they are.
int t4_assembler::releaseFinishedBarcode(int barcode, int contigMinCov) {
  for (int i = (int)seqs.size() - 1; i >= 0; --i) {
    Seq &s = seqs[i];
    if (s.released) continue;
    if (s.frozen || s.pw.empty()) break;
    if (s.barcode != barcode) break;
    if (contigMinCov > 0 && isContigShallow(i, contigMinCov)) {   // ReleaseSeq after taking it out of the index
      index.removeSeq(s.cons.c_str(), (int)s.cons.size(), i, s.barcode, 0);
      s.released = true;
      structuralChange(i);
      continue;
    }
    s.frozen = true;   // before UpdateConsensus as in the reference: index = false, then the consensus is settled without re-indexing
    index.removeSeq(s.cons.c_str(), (int)s.cons.size(), i, s.barcode, 0);
    s.frozen = false; updateConsensus(i, false); s.frozen = true;
    structuralChange(i);
  }
  return T4_OK;
}

// SeqSet::Output (SeqSet.hpp:10939-10994): records of this set with ids shifted by idBase; barcodeName == nullptr prints
// the `>assemble<id>` form
static void writeRecords(FILE *fp, const std::vector<Seq> &seqs, int idBase, const char *barcodeName) {
  for (int i = 0; i < (int)seqs.size(); ++i) {
    const Seq &s = seqs[i];
    if (s.released) continue;
    if (barcodeName) fprintf(fp, ">%s_%d %s\n%s\n", barcodeName, idBase + i, s.name.c_str(), s.cons.c_str());
    else fprintf(fp, ">assemble%d %s\n%s\n", idBase + i, s.name.c_str(), s.cons.c_str());
    // the four count lines ("%d " per column, SeqSet.hpp:10962): digits written into one buffer per line -- a printf per number was
    // most of the output phase (50 M calls for the 25 k contigs of a 1 M-pair barcode run)
    std::string line;
    for (int c = 0; c < 4; ++c) {
      line.clear();
      line.reserve(s.cons.size() * 4 + 2);
      char tmp[16];
      for (size_t j = 0; j < s.cons.size(); ++j) {
        int v = s.pw[j].c[c];
        if (v >= 0 && v < 10) { line.push_back((char)('0' + v)); line.push_back(' '); continue; }
        unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
        int n = 0;
        do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) line.push_back('-');
        while (n) line.push_back(tmp[--n]);
        line.push_back(' ');
      }
      line.push_back('\n');
      fwrite(line.data(), 1, line.size(), fp);
    }
  }
}
int t4_assembler::output(const char *path) const {
  FILE *fp = fopen(path, "w");
  if (!fp) return T4_ERR_IO;
  writeRecords(fp, seqs, 0, nullptr);
  fclose(fp);
  return T4_OK;
}

extern "C" {

int t4_assembler_create(t4_ctx *ctx, int kmer_length, int consider_barcode, t4_assembler **out) {
  if (!ctx || !out || kmer_length < 2 || kmer_length > 31) return T4_ERR_ARG;
  t4_assembler *a = new t4_assembler(ctx, kmer_length);
  a->index.considerBarcode = consider_barcode != 0;
  a->index.mirror = a->live();
  *out = a;
  return T4_OK;
}
void t4_assembler_destroy(t4_assembler *a) {
  if (!a) return;
  if (a->dev) t4_index_destroy(a->dev);
  if (a->owner) return;   // cells belong to their t4_cellset
  delete a;
}
int t4_assembler_set_params(t4_assembler *a, int hit_len_required, int radius, double novel_seq_similarity) {
  if (!a) return T4_ERR_ARG;
  a->hitLenRequired = hit_len_required; a->radius = radius; a->novelSim = novel_seq_similarity; a->dirty = true;
  return T4_OK;
}
int t4_assembler_input_novel_read(t4_assembler *a, const char *id, const char *read, int strand, int barcode) {
  if (!a || !id || !read) return T4_ERR_ARG - 100;
  a->ts.begin();
  const int r = a->inputNovelRead(id, read, strand, barcode);
  a->ts.lap(TS_NOVEL);
  a->processEvents();
  return r;
}
int t4_assembler_add_read(t4_assembler *a, const char *read, const char *gene_name, int *strand, int barcode, int min_kmer_count,
                          int repetitive_data, double similarity_threshold) {
  if (!a || !read || !gene_name || !strand) return T4_ERR_ARG - 100;
  auto t0 = std::chrono::steady_clock::now();
  a->ts.begin();
  const int r = a->addRead(read, gene_name, strand, barcode, min_kmer_count, repetitive_data != 0, similarity_threshold);
  a->ts.lap(TS_ADD_DECIDE);   // (the early returns: no overlap, no candidate of the gene family)
  a->processEvents();
  a->secAddTotal += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return r;
}
int t4_assembler_prefetch(t4_assembler *a, int n, const char *const *reads, const int *strands, const int *barcodes, int repetitive_data) {
  if (!a || n < 0 || (n > 0 && (!reads || !strands))) return T4_ERR_ARG;
  return a->prefetch(n, reads, strands, barcodes, repetitive_data ? 1 : 0);
}
int t4_assembler_window_valid(const t4_assembler *a) {
  if (!a) return 0;
  if (a->live()) return a->idxEvents.empty() && a->structEvents.empty() && !a->order.empty() && a->pool[a->order.front()]->valid;
  return a->cacheHead < a->cache.size() && a->cache[a->cacheHead].valid;
}
int t4_assembler_counters(const t4_assembler *a, int64_t *queries, int64_t *refreshes, int64_t *window_hits) {
  if (!a) return T4_ERR_ARG;
  if (queries) *queries = a->queries;
  if (refreshes) *refreshes = a->refreshes;
  if (window_hits) *window_hits = a->cacheHits;
  return T4_OK;
}
int t4_assembler_timers(const t4_assembler *a, double *sec_refresh, double *sec_query) {
  if (!a) return T4_ERR_ARG;
  if (sec_refresh) *sec_refresh = a->secRefresh;
  if (sec_query) *sec_query = a->secQuery;
  return T4_OK;
}
int t4_assembler_repeat_add_read(t4_assembler *a, const char *read) {
  if (!a) return T4_ERR_ARG - 100;
  a->ts.begin();
  const int r = a->repeatAddRead(read);
  a->ts.lap(TS_REPEAT);
  a->processEvents();
  return r;
}
int t4_assembler_update_all_consensus(t4_assembler *a) { if (!a) return T4_ERR_ARG; a->ts.begin(); a->updateAllConsensus(); a->ts.lap(TS_UPDATE_CONS); a->processEvents(); return T4_OK; }
int t4_assembler_set_threads(t4_assembler *a, int host_threads) { if (!a || host_threads < 1) return T4_ERR_ARG; a->threads = host_threads > 64 ? 64 : host_threads; return T4_OK; }
int t4_assembler_live_counters(const t4_assembler *a, int64_t *out, int n) {
  if (!a || !out || n < 1) return T4_ERR_ARG;
  const int64_t v[16] = {a->rounds, a->readsQueried, a->deltas, a->deltaBytes, a->invalidations, a->invKey, a->invCross, a->invRegion, a->invShift,
                         a->invContig, a->invFragile, a->tolerated, (int64_t)(a->secDelta * 1e6), (int64_t)(a->secGroups * 1e6), (int64_t)(a->secEvents * 1e6), (int64_t)(a->secQuery * 1e6)};
  for (int i = 0; i < n && i < 16; ++i) out[i] = v[i];
  if (n >= 23) t4_add_query_stats(a->ctx, out + 16);
  if (n >= 27) t4_add_query_wide_stats(a->ctx, out + 23);   // the wide query: reads it served, partitions, calls repeated with larger pools, dependency records
  if (getenv("T4_VERIFY_WINDOW")) fprintf(stderr, "T4_VERIFY_WINDOW: the host's replay of the emitted hits equals the wide query's dependency records for all %lld reads it was held against\n", (long long)a->groupSelfChecks);
  if (getenv("T4_VERIFY_WINDOW")) fprintf(stderr, "T4_VERIFY_WINDOW: %lld served window entries queried again at serve time, all equal to their cached results (%lld of them put together from restricted re-queries)\n", (long long)a->verified, (long long)a->verifiedMerged);
  if (getenv("T4_TIMING")) {
    fprintf(stderr, "timing: assembler host seconds: add_read calls %.3f (incl. waits for the head), prefetch calls %.3f; launching %.3f (of which deltas %.3f, dependency sets %.3f, registering k-mers %.3f), harvesting %.3f, event examination %.3f, index edits %.3f\n",
            a->secAddTotal, a->secPrefetch, a->secLaunch, a->secDelta, a->secGroups, a->secRegister, a->secHarvest, a->secEvents, a->index.secOps);
    {
      const double spt = a->ts.secondsPerTick();
      fprintf(stderr, "timing: the chain's host thread by section (seconds):");
      for (int i = 0; i < TS_N; ++i) fprintf(stderr, " %s %.3f%s", TS_NAMES[i], (double)a->ts.acc[i] * spt, i + 1 < TS_N ? ";" : "\n");
      fprintf(stderr, "timing: host index: %lld postings found through their hint, %lld by walking a list (%lld postings walked over)\n", (long long)a->index.hintHits, (long long)a->index.hintWalks, (long long)a->index.walkSteps);
    }
    fprintf(stderr, "timing: query lanes %d: %lld launches (%lld for a head without a result, %lld of them without the whole queries of entries further back), %lld waits for the head's lane in %.3f s, %lld queries killed in flight\n",
            (int)a->lanes.size(), (long long)a->launches, (long long)a->launchesUrgent, (long long)a->lightRounds, (long long)a->headWaits, a->secHeadWait, (long long)a->killedInFlight);
    fprintf(stderr, "timing: rounds whose head waited for a WHOLE query: %lld never queried before, fallen to a change of: a key %lld, a list crossing 100 / 10000 postings %lld, a region %lld, a shift %lld, a whole contig %lld, the tolerance budget %lld, a restricted re-query that fell back %lld\n",
            (long long)a->headWholeWhy[0], (long long)a->headWholeWhy[1], (long long)a->headWholeWhy[2], (long long)a->headWholeWhy[3], (long long)a->headWholeWhy[4], (long long)a->headWholeWhy[5], (long long)a->headWholeWhy[6], (long long)a->headWholeWhy[7]);
    fprintf(stderr, "timing: thresholds checked after an edit of a small group by repeating the statistics loop: %lld entries, %lld of them fell (%lld of the checks for reads with lists beyond 10000 postings: removeOnlyRepeats and the head of the hit array as well)\n", (long long)a->toleranceChecks, (long long)a->toleranceCheckKills, (long long)a->fragileChecks);
    fprintf(stderr, "timing: tolerated index edits %lld, of which %lld met an entry whose group statistics cannot move (no budget spent); tolerance kills %lld, of which %lld for lists beyond 10000 postings\n",
            (long long)a->tolerated, (long long)a->toleratedStable, (long long)a->invFragile, (long long)a->invLongLists);
    fprintf(stderr, "timing: restricted re-queries: %lld entries kept their other contigs when one contig changed, %lld merged, %lld fell back to the whole query, %lld in flight met another change of their contig\n",
            (long long)a->restrictedMarks, (long long)a->restrictedMerged, (long long)a->restrictedFallbacks, (long long)a->restrictedStale);
    fprintf(stderr, "timing: entries that fell whole when one contig changed: %lld with lists beyond 10000 postings, %lld with overlaps on the other strand, %lld with more than 44 candidate overlaps, %lld with ~100 groups of four hits, %lld without the query's report, %lld other\n",
            (long long)a->whyNot[0], (long long)a->whyNot[1], (long long)a->whyNot[2], (long long)a->whyNot[3], (long long)a->whyNot[4], (long long)a->whyNot[5]);
    fprintf(stderr, "timing: candidate store: %lld candidate records kept with whole queries, %lld restricted re-queries merged through the replay of the scan (%lld with more than 50 candidates, %lld with more than 100 groups of four hits on a strand, %lld cut a candidate of another contig), fell back to the whole query: %lld a cut candidate of another contig passes now, %lld the group statistics could move the threshold, %lld an overlap on the other strand, %lld other; %lld whole queries checked against the host's scan; %lld thresholds settled by repeating the statistics loop over the entry's groups, %lld raised thresholds served by dropping the candidates of shorter runs\n",
            (long long)a->candRecords, (long long)a->candMerges, (long long)a->candMergesBig, (long long)a->candMergesStats, (long long)a->candRecut, (long long)a->candFallbackUncut, (long long)a->candFallbackStats, (long long)a->candFallbackStrand, (long long)a->candFallbackOther, (long long)a->candSelfChecks, (long long)a->candExactStats, (long long)a->candRaised);
    fprintf(stderr, "timing: wide query served %lld window entries (%lld dependency records came back with them), %lld reads it was expected for stayed on the LDS tier\n", (long long)a->wideServed, (long long)a->wideGroupRecords, (long long)a->wideMispredicted);
  }
  return T4_OK;
}
int t4_assembler_chain_stats(const t4_assembler *a, double *out, int n) {
  if (!a || !out || n < 1) return T4_ERR_ARG;
  auto pct = [](std::vector<float> v, double p) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return (double)v[(size_t)(p * (double)(v.size() - 1))]; };
  const double v[10] = {(double)a->rounds, (double)a->restrictedOnlyRounds, pct(a->roundKernelMs, 0.05), pct(a->roundKernelMs, 0.5), pct(a->roundWallMs, 0.05), pct(a->roundWallMs, 0.5),
                        (double)a->wholeQueries, (double)a->restrictedQueries, (double)a->candRecords, (double)a->candMerges};
  for (int i = 0; i < n && i < 10; ++i) out[i] = v[i];
  return T4_OK;
}
int t4_assembler_output(t4_assembler *a, const char *path) { return a ? a->output(path) : T4_ERR_ARG; }
int t4_assembler_size(const t4_assembler *a) { return a ? (int)a->seqs.size() : 0; }
int t4_assembler_contig(const t4_assembler *a, int i, t4_contig_view *out) {
  if (!a || !out || i < 0 || i >= (int)a->seqs.size()) return T4_ERR_ARG;
  const Seq &s = a->seqs[i];
  memset(out, 0, sizeof *out);
  out->name = s.name.c_str();
  out->consensus = s.released ? nullptr : s.cons.c_str();
  out->posweight = s.released || s.pw.empty() ? nullptr : (const int32_t *)s.pw.data();
  out->len = s.released ? 0 : (int32_t)s.cons.size(); out->barcode = s.barcode; out->num_read = s.numRead;
  out->min_left_ext_anchor = s.minLeftExtAnchor; out->min_right_ext_anchor = s.minRightExtAnchor; out->in_index = s.frozen ? 0 : 1;
  return T4_OK;
}
int t4_assembler_change_kmer_length(t4_assembler *a, int kmer_length) {
  if (!a || kmer_length < 2 || kmer_length > 31) return T4_ERR_ARG;
  return a->changeKmerLength(kmer_length);
}
int64_t t4_assembler_index_postings(const t4_assembler *a) { return a ? (int64_t)a->index.total : 0; }
int t4_assembler_release_finished_barcode(t4_assembler *a, int barcode, int contig_min_cov) {
  if (!a) return T4_ERR_ARG;
  const int r = a->releaseFinishedBarcode(barcode, contig_min_cov);
  a->processEvents();   // the window of a live set sees the removals before its next entry is served
  return r;
}
int t4_assembler_release_shallow_contigs(t4_assembler *a, int min_cov) { if (!a) return T4_ERR_ARG; a->releaseShallowContigs(min_cov); a->processEvents(); return T4_OK; }
int t4_assembler_output_barcodes(t4_assembler *a, const char *path, const char *const *barcode_names, int n_names) {
  if (!a || !path) return T4_ERR_ARG;
  FILE *fp = fopen(path, "w");
  if (!fp) return T4_ERR_IO;
  // SeqSet::Output(fp, &barcodeIntToStr) of one set that holds several barcodes (--keepNoBarcode): the header depends on the record
  for (int i = 0; i < (int)a->seqs.size(); ++i) {
    const Seq &q = a->seqs[i];
    if (q.released) continue;
    std::vector<Seq> one(1, q);
    const char *nm = (barcode_names && q.barcode >= 0 && q.barcode < n_names) ? barcode_names[q.barcode] : nullptr;
    writeRecords(fp, one, i, nm);
  }
  fclose(fp);
  return T4_OK;
}

// ---- t4_cellset ----------------------------------------------------------------------------------------------
int t4_cellset_create(t4_ctx *ctx, int kmer_length, t4_cellset **out) {
  if (!ctx || !out || kmer_length < 2 || kmer_length > 31) return T4_ERR_ARG;
  t4_cellset *cs = new t4_cellset();
  cs->ctx = ctx; cs->k = kmer_length;
  int r = t4_cellstore_create(ctx, kmer_length, &cs->store);
  if (r) { delete cs; return r; }
  *out = cs;
  return T4_OK;
}
void t4_cellset_destroy(t4_cellset *cs) {
  if (!cs) return;
  for (auto &kv : cs->cells) delete kv.second;
  t4_cellstore_destroy(cs->store);
  delete cs;
}
int t4_cellset_set_params(t4_cellset *cs, int hit_len_required, int radius, double novel_seq_similarity) {
  if (!cs) return T4_ERR_ARG;
  cs->hitLenRequired = hit_len_required; cs->radius = radius; cs->novelSim = novel_seq_similarity;
  for (auto &kv : cs->cells) { kv.second->hitLenRequired = hit_len_required; kv.second->radius = radius; kv.second->novelSim = novel_seq_similarity; kv.second->dirty = true; }
  return t4_cellstore_set_params(cs->store, hit_len_required, radius, novel_seq_similarity);
}
int t4_cellset_cell(t4_cellset *cs, int barcode, t4_assembler **cell) {
  if (!cs || !cell || barcode < 0 || barcode >= 1000003) return T4_ERR_ARG;   // beyond that, barcodes share index lists (KmerIndex.hpp:29-33)
  auto it = cs->cells.find(barcode);
  if (it == cs->cells.end()) {
    t4_assembler *a = new t4_assembler(cs->ctx, cs->k);
    a->index.considerBarcode = true;
    a->owner = cs; a->cellBarcode = barcode;
    a->hitLenRequired = cs->hitLenRequired; a->radius = cs->radius; a->novelSim = cs->novelSim;
    it = cs->cells.emplace(barcode, a).first;
  }
  *cell = it->second;
  return T4_OK;
}
int t4_cellset_close_cell(t4_cellset *cs, t4_assembler *cell) {   // no further queries for this cell: its arena slot is recycled
  if (!cs || !cell || cell->owner != cs) return T4_ERR_ARG;
  cell->dropWindow();
  if (cell->slot >= 0) { int r = t4_cellstore_close(cs->store, cell->slot); cell->slot = -1; cell->dirty = true; return r; }
  return T4_OK;
}

// One query batch for reads of many cells: read i is the next read cells[i] will be offered (a cell may contribute several
// consecutive reads, in the order they will come: that is the cell's speculation window).
int t4_cellset_prefetch(t4_cellset *cs, int n, t4_assembler *const *cells, const char *const *reads, const int *strands, int repetitive_data) {
  if (!cs || n < 0 || (n > 0 && (!cells || !reads || !strands))) return T4_ERR_ARG;
  const int MAXOV = 128;
  const int rep = repetitive_data ? 1 : 0;
  ++cs->batchSeq;
  // the reads of one cell must be consecutive (its speculation window, in the order they will be offered)
  int rc;
  struct Span { t4_assembler *cell; int begin, end, first; bool queried; int err; };
  std::vector<Span> spans;
  for (int i = 0; i < n;) {
    t4_assembler *cell = cells[i];
    if (!cell || cell->owner != cs) return T4_ERR_ARG;
    int j = i;
    while (j < n && cells[j] == cell) ++j;
    if (cell->windowStamp == cs->batchSeq) return T4_ERR_ARG;   // the cell appeared earlier in this batch
    cell->windowStamp = cs->batchSeq;
    spans.push_back(Span{cell, i, j, 0, cell->index.total > 0, 0});   // an empty index has no hit for anybody: no launch needed
    i = j;
  }
  // serial: slots, batch positions and the staging reservation of every image that has to be rebuilt
  auto t0 = std::chrono::steady_clock::now();
  int m = 0, maxSlot = -1;
  size_t stageBytes = 0;
  for (Span &sp : spans) {
    if (!sp.queried) continue;
    sp.first = m; m += sp.end - sp.begin;
    t4_assembler *cell = sp.cell;
    if (cell->slot < 0 && (rc = t4_cellstore_open(cs->store, &cell->slot))) return rc;
    if (cell->slot > maxSlot) maxSlot = cell->slot;
    if (!cell->dirty && !cell->patches.empty()) {
      std::vector<int64_t> offs; std::vector<unsigned char> vals;
      for (const t4_assembler::PwPatch &pp : cell->patches) { offs.push_back(cell->imgPwOff[pp.seq] + pp.pos); vals.push_back(pp.val); }
      if ((rc = t4_cellstore_patch(cs->store, cell->slot, (int)offs.size(), offs.data(), vals.data()))) return rc;
      cell->patches.clear();
    }
    if (cell->dirty) {
      int64_t consBytes = 0;
      for (const Seq &q : cell->seqs) consBytes += (q.released ? 0 : (int64_t)q.cons.size()) + 1;
      stageBytes += t4_cellstore_image_bytes((int)cell->seqs.size(), (int64_t)cell->index.map.size(), (int64_t)cell->index.total, consBytes);
    }
  }
  if ((rc = t4_cellstore_prepare(cs->store, maxSlot, stageBytes))) return rc;
  // parallel over cells: window slots + image builds
  parallelFor((int)spans.size(), cs->threads, [&](int si) {
    Span &sp = spans[si];
    t4_assembler *cell = sp.cell;
    const int cnt = sp.end - sp.begin;
    std::vector<int> bc(cnt, cell->cellBarcode);
    cell->beginWindow(cnt, reads + sp.begin, strands + sp.begin, bc.data(), rep);
    if (sp.queried) sp.err = cell->stageImage();
  });
  for (Span &sp : spans) if (sp.err) { cs->err = t4_last_error(cs->ctx); for (Span &q : spans) q.cell->dropWindow(); return sp.err; }
  cs->secStage += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::string bases; std::vector<int64_t> offs(1, 0); std::vector<int32_t> bcs, sts, slots; std::vector<double> fac;
  bcs.reserve(m); sts.reserve(m); slots.reserve(m); fac.reserve(m); offs.reserve(m + 1);
  for (Span &sp : spans) {
    if (!sp.queried) continue;
    for (int t = sp.begin; t < sp.end; ++t) {
      bases += reads[t]; offs.push_back((int64_t)bases.size()); bcs.push_back(sp.cell->cellBarcode); sts.push_back(strands[t]);
      slots.push_back(sp.cell->slot); fac.push_back(2.0);   // ExtendOverlap's mismatch factor with a barcode (SeqSet.hpp:3597-3598)
    }
  }
  // result buffers persist across batches and are never value-initialised (only the records that exist are written and read)
  if ((size_t)m * MAXOV > cs->resCap) {
    cs->resCap = (size_t)m * MAXOV * 2;
    cs->resOv.reset(new t4_overlap[cs->resCap]); cs->resEx.reset(new t4_overlap[cs->resCap]); cs->resRet.reset(new int32_t[cs->resCap]);
  }
  t4_overlap *ovp = cs->resOv.get(), *exp = cs->resEx.get();
  int32_t *retp = cs->resRet.get();
  std::vector<int32_t> cnts(m);
  if (m > 0) {
    auto tq = std::chrono::steady_clock::now();
    rc = t4_cellstore_query(cs->store, m, slots.data(), bases.data(), offs.data(), bcs.data(), sts.data(), rep, fac.data(), MAXOV,
                            cnts.data(), ovp, exp, retp);
    cs->secQuery += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq).count();
    cs->readsQueried += m;
    if (rc) { for (Span &sp : spans) sp.cell->dropWindow(); return rc; }
  }
  ++cs->queries;
  parallelFor((int)spans.size(), cs->threads, [&](int si) {
    Span &sp = spans[si];
    if (sp.queried) { sp.cell->endWindow(cnts.data() + sp.first, ovp + (size_t)sp.first * MAXOV, exp + (size_t)sp.first * MAXOV, retp + (size_t)sp.first * MAXOV, MAXOV); ++sp.cell->queries; }
    else sp.cell->endWindow(nullptr, nullptr, nullptr, nullptr, MAXOV);
  });
  return T4_OK;
}

// SeqSet::Output(fp, &barcodeIntToStr) of the set the reference would hold: cells in barcode order, contig ids numbered in
// creation order across the cells (contig slots are only ever created by InputNovelRead; merges reuse a slot)
int t4_cellset_output(t4_cellset *cs, const char *path, const char *const *barcode_names, int n_names) {
  return t4_cellset_output_at(cs, path, barcode_names, n_names, 0, 0);
}
// ... of a set that holds a contiguous range of the cells: ids start at id_base (the contig slots of the sets before it), and with
// `append` the records follow what the file holds (the driver's cell groups write one file, group after group)
int t4_cellset_output_at(t4_cellset *cs, const char *path, const char *const *barcode_names, int n_names, int id_base, int append) {
  if (!cs || !path || id_base < 0) return T4_ERR_ARG;
  FILE *fp = fopen(path, append ? "a" : "w");
  if (!fp) return T4_ERR_IO;
  int base = id_base;
  for (auto &kv : cs->cells) {
    const char *nm = (barcode_names && kv.first >= 0 && kv.first < n_names) ? barcode_names[kv.first] : nullptr;
    writeRecords(fp, kv.second->seqs, base, nm);
    base += (int)kv.second->seqs.size();
  }
  fclose(fp);
  return T4_OK;
}
int t4_cellset_set_threads(t4_cellset *cs, int host_threads) {
  if (!cs || host_threads < 1) return T4_ERR_ARG;
  cs->threads = host_threads > 64 ? 64 : host_threads;
  return T4_OK;
}
int t4_cellset_release_shallow_contigs(t4_cellset *cs, int min_cov) {
  if (!cs) return T4_ERR_ARG;
  for (auto &kv : cs->cells) kv.second->releaseShallowContigs(min_cov);
  return T4_OK;
}
int t4_cellset_size(const t4_cellset *cs) {
  if (!cs) return 0;
  int n = 0;
  for (auto &kv : cs->cells) n += (int)kv.second->seqs.size();
  return n;
}
int t4_cellset_update_all_consensus(t4_cellset *cs) {
  if (!cs) return T4_ERR_ARG;
  for (auto &kv : cs->cells) kv.second->updateAllConsensus();
  return T4_OK;
}
int t4_cellset_counters(const t4_cellset *cs, int64_t *query_batches, int64_t *reads_queried, int64_t *images_staged, int64_t *bytes_staged,
                        double *sec_query, double *sec_stage) {
  if (!cs) return T4_ERR_ARG;
  if (query_batches) *query_batches = cs->queries;
  if (reads_queried) *reads_queried = cs->readsQueried;
  if (images_staged) *images_staged = cs->stagedImages.load();
  if (bytes_staged) *bytes_staged = t4_cellstore_bytes_staged(cs->store);
  if (sec_query) *sec_query = cs->secQuery;
  if (sec_stage) *sec_stage = cs->secStage;
  return T4_OK;
}

}  // extern "C"

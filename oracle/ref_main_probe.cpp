// oracle/ref_main_probe.cpp -- TEST INFRASTRUCTURE: the reference's stage-1 translation unit (main.cpp of $(REF), unmodified, its
// `main` renamed on the command line of this file only) behind an extern "C" shim, so that functions that live in main.cpp itself
// -- ProcessRead (main.cpp:224-449), IsLowComplexity (183-205) -- can be called by the tests. Built by oracle/Makefile into
// oracle/_ref/libt4refmain.so from the sources where they lie; nothing of the reference is copied into the repository.
#define main t4_reference_stage1_main
#include "main.cpp"
#undef main

extern "C" {

// ProcessRead of one pair. r1 / q1 / r2 / q2: C strings (q1 = q2 = NULL: no qualities). outR / outQ: room for len1 + len2 + 1.
// Returns the number of records ProcessRead pushed (0-3); *flags: 1 read 1 was pushed, 2 read 2 was, 4 read 1 was pushed twice
// (weight 2; the copy's id ends in ".1"), 8 read 1 has qualities.
int refmain_process_read(const char *r1, const char *q1, const char *r2, const char *q2, char *outR, char *outQ, int *flags) {
  struct _sortRead a, b;
  a.id = strdup("p"); a.read = strdup(r1); a.qual = q1 ? strdup(q1) : NULL;
  b.id = strdup("p"); b.read = strdup(r2); b.qual = q2 ? strdup(q2) : NULL;
  static KmerCount kc(21);   // (neither is touched with countKmer == 0 beyond SeqSet::ReverseComplementInPlace, which has no state)
  static SeqSet set(9);
  std::vector<struct _sortRead> reads;
  ProcessRead(a, b, 0, kc, set, reads);
  *flags = 0;
  outR[0] = outQ[0] = 0;
  int n = (int)reads.size();
  // what was pushed: read 1 first (and its weight-2 copy), then read 2 (ProcessRead's order, main.cpp:401-448)
  bool firstIsR1 = false;
  if (n > 0) {
    // read 2 keeps the pointer b.read when it survives; anything else pushed first is read 1
    firstIsR1 = !(n == 1 && b.read != NULL && reads[0].read == b.read);
    if (firstIsR1) {
      *flags |= 1;
      size_t len = strlen(reads[0].read);
      memcpy(outR, reads[0].read, len + 1);
      if (reads[0].qual) { *flags |= 8; memcpy(outQ, reads[0].qual, len); outQ[len] = 0; }
      if (n >= 2 && reads[1].read != b.read && strlen(reads[1].id) > strlen(reads[0].id)) *flags |= 4;
    }
    if (b.read != NULL && reads[n - 1].read == b.read) *flags |= 2;
  }
  return n;
}
int refmain_is_low_complexity(const char *seq) { return IsLowComplexity(seq) ? 1 : 0; }

}

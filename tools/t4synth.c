/* tools/t4synth.c -- deterministic synthetic 150 bp paired-end read generator.
 *
 * Benchmark/test infrastructure (not part of the product path). Implements the fixed recipe of
 * SURVEY.md section 8(d) so that the GPU engine, the oracle and the compiled reference are all fed
 * the same reads:
 *   clone      = chain in {IGH,IGK,IGL,TRA,TRB} uniformly; one V, one J, one C record of that chain
 *   transcript = last <=320 bp of V (3'-chewed 0-6) + 3-24 random bases + J (5'-chewed 0-5)
 *                + first 350 bp of C
 *   abundance  = Zipf(1) over clones; fragment length U[200,420]; read_len bases from each end,
 *                mate 2 reverse-complemented; 50 % strand flip; 0.3 % substitutions; quality 'I'.
 * Cell mode: n_cells cells x 2 clones (IGH + IGK/IGL, or TRB + TRA), every pair drawn from a
 * uniformly chosen cell and one of its two clones; 16-nt barcode per cell, 10-nt random UMI per pair, written as
 * FASTA files parallel to the reads (`--cells N` -> out_prefix_bc.fa / _umi.fa).
 * Built both as a shared library (ctypes: fills fixed-stride read buffers) and as a CLI
 * (`t4synth ref.fa[.gz] n_pairs n_clones seed out_prefix` -> out_prefix_1.fq / _2.fq).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t rng_next(rng_t *r) { /* splitmix64 */
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static int rng_int(rng_t *r, int lo, int hi) { /* inclusive */
  return lo + (int)(rng_next(r) % (uint64_t)(hi - lo + 1));
}
static double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

typedef struct { char name[64]; char *seq; int len; } rec_t;
typedef struct { rec_t *r; int n, cap; } recs_t;

static int load_fasta(const char *path, recs_t *out) {
  gzFile fp = gzopen(path, "rb");
  if (!fp) return -1;
  out->n = 0; out->cap = 1024; out->r = (rec_t *)malloc(sizeof(rec_t) * out->cap);
  char *line = (char *)malloc(1 << 20);
  rec_t *cur = NULL; int cap = 0;
  while (gzgets(fp, line, 1 << 20)) {
    int l = (int)strlen(line);
    while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
    if (line[0] == '>') {
      if (out->n == out->cap) { out->cap *= 2; out->r = (rec_t *)realloc(out->r, sizeof(rec_t) * out->cap); }
      cur = &out->r[out->n++];
      int i = 0;
      while (line[1 + i] && line[1 + i] != ' ' && line[1 + i] != '\t' && i < 63) { cur->name[i] = line[1 + i]; ++i; }
      cur->name[i] = 0; cap = 1024; cur->seq = (char *)malloc(cap); cur->len = 0;
    } else if (cur) {
      for (int i = 0; i < l; ++i) {
        char c = line[i];
        if (c == '.') continue;
        if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') c = 'A';
        if (cur->len + 1 >= cap) { cap *= 2; cur->seq = (char *)realloc(cur->seq, cap); }
        cur->seq[cur->len++] = c;
      }
      cur->seq[cur->len] = 0;
    }
  }
  free(line);
  gzclose(fp);
  return out->n;
}

static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
static void revcomp(char *dst, const char *src, int n) { for (int i = 0; i < n; ++i) dst[i] = comp(src[n - 1 - i]); }

typedef struct { char *seq; int len; } clone_t;

static const char *CHAINS[5] = {"IGH", "IGK", "IGL", "TRA", "TRB"};

/* gene class of a record name: 0 V, 2 J, 3 C, -1 other (D genes etc.) */
static int gene_class(const char *nm) {
  if (nm[3] == 'V') return 0;
  if (nm[3] == 'J') return 2;
  if (nm[3] == 'D') return (nm[4] >= '0' && nm[4] <= '9') ? -1 : 3;
  return 3;
}

typedef struct {
  clone_t *clones; int nclones; double *cdf; rng_t rng; int read_len;
} synth_t;

static void *synth_open(const char *fasta, int nclones, uint64_t seed, int read_len, int cells) {
  recs_t recs;
  if (load_fasta(fasta, &recs) <= 0) return NULL;
  synth_t *s = (synth_t *)calloc(1, sizeof(synth_t));
  s->rng.s = seed * 0x2545F4914F6CDD1DULL + 12345;
  s->nclones = nclones; s->read_len = read_len;
  s->clones = (clone_t *)malloc(sizeof(clone_t) * nclones);
  /* per chain lists */
  int *lst[5][3]; int cnt[5][3];
  for (int c = 0; c < 5; ++c) for (int g = 0; g < 3; ++g) { lst[c][g] = (int *)malloc(sizeof(int) * recs.n); cnt[c][g] = 0; }
  for (int i = 0; i < recs.n; ++i) {
    int gc = gene_class(recs.r[i].name);
    if (gc < 0) continue;
    int g = gc == 0 ? 0 : gc == 2 ? 1 : 2;
    for (int c = 0; c < 5; ++c) if (!strncmp(recs.r[i].name, CHAINS[c], 3)) lst[c][g][cnt[c][g]++] = i;
  }
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  for (int k = 0; k < nclones; ++k) {
    int c = rng_int(&s->rng, 0, 4);
    if (cells) {   /* clone 2i: heavy / beta chain of cell i, clone 2i+1: its light / alpha chain */
      static int isB;
      if ((k & 1) == 0) { isB = (int)(rng_next(&s->rng) & 1); c = isB ? 0 : 4; }
      else c = isB ? rng_int(&s->rng, 1, 2) : 3;
    }
    rec_t *v = &recs.r[lst[c][0][rng_int(&s->rng, 0, cnt[c][0] - 1)]];
    rec_t *j = &recs.r[lst[c][1][rng_int(&s->rng, 0, cnt[c][1] - 1)]];
    rec_t *cg = &recs.r[lst[c][2][rng_int(&s->rng, 0, cnt[c][2] - 1)]];
    int vchew = rng_int(&s->rng, 0, 6), jchew = rng_int(&s->rng, 0, 5), nlen = rng_int(&s->rng, 3, 24);
    int vend = v->len - vchew; if (vend < 1) vend = 1;
    int vstart = vend - 320; if (vstart < 0) vstart = 0;
    int jstart = jchew < j->len ? jchew : 0;
    int clen = cg->len < 350 ? cg->len : 350;
    int tl = (vend - vstart) + nlen + (j->len - jstart) + clen;
    char *t = (char *)malloc(tl + 1); int p = 0;
    memcpy(t + p, v->seq + vstart, vend - vstart); p += vend - vstart;
    for (int i = 0; i < nlen; ++i) t[p++] = ACGT[rng_int(&s->rng, 0, 3)];
    memcpy(t + p, j->seq + jstart, j->len - jstart); p += j->len - jstart;
    memcpy(t + p, cg->seq, clen); p += clen;
    t[p] = 0;
    s->clones[k].seq = t; s->clones[k].len = p;
  }
  s->cdf = (double *)malloc(sizeof(double) * nclones);
  double tot = 0;
  for (int k = 0; k < nclones; ++k) { tot += cells ? 1.0 : 1.0 / (k + 1); s->cdf[k] = tot; }
  for (int k = 0; k < nclones; ++k) s->cdf[k] /= tot;
  for (int c = 0; c < 5; ++c) for (int g = 0; g < 3; ++g) free(lst[c][g]);
  for (int i = 0; i < recs.n; ++i) free(recs.r[i].seq);
  free(recs.r);
  return s;
}

void *t4synth_open(const char *fasta, int nclones, uint64_t seed, int read_len) { return synth_open(fasta, nclones, seed, read_len, 0); }
void *t4synth_open_cells(const char *fasta, int ncells, uint64_t seed, int read_len) { return synth_open(fasta, 2 * ncells, seed, read_len, 1); }

void t4synth_close(void *h) {
  synth_t *s = (synth_t *)h;
  for (int k = 0; k < s->nclones; ++k) free(s->clones[k].seq);
  free(s->clones); free(s->cdf); free(s);
}

/* Fill n_pairs pairs. r1/r2: n_pairs * (read_len+1) bytes, NUL-terminated fixed-stride records.
 * The generator is stateful: successive calls continue the same stream. */
void t4synth_next_ex(void *h, int64_t n_pairs, char *r1, char *r2, int *clone_of);
void t4synth_next(void *h, int64_t n_pairs, char *r1, char *r2) { t4synth_next_ex(h, n_pairs, r1, r2, NULL); }
/* clone_of (optional): index of the clone every pair was drawn from (cell mode: cell = clone / 2) */
void t4synth_next_ex(void *h, int64_t n_pairs, char *r1, char *r2, int *clone_of) {
  synth_t *s = (synth_t *)h;
  int L = s->read_len, stride = L + 1;
  char frag[512], tmp[512];
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  for (int64_t n = 0; n < n_pairs; ++n) {
    double u = rng_unit(&s->rng);
    int lo = 0, hi = s->nclones - 1;
    while (lo < hi) { int m = (lo + hi) / 2; if (s->cdf[m] < u) lo = m + 1; else hi = m; }
    clone_t *c = &s->clones[lo];
    if (clone_of) clone_of[n] = lo;
    int flen = rng_int(&s->rng, 200, 420);
    if (flen > c->len) flen = c->len;
    if (flen < L) flen = L; /* transcripts are always > 150 bp for this reference */
    int start = rng_int(&s->rng, 0, c->len - flen);
    memcpy(frag, c->seq + start, flen);
    if (rng_next(&s->rng) & 1) { revcomp(tmp, frag, flen); memcpy(frag, tmp, flen); }
    char *a = r1 + n * stride, *b = r2 + n * stride;
    memcpy(a, frag, L); a[L] = 0;
    revcomp(b, frag + flen - L, L); b[L] = 0;
    for (int i = 0; i < L; ++i) {
      if (rng_unit(&s->rng) < 0.003) { char o = a[i]; do { a[i] = ACGT[rng_int(&s->rng, 0, 3)]; } while (a[i] == o); }
      if (rng_unit(&s->rng) < 0.003) { char o = b[i]; do { b[i] = ACGT[rng_int(&s->rng, 0, 3)]; } while (b[i] == o); }
    }
  }
}

#ifdef T4SYNTH_MAIN
int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s ref.fa[.gz] n_pairs n_clones seed out_prefix [--cells N]\n", argv[0]); return 1; }
  int64_t n = atoll(argv[2]); int nclones = atoi(argv[3]); uint64_t seed = strtoull(argv[4], 0, 10);
  int ncells = (argc >= 8 && !strcmp(argv[6], "--cells")) ? atoi(argv[7]) : 0;
  void *h = ncells ? t4synth_open_cells(argv[1], ncells, seed, 150) : t4synth_open(argv[1], nclones, seed, 150);
  if (!h) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  char p1[1024], p2[1024];
  snprintf(p1, sizeof p1, "%s_1.fq", argv[5]); snprintf(p2, sizeof p2, "%s_2.fq", argv[5]);
  FILE *f1 = fopen(p1, "w"), *f2 = fopen(p2, "w"), *fb = NULL, *fu = NULL;
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  rng_t urng; urng.s = seed * 77 + 5;
  if (ncells) {
    char pb[1024], pu[1024];
    snprintf(pb, sizeof pb, "%s_bc.fa", argv[5]); snprintf(pu, sizeof pu, "%s_umi.fa", argv[5]);
    fb = fopen(pb, "w"); fu = fopen(pu, "w");
  }
  char q[151]; memset(q, 'I', 150); q[150] = 0;
  const int64_t B = 65536; char *r1 = (char *)malloc(B * 151), *r2 = (char *)malloc(B * 151);
  int *cl = (int *)malloc(sizeof(int) * B);
  for (int64_t done = 0; done < n; done += B) {
    int64_t m = n - done < B ? n - done : B;
    t4synth_next_ex(h, m, r1, r2, cl);
    for (int64_t i = 0; i < m; ++i) {
      fprintf(f1, "@r%lld\n%s\n+\n%s\n", (long long)(done + i), r1 + i * 151, q);
      fprintf(f2, "@r%lld\n%s\n+\n%s\n", (long long)(done + i), r2 + i * 151, q);
      if (ncells) {
        char bc[17], umi[11];
        rng_t b; b.s = (uint64_t)(cl[i] / 2) * 0x9E3779B97F4A7C15ULL + seed;   /* the barcode is a function of the cell */
        uint64_t x = rng_next(&b);
        for (int t = 0; t < 16; ++t) bc[t] = ACGT[(x >> (2 * t)) & 3];
        bc[16] = 0;
        for (int t = 0; t < 10; ++t) umi[t] = ACGT[rng_int(&urng, 0, 3)];
        umi[10] = 0;
        fprintf(fb, ">r%lld\n%s\n", (long long)(done + i), bc);
        fprintf(fu, ">r%lld\n%s\n", (long long)(done + i), umi);
      }
    }
  }
  fclose(f1); fclose(f2); if (fb) fclose(fb); if (fu) fclose(fu); t4synth_close(h);
  return 0;
}
#endif

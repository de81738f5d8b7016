"""ctypes bindings for the CHECKERS used by the tests and by bench.py's cpu_baseline leg:

  oracle/liboracle.so       plain-C restatement (oracle/t4_oracle.c)          -> class Oracle
  oracle/_ref/libt4ref.so   the unmodified reference behind an extern "C" shim -> class Ref
  tools/libt4synth.so       deterministic synthetic read generator            -> class Synth

Nothing in the product path (trust4_amd/) imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FA = os.path.join(ROOT, "data", "hg38_bcrtcr.fa.gz")


class Overlap(C.Structure):
    _fields_ = [("seqIdx", C.c_int), ("readStart", C.c_int), ("readEnd", C.c_int),
                ("seqStart", C.c_int), ("seqEnd", C.c_int), ("strand", C.c_int),
                ("matchCnt", C.c_int), ("indelCnt", C.c_int), ("similarity", C.c_double)]

    def tup(self):
        return (self.seqIdx, self.readStart, self.readEnd, self.seqStart, self.seqEnd, self.strand,
                self.matchCnt, self.indelCnt, self.similarity)


OV_DTYPE = np.dtype([("seqIdx", "<i4"), ("readStart", "<i4"), ("readEnd", "<i4"),
                     ("seqStart", "<i4"), ("seqEnd", "<i4"), ("strand", "<i4"),
                     ("matchCnt", "<i4"), ("indelCnt", "<i4"), ("similarity", "<f8")])


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


def build_checkers():
    import sys
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=sys.stderr)   # bench.py's stdout is ONE JSON line
    so = os.path.join(ROOT, "tools", "libt4synth.so")
    src = os.path.join(ROOT, "tools", "t4synth.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-o", so, src, "-lz"], check=True)
    cli = os.path.join(ROOT, "tools", "t4synth")
    if not os.path.exists(cli) or os.path.getmtime(cli) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-std=gnu99", "-DT4SYNTH_MAIN", "-o", cli, src, "-lz"], check=True)


class RefMain:
    """Functions of the reference's main.cpp itself (oracle/_ref/libt4refmain.so: ProcessRead, IsLowComplexity)."""
    PATH = os.path.join(ROOT, "oracle", "_ref", "libt4refmain.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)
        self.lib.refmain_process_read.restype = C.c_int
        self.lib.refmain_process_read.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        self.lib.refmain_is_low_complexity.restype = C.c_int
        self.lib.refmain_is_low_complexity.argtypes = [C.c_char_p]

    def process_read(self, r1, q1, r2, q2):
        """-> (records pushed, read 1 as pushed or "", its qualities, flags: 1 read 1 pushed, 2 read 2 pushed, 4 weight 2, 8 qualities)"""
        n = len(r1) + len(r2) + 2
        outr, outq = C.create_string_buffer(n), C.create_string_buffer(n)
        fl = C.c_int(0)
        cnt = self.lib.refmain_process_read(_b(r1), None if q1 is None else _b(q1), _b(r2), None if q2 is None else _b(q2), outr, outq, C.byref(fl))
        rd = outr.value.decode()
        return cnt, rd, outq.raw[:len(rd)], fl.value

    def is_low_complexity(self, s):
        return self.lib.refmain_is_low_complexity(_b(s))


class _SetAPI:
    """Common surface of Oracle and Ref (same call signatures)."""
    P = ""  # symbol prefix

    def _f(self, name, restype, *argtypes):
        fn = getattr(self.lib, self.P + name)
        fn.restype = restype
        fn.argtypes = list(argtypes)
        return fn

    def hits(self, read, strand=0, barcode=-1, allow_total_skip=0, sort=1, cap=1 << 17):
        buf = (C.c_int * (5 * cap))()
        n = self._hits(self.h, _b(read), strand, barcode, allow_total_skip, sort, buf, cap)
        assert n <= cap
        return np.frombuffer(buf, dtype=np.int32, count=5 * n).reshape(n, 5).copy()

    def overlaps_from_hits(self, read, strand=0, barcode=-1, allow_total_skip=0, hit_len_required=17,
                           filt=1, cap=4096, ccap=1 << 18):
        out = (Overlap * cap)()
        off = (C.c_int * (cap + 1))()
        co = (C.c_int * (2 * ccap))()
        n = self._ovh(self.h, _b(read), strand, barcode, allow_total_skip, hit_len_required, filt,
                      out, cap, off, co, ccap)
        assert n < cap
        res = []
        for i in range(n):
            chain = [(co[2 * k], co[2 * k + 1]) for k in range(off[i], off[i + 1])]
            res.append((out[i].tup(), chain))
        return res

    def overlaps_from_read(self, read, strand=0, barcode=-1, read_type=0, skip_repeats=0, cap=4096):
        out = (Overlap * cap)()
        ret = self._ovr(self.h, _b(read), strand, barcode, read_type, skip_repeats, out, cap)
        return ret, [out[i].tup() for i in range(max(ret, 0))]

    def annotate_read0(self, read):
        out = (Overlap * 4)()
        ret = self._ann(self.h, _b(read), out)
        return ret, [out[i].tup() for i in range(4)]

    def extend_overlap(self, read, mm_factor, ov):
        a = Overlap(*ov)
        b = Overlap()
        ret = self._ext(self.h, _b(read), mm_factor, C.byref(a), C.byref(b))
        return ret, b.tup()

    def assign_read(self, read, strand=0, barcode=-1):
        b = Overlap()
        ret = self._asg(self.h, _b(read), strand, barcode, C.byref(b))
        return ret, b.tup()

    def recompute_posweight(self, reads, assign):
        """SeqSet::RecomputePosWeight (SeqSet.hpp:4705-4738): reads = list of str, assign = list of overlap tuples"""
        n = len(reads)
        arr = (C.c_char_p * max(n, 1))(*[_b(r) for r in reads])
        ov = (Overlap * max(n, 1))(*[Overlap(*a) for a in assign])
        fn = self._f("recompute_posweight", None, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(Overlap))
        fn(self.h, n, arr, ov)

    def update_all_consensus_chars(self):
        """UpdateConsensus of every contig (SeqSet.hpp:4537-4588); the oracle changes the characters only, the reference also its index"""
        if self.P == "ref_":
            fn = self._f("update_all_consensus", None, C.c_void_p)
            fn(self.h)
            return None
        fn = self._f("update_all_consensus_chars", C.c_int, C.c_void_p)
        return fn(self.h)

    def posweight(self, i):
        n = self._slen(self.h, i)
        out = np.zeros((n, 4), dtype=np.int32)
        fn = self._f("seq_posweight", None, C.c_void_p, C.c_int, C.POINTER(C.c_int))
        fn(self.h, i, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def set_novel_similarity(self, v):
        fn = self._f("set_novel_seq_similarity" if self.P == "ref_" else "set_novel_similarity", None, C.c_void_p, C.c_double)
        fn(self.h, v)

    def global_alignment(self, t, p):
        t, p = _b(t), _b(p)
        al = (C.c_byte * (2 * (len(t) + len(p)) + 8))()
        sc = self._ga(t, len(t), p, len(p), al)
        out = []
        for v in al:
            if v == -1:
                break
            out.append(v)
        return sc, out

    def global_alignment_posweight(self, w, p):
        p = _b(p)
        w = np.ascontiguousarray(w, dtype=np.int32)
        lent = w.shape[0]
        al = (C.c_byte * (2 * (lent + len(p)) + 8))()
        sc = self._gapw(w.ctypes.data_as(C.POINTER(C.c_int)), lent, p, len(p), al)
        out = []
        for v in al:
            if v == -1:
                break
            out.append(v)
        return sc, out

    def has_hit_in_set(self, read, mode=0):
        return self._hashit(self.h, _b(read), mode)

    def process_read(self, r1, q1, r2, q2):
        """ProcessRead (main.cpp:224-449), oracle only -> (kind, read 1 afterwards, its qualities as bytes, flags)"""
        n = len(r1) + len(r2) + 1
        outr, outq = C.create_string_buffer(n + 1), C.create_string_buffer(n + 1)
        fl = C.c_int(0)
        fn = self._f("process_read", C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int))
        kind = fn(_b(r1), None if q1 is None else _b(q1), _b(r2), None if q2 is None else _b(q2), outr, outq, C.byref(fl))
        rd = outr.value.decode()
        return kind, rd, outq.raw[:len(rd)], fl.value

    def is_mate_overlap(self, fr, sr, min_overlap, check_tandem=1):
        fr, sr = _b(fr), _b(sr)
        off, bm = C.c_int(-1), C.c_int(-1)
        r = self._mate(fr, len(fr), sr, len(sr), min_overlap, C.byref(off), C.byref(bm), check_tandem)
        return r, off.value, bm.value

    def _bind_common(self):
        I, P, D = C.c_int, C.c_void_p, C.c_double
        OVP, IP = C.POINTER(Overlap), C.POINTER(C.c_int)
        self._hits = self._f("hits", I, P, C.c_char_p, I, I, I, I, IP, I)
        self._ovh = self._f("overlaps_from_hits", I, P, C.c_char_p, I, I, I, I, I, OVP, I, IP, IP, I)
        self._ovr = self._f("overlaps_from_read", I, P, C.c_char_p, I, I, I, I, OVP, I)
        self._ann = self._f("annotate_read0", I, P, C.c_char_p, OVP)
        self._ext = self._f("extend_overlap", I, P, C.c_char_p, D, OVP, OVP)
        self._asg = self._f("assign_read", I, P, C.c_char_p, I, I, OVP)
        self._ga = self._f("global_alignment", I, C.c_char_p, I, C.c_char_p, I, C.POINTER(C.c_byte))
        self._gapw = self._f("global_alignment_posweight", I, IP, I, C.c_char_p, I, C.POINTER(C.c_byte))
        self._mate = self._f("is_mate_overlap", I, C.c_char_p, I, C.c_char_p, I, I, IP, IP, I)
        self._hashit = self._f("has_hit_in_set", I, P, C.c_char_p, I)


class Oracle(_SetAPI):
    P = "t4o_"
    PATH = os.path.join(ROOT, "oracle", "liboracle.so")

    def __init__(self, k=9, ref_fa=None, hit_len_required=None):
        self.lib = C.CDLL(self.PATH)
        self.lib.t4o_new.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.t4o_new(k))
        self._bind_common()
        I, P = C.c_int, C.c_void_p
        self._size = self._f("size", I, P)
        self._name = self._f("seq_name", C.c_char_p, P, I)
        self._cons = self._f("seq_consensus", C.c_char_p, P, I)
        self._slen = self._f("seq_len", I, P, I)
        self._load = self._f("load_ref_fasta", I, P, C.c_char_p)
        self._novel = self._f("add_novel_seq", I, P, C.c_char_p, C.c_char_p, I, I, C.POINTER(C.c_int))
        self._lis = self._f("lis", I, C.POINTER(C.c_int), I, C.POINTER(C.c_int))
        self._batch = self._f("annotate_batch", C.c_int64, P, C.c_char_p, I, C.c_int64,
                              C.POINTER(Overlap), C.POINTER(C.c_int64))
        self._f("set_hit_len_required", None, P, I)
        self._f("set_radius", None, P, I)
        self._f("set_consider_barcode", None, P, I)
        if ref_fa:
            assert self._load(self.h, _b(ref_fa)) > 0
        if hit_len_required is not None:
            self.set_hit_len_required(hit_len_required)

    def set_hit_len_required(self, l):
        self.lib.t4o_set_hit_len_required(self.h, l)

    def set_radius(self, r):
        self.lib.t4o_set_radius(self.h, r)

    def size(self):
        return self._size(self.h)

    def name(self, i):
        return self._name(self.h, i).decode()

    def consensus(self, i):
        return self._cons(self.h, i).decode()

    def add_novel(self, name, seq, strand=1, barcode=-1, posweight=None):
        pw = None
        if posweight is not None:
            posweight = np.ascontiguousarray(posweight, dtype=np.int32)
            pw = posweight.ctypes.data_as(C.POINTER(C.c_int))
        return self._novel(self.h, _b(name), _b(seq), strand, barcode, pw)

    def lis(self, pairs):
        arr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.zeros_like(arr)
        n = self._lis(arr.ctypes.data_as(C.POINTER(C.c_int)), len(arr), out.ctypes.data_as(C.POINTER(C.c_int)))
        return [tuple(x) for x in out[:n].tolist()]

    def annotate_batch(self, reads_buf, stride, n):
        """reads_buf: bytes/np.uint8 of n fixed-stride NUL-terminated reads. -> (ov[n,4], hits[n], total)."""
        out = np.zeros((n, 4), dtype=OV_DTYPE)
        hp = np.zeros(n, dtype=np.int64)
        buf = np.ascontiguousarray(np.frombuffer(reads_buf, dtype=np.uint8))
        tot = self._batch(self.h, buf.ctypes.data_as(C.c_char_p), stride, n,
                          out.ctypes.data_as(C.POINTER(Overlap)), hp.ctypes.data_as(C.POINTER(C.c_int64)))
        return out, hp, tot

    def __del__(self):
        try:
            self.lib.t4o_free(self.h)
        except Exception:
            pass


class Ref(_SetAPI):
    """The real reference (oracle/_ref/libt4ref.so). available() is False when it was not built."""
    P = "ref_"
    PATH = os.path.join(ROOT, "oracle", "_ref", "libt4ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self, k=9, ref_fa=None, hit_len_required=None):
        self.lib = C.CDLL(self.PATH)
        self.lib.ref_seqset_new.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.ref_seqset_new(k))
        self._bind_common()
        I, P = C.c_int, C.c_void_p
        self._size = self._f("size", I, P)
        self._name = self._f("seq_name", C.c_char_p, P, I)
        self._cons = self._f("seq_consensus", C.c_char_p, P, I)
        self._slen = self._f("seq_len", I, P, I)
        self._f("input_ref_fa", None, P, C.c_char_p)
        self._f("set_hit_len_required", None, P, I)
        self._f("set_radius", None, P, I)
        self._novel = self._f("input_novel_read", I, P, C.c_char_p, C.c_char_p, I, I)
        self._setpw = self._f("seq_set_posweight", None, P, I, C.POINTER(C.c_int))
        self._rlis = self._f("lis", I, P, C.POINTER(C.c_int), I, C.POINTER(C.c_int))
        if ref_fa:
            self.lib.ref_input_ref_fa(self.h, _b(ref_fa))
        if hit_len_required is not None:
            self.set_hit_len_required(hit_len_required)

    def set_hit_len_required(self, l):
        self.lib.ref_set_hit_len_required(self.h, l)

    def set_radius(self, r):
        self.lib.ref_set_radius(self.h, r)

    def size(self):
        return self._size(self.h)

    def name(self, i):
        return self._name(self.h, i).decode()

    def consensus(self, i):
        return self._cons(self.h, i).decode()

    def add_novel(self, name, seq, strand=1, barcode=-1, posweight=None):
        i = self._novel(self.h, _b(name), _b(seq), strand, barcode)
        if posweight is not None:
            posweight = np.ascontiguousarray(posweight, dtype=np.int32)
            self._setpw(self.h, i, posweight.ctypes.data_as(C.POINTER(C.c_int)))
        return i

    def lis(self, pairs):
        arr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.zeros_like(arr)
        n = self._rlis(self.h, arr.ctypes.data_as(C.POINTER(C.c_int)), len(arr), out.ctypes.data_as(C.POINTER(C.c_int)))
        return [tuple(x) for x in out[:n].tolist()]


class Synth:
    PATH = os.path.join(ROOT, "tools", "libt4synth.so")

    def __init__(self, n_clones, seed, read_len=150, fasta=REF_FA):
        self.lib = C.CDLL(self.PATH)
        self.lib.t4synth_open.restype = C.c_void_p
        self.lib.t4synth_open.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_int]
        self.lib.t4synth_next.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p]
        self.lib.t4synth_close.argtypes = [C.c_void_p]
        self.read_len = read_len
        self.h = C.c_void_p(self.lib.t4synth_open(_b(fasta), n_clones, seed, read_len))
        assert self.h

    def next_pairs(self, n):
        """-> (r1, r2) uint8 arrays of shape [n, read_len+1] (NUL-terminated rows)."""
        st = self.read_len + 1
        r1 = np.zeros((n, st), dtype=np.uint8)
        r2 = np.zeros((n, st), dtype=np.uint8)
        self.lib.t4synth_next(self.h, n, r1.ctypes.data_as(C.c_char_p), r2.ctypes.data_as(C.c_char_p))
        return r1, r2

    def next_reads(self, n_pairs):
        """Interleaved mate1, mate2, ... as one [2n, read_len+1] array."""
        r1, r2 = self.next_pairs(n_pairs)
        out = np.empty((2 * n_pairs, self.read_len + 1), dtype=np.uint8)
        out[0::2] = r1
        out[1::2] = r2
        return out

    def __del__(self):
        try:
            self.lib.t4synth_close(self.h)
        except Exception:
            pass


def rows_to_strs(arr):
    return [bytes(r[:-1]).split(b"\0")[0].decode() for r in arr]


class RefSeqSet:
    """A mutable reference SeqSet (novel contigs) driven through oracle/_ref: the checker of the Add path."""

    def __init__(self, k, hit_len_required=31, consider_barcode=False):
        self.lib = C.CDLL(Ref.PATH)
        self.lib.ref_seqset_new.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.ref_seqset_new(k))
        I, P = C.c_int, C.c_void_p
        self.lib.ref_set_hit_len_required.argtypes = [P, I]
        self.lib.ref_input_novel_read.argtypes = [P, C.c_char_p, C.c_char_p, I, I]
        self.lib.ref_add_read.argtypes = [P, C.c_char_p, C.c_char_p, C.POINTER(I), I, I, I, C.c_double]
        self.lib.ref_repeat_add_read.argtypes = [P, C.c_char_p]
        self.lib.ref_update_all_consensus.argtypes = [P]
        self.lib.ref_change_kmer_length.argtypes = [P, I]
        self.lib.ref_output.argtypes = [P, C.c_char_p]
        self.lib.ref_size.argtypes = [P]
        self.lib.ref_set_consider_barcode.argtypes = [P, I]
        self.lib.ref_release_finished_barcode.argtypes = [P, I, I]
        self.lib.ref_output_barcodes.argtypes = [P, C.c_char_p, P, I]
        self.lib.ref_set_hit_len_required(self.h, hit_len_required)
        if consider_barcode:
            self.lib.ref_set_consider_barcode(self.h, 1)

    def release_finished_barcode(self, barcode, total=1):
        self.lib.ref_release_finished_barcode(self.h, barcode, total)

    def output_barcodes(self, path, names):
        arr = (C.c_char_p * len(names))(*[_b(x) for x in names])
        self.lib.ref_output_barcodes(self.h, _b(path), C.cast(arr, C.c_void_p), len(names))

    def input_novel_read(self, name, read, strand, barcode=-1):
        return self.lib.ref_input_novel_read(self.h, _b(name), _b(read), strand, barcode)

    def add_read(self, read, gene_name, strand, barcode=-1, min_kmer_count=1, repetitive_data=0, similarity_threshold=0.9):
        st = C.c_int(strand)
        r = self.lib.ref_add_read(self.h, _b(read), _b(gene_name), C.byref(st), barcode, min_kmer_count, repetitive_data, similarity_threshold)
        return r, st.value

    def repeat_add_read(self, read):
        return self.lib.ref_repeat_add_read(self.h, _b(read))

    def update_all_consensus(self):
        self.lib.ref_update_all_consensus(self.h)

    def change_kmer_length(self, k):
        self.lib.ref_change_kmer_length(self.h, k)

    def output(self, path):
        self.lib.ref_output(self.h, _b(path))

    def size(self):
        return self.lib.ref_size(self.h)


class KmerCountChecker:
    """KmerCount of the oracle ("t4o_kc_") or of the compiled reference ("ref_kc_"): add reads, then per-read count statistics
    with the quality trimming (GetCountStatsAndTrim). stats() returns (ret, min, median, avg, read_after, qual_after)."""

    def __init__(self, k=21, reference=False):
        self.lib = C.CDLL(Ref.PATH if reference else Oracle.PATH)
        p = "ref_kc_" if reference else "t4o_kc_"
        self._new, self._free, self._add, self._stats = (getattr(self.lib, p + n) for n in ("new", "free", "add", "stats"))
        self._new.restype = C.c_void_p
        self._new.argtypes = [C.c_int]
        self._free.argtypes = [C.c_void_p]
        self._add.argtypes = [C.c_void_p, C.c_char_p]
        self._stats.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        self.h = C.c_void_p(self._new(k))

    def add(self, read):
        return self._add(self.h, _b(read))

    def stats(self, read, qual=None):
        rb = C.create_string_buffer(_b(read), len(read) + 1)
        qb = C.create_string_buffer(_b(qual), len(qual) + 1) if qual is not None else None
        mn, md, av = C.c_int(0), C.c_int(0), C.c_float(0)
        ret = self._stats(self.h, rb, qb, C.byref(mn), C.byref(md), C.byref(av))
        return ret, mn.value, md.value, av.value, rb.value.decode(), (qb.value.decode() if qb is not None else None)

    def __del__(self):
        try:
            self._free(self.h)
        except Exception:
            pass

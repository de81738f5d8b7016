"""ctypes mirror of include/trust4_hip.h (same names, argument meaning and error behaviour)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

OV_DTYPE = np.dtype([("seqIdx", "<i4"), ("readStart", "<i4"), ("readEnd", "<i4"), ("seqStart", "<i4"),
                     ("seqEnd", "<i4"), ("strand", "<i4"), ("matchCnt", "<i4"), ("indelCnt", "<i4"),
                     ("similarity", "<f8")])
HIT_DTYPE = np.dtype([("idx", "<i4"), ("offset", "<i4"), ("readOffset", "<i4"), ("strand", "<i4"),
                      ("repeats", "<i4")])


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("chain_kernel_ms", C.c_double), ("total_hits", C.c_int64),
                ("reads", C.c_int64), ("tier_reads", C.c_int64 * 6), ("launches", C.c_int64)]


class T4Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("t4 error %d: %s" % (code, msg))
        self.code = code


class SeqRecord(C.Structure):   # t4_seq_record
    _fields_ = [("base_off", C.c_int64), ("len", C.c_int32), ("barcode", C.c_int32), ("name", C.c_char * 8)]


class IndexDelta(C.Structure):   # t4_index_delta
    _fields_ = [("table_slots", C.c_int64), ("table_rebuilt", C.c_int32), ("seq_cap", C.c_int32), ("post_cap", C.c_int64),
                ("base_cap", C.c_int64), ("nseq", C.c_int32), ("max_seq_len", C.c_int32),
                ("n_slots", C.c_int64), ("slot", C.c_void_p), ("slot_code", C.c_void_p), ("slot_start", C.c_void_p), ("slot_cnt", C.c_void_p),
                ("n_post_runs", C.c_int64), ("post_at", C.c_void_p), ("post_len", C.c_void_p), ("post_data", C.c_void_p),
                ("n_seqs", C.c_int32), ("seq_id", C.c_void_p), ("seq", C.c_void_p),
                ("n_base_runs", C.c_int64), ("base_at", C.c_void_p), ("base_len", C.c_void_p), ("base_cons", C.c_void_p), ("base_pw", C.c_void_p)]


def mix64(z):
    """slot hash of the table of a live set (== t4k::mix64): first free slot of mix64(code) & (slots - 1), +1, ..."""
    m = (1 << 64) - 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def lib_path():
    """The product library. T4_LIB overrides it (the CPU test-suite points it at the emulator build)."""
    return os.environ.get("T4_LIB", os.path.join(HERE, "libt4hip.so"))


def _load():
    path = lib_path()
    if not os.path.exists(path):
        raise T4Error(-2, "%s not found: build it with `python -m trust4_amd.build` (there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    P, I, L = C.c_void_p, C.c_int, C.c_int64
    sig = {
        "t4_init": (I, [I, C.POINTER(P)]), "t4_destroy": (None, [P]), "t4_sync": (I, [P]),
        "t4_last_error": (C.c_char_p, [P]), "t4_device_cus": (I, [P]), "t4_last_stats": (I, [P, C.POINTER(Stats)]),
        "t4_index_create": (I, [P, I, I, C.POINTER(P)]), "t4_index_destroy": (None, [P]),
        "t4_index_set_params": (I, [P, I, I, C.c_double]), "t4_index_load_ref_fasta": (I, [P, C.c_char_p]),
        "t4_index_add_ref_record": (I, [P, C.c_char_p, C.c_char_p, C.POINTER(I)]),
        "t4_index_add_contig": (I, [P, C.c_char_p, C.c_char_p, I, P, C.POINTER(I)]),
        "t4_index_apply_delta": (I, [P, C.POINTER(IndexDelta)]), "t4_reads_upload_flags": (I, [P, P, P, P, L, I, C.POINTER(P)]),
        "t4_index_commit": (I, [P]), "t4_index_size": (I, [P]), "t4_index_seq_len": (I, [P, I]),
        "t4_index_seq_name": (C.c_char_p, [P, I]), "t4_index_seq_consensus": (C.c_char_p, [P, I]),
        "t4_reads_upload": (I, [P, P, P, P, L, C.POINTER(P)]), "t4_batch_destroy": (None, [P]),
        "t4_batch_size": (L, [P]),
        "t4_hits": (I, [P, P, I, I, P, P, L]), "t4_overlaps": (I, [P, P, I, I, I, P, P]),
        "t4_annotate_rough": (I, [P, P, P]),
        "t4_gap_dp": (I, [P, I, I, I, P, P, P, P, P]),
        "t4_gap_dp_align": (I, [P, I, I, I, P, P, P, P, P, P, I]),
        "t4_mate_overlap": (I, [P, I, P, P, P, P, P, I, P]), "t4_has_hit": (I, [P, P, I, P]),
        "t4_process_pairs": (I, [P, I, P, P, P, P, P, P, P, P, P, P, P]),
        "t4_extend": (I, [P, P, I, P, P, C.c_double, P, P]), "t4_assign": (I, [P, P, I, P, P]), "t4_assign_strands": (I, [P, P, P, P, P]),
        "t4_posweight_recompute": (I, [P, P, P, P, P, C.c_int64]),
        "t4_consensus_recompute": (I, [P, P, P, P, P, C.c_int64, P, C.c_int64, P]),
        "t4_assembler_create": (I, [P, I, I, C.POINTER(P)]), "t4_assembler_destroy": (None, [P]),
        "t4_assembler_set_params": (I, [P, I, I, C.c_double]),
        "t4_assembler_input_novel_read": (I, [P, C.c_char_p, C.c_char_p, I, I]),
        "t4_assembler_add_read": (I, [P, C.c_char_p, C.c_char_p, C.POINTER(I), I, I, I, C.c_double]),
        "t4_assembler_prefetch": (I, [P, I, P, P, P, I]), "t4_assembler_window_valid": (I, [P]),
        "t4_assembler_counters": (I, [P, P, P, P]),
        "t4_assembler_repeat_add_read": (I, [P, C.c_char_p]), "t4_assembler_update_all_consensus": (I, [P]),
        "t4_assembler_change_kmer_length": (I, [P, I]),
        "t4_assembler_output": (I, [P, C.c_char_p]), "t4_assembler_size": (I, [P]), "t4_assembler_index_postings": (L, [P]),
        "t4_assembler_release_finished_barcode": (I, [P, I, I]), "t4_assembler_release_shallow_contigs": (I, [P, I]),
        "t4_assembler_output_barcodes": (I, [P, C.c_char_p, P, I]), "t4_cellset_release_shallow_contigs": (I, [P, I]),
        "t4_cellset_create": (I, [P, I, C.POINTER(P)]), "t4_cellset_destroy": (None, [P]),
        "t4_cellset_set_params": (I, [P, I, I, C.c_double]), "t4_cellset_cell": (I, [P, I, C.POINTER(P)]),
        "t4_cellset_close_cell": (I, [P, P]), "t4_cellset_prefetch": (I, [P, I, P, P, P, I]),
        "t4_cellset_update_all_consensus": (I, [P]), "t4_cellset_set_threads": (I, [P, I]), "t4_cellset_size": (I, [P]),
        "t4_cellset_output": (I, [P, C.c_char_p, P, I]), "t4_cellset_counters": (I, [P, P, P, P, P, P, P]),
    }
    for name, (res, args) in sig.items():
        if not hasattr(lib, name) and os.environ.get("T4_LIB"):   # an older build of the library under T4_LIB (A/B timing of a kernel): the calls it lacks fail when made
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class Engine:
    """t4_ctx: one GPU, one stream."""

    def __init__(self, device=0):
        self.lib = _load()
        h = C.c_void_p()
        rc = self.lib.t4_init(device, C.byref(h))
        if rc != 0:
            raise T4Error(rc, "t4_init failed (no MI355X visible?)")
        self.h = h

    def check(self, rc):
        if rc != 0:
            raise T4Error(rc, self.lib.t4_last_error(self.h).decode())

    def cus(self):
        return self.lib.t4_device_cus(self.h)

    def stats(self):
        s = Stats()
        self.check(self.lib.t4_last_stats(self.h, C.byref(s)))
        return {"kernel_ms": s.kernel_ms, "chain_kernel_ms": s.chain_kernel_ms, "total_hits": s.total_hits,
                "reads": s.reads, "tier_reads": list(s.tier_reads), "launches": s.launches}

    def gap_dp(self, kind, targets, patterns, impl=0):
        """targets: list of str (kind 0) or list of [len,4] int32 arrays (kind 1); patterns: list of str.
        -> int32 [n, 4] (matches, mismatches, indels, status)."""
        n = len(patterns)
        toff = np.zeros(n + 1, dtype=np.int64)
        poff = np.zeros(n + 1, dtype=np.int64)
        toff[1:] = np.cumsum([len(t) for t in targets])
        poff[1:] = np.cumsum([len(p) for p in patterns])
        pbuf = np.frombuffer(("".join(patterns) + "\0").encode(), dtype=np.uint8)
        if kind == 0:
            tbuf = np.frombuffer(("".join(targets) + "\0").encode(), dtype=np.uint8)
        else:
            tbuf = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1, 4) for t in targets] + [np.zeros((1, 4), np.int32)]))
        out = np.zeros((n, 4), dtype=np.int32)
        if impl == 4:   # posWeight aligner with its traceback: the edit strings come back too (lists of 0 match / 1 mismatch / 2 insert / 3 delete)
            stride = int(max(len(t) + len(p) for t, p in zip(targets, patterns))) + 2 if n else 2
            al = np.zeros((n, stride), dtype=np.int8)
            self.check(self.lib.t4_gap_dp_align(self.h, kind, impl, n, toff.ctypes.data_as(C.c_void_p), poff.ctypes.data_as(C.c_void_p),
                                                tbuf.ctypes.data_as(C.c_void_p), pbuf.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                al.ctypes.data_as(C.c_void_p), stride))
            strings = []
            for row in al:
                end = np.nonzero(row == -1)[0]
                strings.append(row[: int(end[0])].tolist() if len(end) else None)
            return out, strings
        self.check(self.lib.t4_gap_dp(self.h, kind, impl, n, toff.ctypes.data_as(C.c_void_p), poff.ctypes.data_as(C.c_void_p),
                                      tbuf.ctypes.data_as(C.c_void_p), pbuf.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def mate_overlap(self, firsts, seconds, min_overlaps, check_tandem=False):
        """-> int32 [n, 3]: IsMateOverlap return value, offset, bestMatchCnt"""
        n = len(firsts)
        foff = np.zeros(n + 1, dtype=np.int64); soff = np.zeros(n + 1, dtype=np.int64)
        foff[1:] = np.cumsum([len(x) for x in firsts]); soff[1:] = np.cumsum([len(x) for x in seconds])
        fb = np.frombuffer(("".join(firsts) + "\0").encode(), dtype=np.uint8)
        sb = np.frombuffer(("".join(seconds) + "\0").encode(), dtype=np.uint8)
        mo = np.ascontiguousarray(min_overlaps, dtype=np.int32)
        out = np.zeros((n, 3), dtype=np.int32)
        V = C.c_void_p
        self.check(self.lib.t4_mate_overlap(self.h, n, foff.ctypes.data_as(V), fb.ctypes.data_as(V), soff.ctypes.data_as(V), sb.ctypes.data_as(V),
                                            mo.ctypes.data_as(V), 1 if check_tandem else 0, out.ctypes.data_as(V)))
        return out

    def process_pairs(self, r1, q1, r2, q2):
        """ProcessRead (main.cpp:224-449) of n mate pairs: r1 / r2 lists of str, q1 / q2 lists of str or None (no qualities).
        -> list of (kind, read 1 afterwards, its qualities as bytes, flags) -- see t4_process_pairs"""
        n = len(r1)
        o1 = np.zeros(n + 1, dtype=np.int64); o2 = np.zeros(n + 1, dtype=np.int64); oo = np.zeros(n + 1, dtype=np.int64)
        o1[1:] = np.cumsum([len(x) for x in r1]); o2[1:] = np.cumsum([len(x) for x in r2])
        oo[1:] = np.cumsum([len(a) + len(b) + 1 for a, b in zip(r1, r2)])
        V = C.c_void_p
        buf = lambda strs: np.frombuffer(("".join(strs) + "\0").encode("latin-1"), dtype=np.uint8)
        b1, b2 = buf(r1), buf(r2)
        bq1 = buf(q1) if q1 is not None else None
        bq2 = buf(q2) if q2 is not None else None
        hq = np.full(n, (1 if q1 is not None else 0) | (2 if q2 is not None else 0), dtype=np.uint8)
        outr = np.zeros(int(oo[n]) + 1, dtype=np.uint8); outq = np.zeros(int(oo[n]) + 1, dtype=np.uint8)
        meta = np.zeros((n, 4), dtype=np.int32)
        self.check(self.lib.t4_process_pairs(self.h, n, o1.ctypes.data_as(V), b1.ctypes.data_as(V), None if bq1 is None else bq1.ctypes.data_as(V),
                                             o2.ctypes.data_as(V), b2.ctypes.data_as(V), None if bq2 is None else bq2.ctypes.data_as(V),
                                             hq.ctypes.data_as(V), oo.ctypes.data_as(V), outr.ctypes.data_as(V), outq.ctypes.data_as(V), meta.ctypes.data_as(V)))
        res = []
        for i in range(n):
            kind, ln, fl = int(meta[i, 0]), int(meta[i, 1]), int(meta[i, 2])
            if fl & 16:
                rd = outr[oo[i]:oo[i] + ln].tobytes().decode("latin-1"); ql = outq[oo[i]:oo[i] + ln].tobytes()
            else:
                rd = r1[i]; ql = q1[i].encode("latin-1") if q1 is not None else bytes(ln)
            res.append((kind, rd, ql, fl & 15))
        return res

    def index(self, k, consider_barcode=False):
        return Index(self, k, consider_barcode)

    def kmer_counter(self, k=21, max_kmers=1 << 20, per_barcode=False):
        return KmerCounter(self, k, max_kmers, per_barcode)

    def upload(self, reads, barcodes=None):
        return Batch(self, reads, barcodes)

    def close(self):
        if getattr(self, "h", None):
            self.lib.t4_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KmerCounter:
    """t4_kmer_counter: the reference's `KmerCount` (canonical k-mer counts of the read set, GetCountStatsAndTrim per read)."""

    def __init__(self, eng, k, max_kmers, per_barcode=False):
        self.eng = eng
        self.h = C.c_void_p()
        eng.check(eng.lib.t4_kmer_count_create(eng.h, k, C.c_int64(max_kmers), 1 if per_barcode else 0, C.byref(self.h)))

    def add(self, batch):
        self.eng.check(self.eng.lib.t4_kmer_count_add(self.h, batch.h))
        return self

    def stats(self, batch, quals=None):
        """-> (min int32[n], median int32[n], avg float32[n], new_len int32[n]); quals: list of str, one per read, or None"""
        n = batch.n
        mn, md, ln = (np.zeros(n, dtype=np.int32) for _ in range(3))
        av = np.zeros(n, dtype=np.float32)
        V = C.c_void_p
        qb, qo = None, None
        if quals is not None:
            qoff = np.zeros(n + 1, dtype=np.int64)
            qoff[1:] = np.cumsum([len(x) for x in quals])
            qbuf = np.frombuffer(("".join(quals) + "\0").encode(), dtype=np.uint8)
            qb, qo = qbuf.ctypes.data_as(V), qoff.ctypes.data_as(V)
        self.eng.check(self.eng.lib.t4_kmer_count_stats(self.h, batch.h, qb, qo, mn.ctypes.data_as(V), md.ctypes.data_as(V), av.ctypes.data_as(V), ln.ctypes.data_as(V)))
        return mn, md, av, ln

    def export(self):
        """-> (codes uint64[n], counts int32[n]): the table's pairs, in any order (t4_kmer_count_export)"""
        n = C.c_int64(0)
        self.eng.check(self.eng.lib.t4_kmer_count_export(self.h, None, None, C.c_int64(0), C.byref(n)))
        codes, vals = np.zeros(max(1, n.value), dtype=np.uint64), np.zeros(max(1, n.value), dtype=np.int32)
        if n.value:
            self.eng.check(self.eng.lib.t4_kmer_count_export(self.h, codes.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), C.c_int64(n.value), C.byref(n)))
        return codes[:n.value], vals[:n.value]

    def merge(self, codes, counts, only_present=False):
        """count[codes[i]] += counts[i] (t4_kmer_count_merge); only_present: k-mers the table does not hold are passed over"""
        codes, counts = np.ascontiguousarray(codes, dtype=np.uint64), np.ascontiguousarray(counts, dtype=np.int32)
        self.eng.check(self.eng.lib.t4_kmer_count_merge(self.h, codes.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), C.c_int64(len(codes)), 1 if only_present else 0))
        return self

    def distinct(self):
        self.eng.lib.t4_kmer_count_distinct.restype = C.c_int64
        return int(self.eng.lib.t4_kmer_count_distinct(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.eng.lib.t4_kmer_count_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Assembler:
    """t4_assembler: the reference's `SeqSet seqSet` of novel contigs with its Add path (same call signatures)."""

    def __init__(self, eng, k, consider_barcode=False, hit_len_required=31, radius=10, novel_seq_similarity=0.9, _cell=None):
        self.eng = eng
        self.owned = _cell is None
        if _cell is not None:      # a cell of a CellSet: the handle belongs to the set
            self.h = _cell
            return
        h = C.c_void_p()
        eng.check(eng.lib.t4_assembler_create(eng.h, k, 1 if consider_barcode else 0, C.byref(h)))
        self.h = h
        eng.check(eng.lib.t4_assembler_set_params(h, hit_len_required, radius, novel_seq_similarity))

    def release_finished_barcode(self, barcode, contig_min_cov=0):
        self.eng.check(self.eng.lib.t4_assembler_release_finished_barcode(self.h, barcode, contig_min_cov))

    def _ret(self, r):
        if r < -50:
            raise T4Error(r + 100, self.eng.lib.t4_last_error(self.eng.h).decode())
        return r

    def input_novel_read(self, name, read, strand, barcode=-1):
        return self._ret(self.eng.lib.t4_assembler_input_novel_read(self.h, name.encode(), read.encode(), strand, barcode))

    def add_read(self, read, gene_name, strand, barcode=-1, min_kmer_count=1, repetitive_data=0, similarity_threshold=0.9):
        st = C.c_int(strand)
        r = self._ret(self.eng.lib.t4_assembler_add_read(self.h, read.encode(), gene_name.encode(), C.byref(st), barcode, min_kmer_count,
                                                        repetitive_data, similarity_threshold))
        return r, st.value

    def repeat_add_read(self, read):
        return self._ret(self.eng.lib.t4_assembler_repeat_add_read(self.h, read.encode()))

    def prefetch(self, reads, strands, barcodes=None, repetitive_data=0):
        n = len(reads)
        arr = (C.c_char_p * n)(*[r.encode() for r in reads])
        st = np.ascontiguousarray(strands, dtype=np.int32)
        bc = None if barcodes is None else np.ascontiguousarray(barcodes, dtype=np.int32)
        self.eng.check(self.eng.lib.t4_assembler_prefetch(self.h, n, C.cast(arr, C.c_void_p), st.ctypes.data_as(C.c_void_p),
                                                          None if bc is None else bc.ctypes.data_as(C.c_void_p), repetitive_data))

    def window_valid(self):
        return bool(self.eng.lib.t4_assembler_window_valid(self.h))

    def counters(self):
        q, r, h = C.c_int64(), C.c_int64(), C.c_int64()
        self.eng.check(self.eng.lib.t4_assembler_counters(self.h, C.byref(q), C.byref(r), C.byref(h)))
        return {"queries": q.value, "refreshes": r.value, "window_hits": h.value}

    def change_kmer_length(self, k):
        self.eng.check(self.eng.lib.t4_assembler_change_kmer_length(self.h, k))

    def update_all_consensus(self):
        self.eng.check(self.eng.lib.t4_assembler_update_all_consensus(self.h))

    def output(self, path):
        self.eng.check(self.eng.lib.t4_assembler_output(self.h, path.encode()))

    def size(self):
        return self.eng.lib.t4_assembler_size(self.h)

    def close(self):
        if getattr(self, "h", None):
            if self.owned:
                self.eng.lib.t4_assembler_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CellSet:
    """t4_cellset: the reference's barcode-mode `SeqSet seqSet` as a disjoint union of per-barcode sets."""

    def __init__(self, eng, k, hit_len_required=13, radius=10, novel_seq_similarity=0.9):
        self.eng = eng
        h = C.c_void_p()
        eng.check(eng.lib.t4_cellset_create(eng.h, k, C.byref(h)))
        self.h = h
        eng.check(eng.lib.t4_cellset_set_params(h, hit_len_required, radius, novel_seq_similarity))
        self.cells = {}

    def cell(self, barcode):
        if barcode not in self.cells:
            c = C.c_void_p()
            self.eng.check(self.eng.lib.t4_cellset_cell(self.h, barcode, C.byref(c)))
            self.cells[barcode] = Assembler(self.eng, 0, _cell=c)
        return self.cells[barcode]

    def close_cell(self, barcode):
        self.eng.check(self.eng.lib.t4_cellset_close_cell(self.h, self.cell(barcode).h))

    def prefetch(self, barcodes, reads, strands, repetitive_data=0):
        n = len(reads)
        cells = (C.c_void_p * n)(*[self.cell(b).h.value for b in barcodes])
        arr = (C.c_char_p * n)(*[r.encode() for r in reads])
        st = np.ascontiguousarray(strands, dtype=np.int32)
        self.eng.check(self.eng.lib.t4_cellset_prefetch(self.h, n, C.cast(cells, C.c_void_p), C.cast(arr, C.c_void_p),
                                                        st.ctypes.data_as(C.c_void_p), repetitive_data))

    def update_all_consensus(self):
        self.eng.check(self.eng.lib.t4_cellset_update_all_consensus(self.h))

    def size(self):
        return self.eng.lib.t4_cellset_size(self.h)

    def output(self, path, names):
        arr = (C.c_char_p * len(names))(*[x.encode() for x in names])
        self.eng.check(self.eng.lib.t4_cellset_output(self.h, path.encode(), C.cast(arr, C.c_void_p), len(names)))

    def counters(self):
        q, r, i, b = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        sq, ss = C.c_double(), C.c_double()
        self.eng.check(self.eng.lib.t4_cellset_counters(self.h, C.byref(q), C.byref(r), C.byref(i), C.byref(b), C.byref(sq), C.byref(ss)))
        return {"query_batches": q.value, "reads_queried": r.value, "images_staged": i.value, "bytes_staged": b.value,
                "sec_query": sq.value, "sec_stage": ss.value}

    def close(self):
        if getattr(self, "h", None):
            for c in self.cells.values():
                c.h = None
            self.eng.lib.t4_cellset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """t4_index: the device image of one SeqSet."""

    def __init__(self, eng, k, consider_barcode=False):
        self.eng = eng
        h = C.c_void_p()
        eng.check(eng.lib.t4_index_create(eng.h, k, 1 if consider_barcode else 0, C.byref(h)))
        self.h = h

    def set_params(self, hit_len_required=31, radius=10, novel_seq_similarity=0.9):
        self.eng.check(self.eng.lib.t4_index_set_params(self.h, hit_len_required, radius, novel_seq_similarity))
        return self

    def load_ref_fasta(self, path):
        self.eng.check(self.eng.lib.t4_index_load_ref_fasta(self.h, path.encode()))
        return self

    def add_ref_record(self, name, seq):
        sid = C.c_int(-1)
        self.eng.check(self.eng.lib.t4_index_add_ref_record(self.h, name.encode(), seq.encode(), C.byref(sid)))
        return sid.value

    def add_contig(self, name, consensus, barcode=-1, posweight=None):
        sid = C.c_int(-1)
        pw = None
        if posweight is not None:
            posweight = np.ascontiguousarray(posweight, dtype=np.int32)
            pw = posweight.ctypes.data_as(C.c_void_p)
        self.eng.check(self.eng.lib.t4_index_add_contig(self.h, name.encode(), consensus.encode(), barcode, pw, C.byref(sid)))
        return sid.value

    def commit(self):
        self.eng.check(self.eng.lib.t4_index_commit(self.h))
        return self

    def apply_delta(self, table_slots, table_rebuilt, post_cap, base_cap, seq_cap, nseq, max_seq_len, slots=(), post_runs=(), seqs=(), base_runs=()):
        """t4_index_apply_delta. slots: (slot, code, start, cnt); post_runs: (at, [(idx, offset), ...]); seqs: (id, base_off, len,
        barcode, name); base_runs: (at, consensus bytes, predicate bytes)."""
        d = IndexDelta()
        d.table_slots, d.table_rebuilt, d.post_cap, d.base_cap, d.seq_cap, d.nseq, d.max_seq_len = table_slots, table_rebuilt, post_cap, base_cap, seq_cap, nseq, max_seq_len
        keep = []

        def arr(values, dtype):
            a = np.ascontiguousarray(np.array(list(values), dtype=dtype))
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        d.n_slots = len(slots)
        d.slot, d.slot_code = arr((x[0] for x in slots), np.int64), arr((x[1] for x in slots), np.uint64)
        d.slot_start, d.slot_cnt = arr((x[2] for x in slots), np.uint32), arr((x[3] for x in slots), np.uint32)
        d.n_post_runs = len(post_runs)
        d.post_at, d.post_len = arr((r[0] for r in post_runs), np.int64), arr((len(r[1]) for r in post_runs), np.int32)
        d.post_data = arr((v for r in post_runs for p in r[1] for v in p), np.int32)
        recs = (SeqRecord * max(1, len(seqs)))()
        for i, (sid, off, ln, bc, nm) in enumerate(seqs):
            recs[i].base_off, recs[i].len, recs[i].barcode, recs[i].name = off, ln, bc, nm.encode()[:8]
        keep.append(recs)
        d.n_seqs, d.seq_id, d.seq = len(seqs), arr((x[0] for x in seqs), np.int32), C.cast(recs, C.c_void_p)
        d.n_base_runs = len(base_runs)
        d.base_at, d.base_len = arr((r[0] for r in base_runs), np.int64), arr((len(r[1]) for r in base_runs), np.int32)
        cons, pw = b"".join(r[1] for r in base_runs), b"".join(r[2] for r in base_runs)
        keep += [cons, pw]
        d.base_cons, d.base_pw = C.cast(C.c_char_p(cons), C.c_void_p), C.cast(C.c_char_p(pw), C.c_void_p)
        self.eng.check(self.eng.lib.t4_index_apply_delta(self.h, C.byref(d)))
        return self

    def size(self):
        return self.eng.lib.t4_index_size(self.h)

    def name(self, i):
        return self.eng.lib.t4_index_seq_name(self.h, i).decode()

    def consensus(self, i):
        return self.eng.lib.t4_index_seq_consensus(self.h, i).decode()

    def hits(self, batch, strand=0, allow_total_skip=0):
        n = batch.n
        off = np.zeros(n + 1, dtype=np.int64)
        self.eng.check(self.eng.lib.t4_hits(self.h, batch.h, strand, allow_total_skip, off.ctypes.data_as(C.c_void_p), None, 0))
        hits = np.zeros(int(off[-1]), dtype=HIT_DTYPE)
        self.eng.check(self.eng.lib.t4_hits(self.h, batch.h, strand, allow_total_skip, off.ctypes.data_as(C.c_void_p),
                                            hits.ctypes.data_as(C.c_void_p), len(hits)))
        return off, hits

    def overlaps(self, batch, strand=0, skip_repeats=0, max_per_read=64, fetch=True):
        n = batch.n
        counts = np.zeros(n, dtype=np.int32)
        out = np.zeros((n, max_per_read), dtype=OV_DTYPE) if fetch else None
        self.eng.check(self.eng.lib.t4_overlaps(self.h, batch.h, strand, skip_repeats, max_per_read,
                                                counts.ctypes.data_as(C.c_void_p),
                                                out.ctypes.data_as(C.c_void_p) if fetch else None))
        return counts, out

    def extend(self, batch, counts, overlaps, mismatch_factor=1.0):
        """overlaps: OV_DTYPE [n, max_per_read] (as returned by overlaps()). -> (ret int32 [n, m], out OV_DTYPE [n, m])."""
        n, m = overlaps.shape
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        overlaps = np.ascontiguousarray(overlaps)
        ret = np.zeros((n, m), dtype=np.int32)
        out = np.zeros((n, m), dtype=OV_DTYPE)
        self.eng.check(self.eng.lib.t4_extend(self.h, batch.h, m, counts.ctypes.data_as(C.c_void_p), overlaps.ctypes.data_as(C.c_void_p),
                                              mismatch_factor, ret.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return ret, out

    def assign(self, batch, strand=0, fetch=True):
        ret = np.zeros(batch.n, dtype=np.int32)
        out = np.zeros(batch.n, dtype=OV_DTYPE) if fetch else None
        self.eng.check(self.eng.lib.t4_assign(self.h, batch.h, strand, ret.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p) if fetch else None))
        return ret, out

    def assign_strands(self, batch, strands):
        """SeqSet::AssignRead with every read's own strand argument (main.cpp:2075-2116) -> (ret int32 [n], out OV_DTYPE [n])"""
        st = np.ascontiguousarray(strands, dtype=np.int32)
        ret = np.zeros(batch.n, dtype=np.int32)
        out = np.zeros(batch.n, dtype=OV_DTYPE)
        self.eng.check(self.eng.lib.t4_assign_strands(self.h, batch.h, st.ctypes.data_as(C.c_void_p), ret.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return ret, out

    def posweight_recompute(self, batch, assign, total_bases, mult=None):
        """SeqSet::RecomputePosWeight (SeqSet.hpp:4705-4738) -> int32 [total_bases, 4], contigs in id order"""
        assign = np.ascontiguousarray(assign, dtype=OV_DTYPE)
        out = np.zeros((total_bases, 4), dtype=np.int32)
        m = None if mult is None else np.ascontiguousarray(mult, dtype=np.int32)
        self.eng.check(self.eng.lib.t4_posweight_recompute(self.h, batch.h, assign.ctypes.data_as(C.c_void_p), None if m is None else m.ctypes.data_as(C.c_void_p),
                                                           out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def consensus_recompute(self, batch, assign, total_bases, mult=None):
        """RecomputePosWeight + UpdateConsensus of every contig (SeqSet.hpp:4705-4738, 4537-4588) -> (int32 [total_bases, 4], bytes of
        the new consensus of all contigs in id order, number of changed bases)"""
        assign = np.ascontiguousarray(assign, dtype=OV_DTYPE)
        out = np.zeros((total_bases, 4), dtype=np.int32)
        cons = C.create_string_buffer(max(total_bases, 1))
        changed = C.c_int64(0)
        m = None if mult is None else np.ascontiguousarray(mult, dtype=np.int32)
        self.eng.check(self.eng.lib.t4_consensus_recompute(self.h, batch.h, assign.ctypes.data_as(C.c_void_p), None if m is None else m.ctypes.data_as(C.c_void_p),
                                                           out.ctypes.data_as(C.c_void_p), out.size, C.cast(cons, C.c_void_p), total_bases, C.byref(changed)))
        return out, cons.raw[:total_bases], int(changed.value)

    def has_hit(self, batch, mode=0):
        """SeqSet::HasHitInSet per read -> int32 [n] of -1 / 0 / 1"""
        out = np.zeros(batch.n, dtype=np.int32)
        self.eng.check(self.eng.lib.t4_has_hit(self.h, batch.h, mode, out.ctypes.data_as(C.c_void_p)))
        return out

    def annotate_rough(self, batch, fetch=True):
        out = np.zeros((batch.n, 4), dtype=OV_DTYPE) if fetch else None
        self.eng.check(self.eng.lib.t4_annotate_rough(self.h, batch.h, out.ctypes.data_as(C.c_void_p) if fetch else None))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.eng.lib.t4_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """t4_batch: 2-bit packed reads in HBM. `reads`: list of str/bytes, or a uint8 [n, stride] array of
    NUL-terminated fixed-stride records (as produced by the synthetic generator)."""

    def __init__(self, eng, reads, barcodes=None):
        self.eng = eng
        if isinstance(reads, np.ndarray):
            arr = np.ascontiguousarray(reads, dtype=np.uint8)
            n, stride = arr.shape
            lens = np.where((arr == 0).any(axis=1), (arr == 0).argmax(axis=1), stride).astype(np.int64)
            offs = np.arange(n + 1, dtype=np.int64) * stride
            if (lens == lens[0]).all() if n else True:
                # offsets[i+1]-offsets[i] must equal the length: repack only when lengths are ragged
                L = int(lens[0]) if n else 0
                buf = np.ascontiguousarray(arr[:, :L]).reshape(-1)
                offs = np.arange(n + 1, dtype=np.int64) * L
            else:
                buf = np.concatenate([arr[i, :lens[i]] for i in range(n)]) if n else np.zeros(0, np.uint8)
                offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        else:
            bs = [r if isinstance(r, bytes) else r.encode() for r in reads]
            n = len(bs)
            buf = np.frombuffer(b"".join(bs), dtype=np.uint8) if n else np.zeros(0, np.uint8)
            offs = np.zeros(n + 1, dtype=np.int64)
            if n:
                offs[1:] = np.cumsum([len(b) for b in bs])
        if len(buf) == 0:
            buf = np.zeros(1, np.uint8)
        self.n = n
        bc = None
        if barcodes is not None:
            barcodes = np.ascontiguousarray(barcodes, dtype=np.int32)
            bc = barcodes.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        eng.check(eng.lib.t4_reads_upload(eng.h, buf.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), bc, n, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.eng.lib.t4_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

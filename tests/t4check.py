"""Shared comparison helpers: engine (HIP or emulated) output vs oracle / golden vectors."""
import os
import subprocess

import numpy as np

from t4libs import ROOT

EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "libt4hip_emu.so")


def build_emulator_lib():
    """g++ build of the SAME kernel sources against tests/hipemu (fiber emulator). Test infra only."""
    csrc = os.path.join(ROOT, "trust4_amd", "csrc")
    units = [os.path.join(csrc, "t4_api.hip"), os.path.join(csrc, "t4_assembler.cpp"), os.path.join(ROOT, "tests", "hipemu", "hip_emu.cpp")]
    deps = units + [os.path.join(csrc, f) for f in ("t4_kernels.h", "t4_wide.h", "t4_device.h", "t4_internal.h")]
    deps += [os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "trust4_hip.h")]
    if os.path.exists(EMU_LIB) and all(os.path.getmtime(EMU_LIB) >= os.path.getmtime(d) for d in deps):
        return EMU_LIB
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-DT4_TEST_KNOBS", "-ffp-contract=off", "-fPIC", "-shared", "-I",
                    os.path.join(ROOT, "tests", "hipemu"), "-o", EMU_LIB, "-x", "c++"] + units + ["-lz", "-lpthread"], check=True)
    return EMU_LIB


def ann_equal(engine_rec, expect):
    """engine_rec: numpy OV_DTYPE record; expect: 9-tuple. Entries with seqIdx == -1 compare on seqIdx only."""
    x = tuple(engine_rec.tolist())
    if expect[0] == -1:
        return x[0] == -1
    return x == tuple(expect)


def check_annotate(ann, reads, oracle):
    bad = []
    for i, rd in enumerate(reads):
        _, g = oracle.annotate_read0(rd)
        for t in range(4):
            if not ann_equal(ann[i, t], g[t]):
                bad.append((i, t, g[t], tuple(ann[i, t].tolist())))
    return bad


def check_overlaps(counts, ov, reads, oracle, skip_repeats=0, strand=0, barcodes=None):
    bad = []
    for i, rd in enumerate(reads):
        bc = -1 if barcodes is None else int(barcodes[i])
        ret, lst = oracle.overlaps_from_read(rd, strand=strand, barcode=bc, skip_repeats=skip_repeats)
        if ret != counts[i]:
            bad.append((i, "count", ret, int(counts[i])))
            continue
        mine = [tuple(x) for x in ov[i, :max(ret, 0)].tolist()]
        if mine != [tuple(x) for x in lst]:
            bad.append((i, "list"))
    return bad


def hits_as_sorted_rows(off, hits, i):
    h = hits[off[i]:off[i + 1]]
    m = np.stack([h["idx"], h["offset"], h["readOffset"], h["strand"], h["repeats"]], axis=1).astype(np.int32)
    return m


def check_hits(off, hits, reads, oracle, strand=0, allow_total_skip=0, barcodes=None):
    """The engine returns hits ordered by (strand, idx, readOffset, offset); the reference's order among equal
    (strand, idx, readOffset) is posting order, which nothing downstream observes -> compare canonicalised."""
    bad = []
    for i, rd in enumerate(reads):
        bc = -1 if barcodes is None else int(barcodes[i])
        ho = oracle.hits(rd, strand=strand, barcode=bc, allow_total_skip=allow_total_skip)
        ho = ho[np.lexsort((ho[:, 1], ho[:, 2], ho[:, 0], ho[:, 3]))] if len(ho) else ho
        m = hits_as_sorted_rows(off, hits, i)
        if ho.shape != m.shape or not (ho == m).all():
            bad.append(i)
    return bad

#!/usr/bin/env python3
"""bench.py -- whole stage-1 throughput on N MI355X of one node (driver contract in the task brief).

A "step" is ONE WHOLE STAGE 1 (`trust4_amd/bin/trust4-hip`: FASTQ in -> ProcessRead / 21-mer counts / sort -> rough
annotation of every read on the GPU -> the order-dependent AddRead pass (host commits + GPU queries against a device
image patched by deltas) -> `_raw.out`, `_assembled_reads.fa`, `_final.out`), process start to exit: the parse of the FASTQ
text, device bring-up and every host phase are INSIDE the timed region (the boundary hands over files, as run-trust4:508 does,
so there is no "inputs resident in HBM" variant of this metric; the kernel-only numbers are in `roofline` / `passes`).

N = 1. The workload is BASELINE.json's config C2 ITSELF (1 M synthetic 150 bp PE pairs, 20 k clones, seed 1, -f hg38_bcrtcr.fa,
bulk mode) whenever `steps + warmup` runs of it fit into the time the WHOLE run of bench.py may take (T4_BENCH_BUDGET_S, default 1770 s from
process start to the printed line: the driver stops bench.py after 1800 s, and its 20 + 5 steps of C2 at 58-61 s per step need
1450-1525 s of that; decided from the first run's own seconds and what has been spent by then): the first run is C2 in
any case -- it is compared with the committed md5 sums of the reference's outputs, it is the first warm-up step when C2 is the
workload, and it is reported as `c2` (with its own roofline block and a same-box reference timing on a stated prefix of the same
files). When the runs do not fit, the steps are timed on the C2 recipe at 100 k pairs (`config.workload` says which ran).
Rank 0 also reports
  cpu_baseline   the reference binary (oracle/_ref/trust4) on the SAME box: on the whole batch of the timed steps when that is the
                 100 k-pair batch (outputs compared byte for byte, `parity_on_bench_batch`), on the first 100 k pairs of C2's files
                 when C2 is the workload (-t <host cores>; `sample` names the files);
  roofline       the dominant kernels of a step (the AddRead query launches), HIP-event time on the engine's stream, bytes as
                 SURVEY 8d defines them, `traffic` from two rocprofv3 PMC passes of the 100 k-pair command;
  passes         kernel-level numbers of the two GPU passes and the rough-annotation pass alone over a resident C2 batch;
  stage1_cells   whole stage 1 in barcode mode (C5 recipe sample) next to the reference binary;
  stage1_cells_1m  the same at 1 M pairs / 10 k cells (-t 32), files against the digests of the reference's run on that sample;
  stage0_e2e     the stage-0 candidate filter next to the reference binary.

N > 1 measures the path that shards: ONE barcode-mode sample of the C5 recipe (`--cells-pairs` pairs and
`--cells` cells per GPU of the job), cells sharded by rank through `trust4-hip --cellShard R/N --rcclId FILE` -- every rank splits
the text of the sample into records but builds, ProcessReads and counts the pairs of its own cells only (the cells follow from the
barcode file; the ranks' 21-mer tables are put together in one all-gather), then statistics, sort, rough annotation, barcode-wise
counts and the cell pass on those; no exchange during assembly, the contig records gathered to rank 0 inside the engine at the
end, every rank writes its slice of the reads. Strong scaling: the sample is fixed by N, `value` = its pairs / the slowest
rank's wall time. Rank 0 then runs the same sample on one rank and compares the md5 sums of the three output files (`one_rank`),
and reports every rank's seconds in the replicated phases, in the phases on its own cells and in the cell pass (`config.per_rank_s`).
Bulk-mode stage 1 itself is one ordered chain (replicas only): it is what N = 1 measures.
"""
import argparse
import filecmp
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

T_PROC0 = time.time()   # (the driver's clock runs from the start of the process: the interpreter's own start-up is a fraction of a second before this line)
try:
    import psutil
    T_PROC0 = psutil.Process().create_time()
except Exception:   # noqa: BLE001
    pass

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
DRIVER = os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "trust4")
OUT_SUFFIXES = ("_raw.out", "_assembled_reads.fa", "_final.out")


def annotate_bytes(n_reads, read_len, total_hits):
    """ceil(L/4) + ceil(L/8) + 8*H + 4*32 per read."""
    return n_reads * ((read_len + 3) // 4 + (read_len + 7) // 8 + 128) + 8 * total_hits


def add_bytes(n_queries, read_len, total_hits):
    """SURVEY.md 8(d), AddRead pass: the same formula with its own H and + 16 * L for the posWeight read-modify-write."""
    return n_queries * ((read_len + 3) // 4 + (read_len + 7) // 8 + 128 + 16 * read_len) + 8 * total_hits


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def make_batch(tmp, pairs, clones, seed):
    """FASTQ files of one batch (t4synth: the SURVEY 8d recipe) + the plain-text gene FASTA."""
    fa = os.path.join(tmp, "ref.fa")
    import t4libs
    if not os.path.exists(fa):
        with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
            shutil.copyfileobj(f, g)
    pre = os.path.join(tmp, "b%d" % seed)
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), str(clones), str(seed), pre], check=True, stdout=subprocess.DEVNULL)
    return fa, pre + "_1.fq", pre + "_2.fq"


def run_stage1(fa, f1, f2, out, threads, device, stats=None, extra=()):
    env = dict(os.environ, T4_DEVICE=str(device))
    if stats:
        env["T4_STATS_JSON"] = stats
    t0 = time.perf_counter()
    p = subprocess.run([DRIVER, "-t", str(threads), "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", out] + list(extra),
                       env=env, stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0
    if p.returncode:
        raise SystemExit("trust4-hip failed (%d):\n%s" % (p.returncode, "\n".join(p.stderr.strip().split("\n")[-10:])))
    return dt


def head_fastq(src, dst, pairs):
    with open(src) as f, open(dst, "w") as g:
        for i, line in enumerate(f):
            if i >= 4 * pairs:
                break
            g.write(line)


def cpu_baseline(tmp, fa, f1, f2, pairs, mine_prefix, single_prefix_pairs):
    """The reference's pthreads CPU path on the same files (all host cores), outputs compared with the GPU run's; and its
    single-thread time on a prefix of the same files."""
    cores = host_cores()
    t0 = time.perf_counter()
    subprocess.run([REF_BIN, "-t", str(cores), "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", os.path.join(tmp, "ref")],
                   check=True, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    same = all(filecmp.cmp(os.path.join(tmp, "ref" + x), mine_prefix + x, shallow=False) for x in OUT_SUFFIXES)
    out = {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "reference",
           "sample": "oracle/_ref/trust4 -t %d --skipMateExtension on the whole batch of the timed steps (%d pairs), %.1f s wall" % (cores, pairs, dt),
           "seconds": dt}
    if single_prefix_pairs > 0:
        n1 = min(single_prefix_pairs, pairs)
        h1, h2 = os.path.join(tmp, "p_1.fq"), os.path.join(tmp, "p_2.fq")
        head_fastq(f1, h1, n1)
        head_fastq(f2, h2, n1)
        t0 = time.perf_counter()
        subprocess.run([REF_BIN, "-t", "1", "--skipMateExtension", "-f", fa, "-1", h1, "-2", h2, "-o", os.path.join(tmp, "ref1")],
                       check=True, stderr=subprocess.DEVNULL)
        d1 = time.perf_counter() - t0
        out["single_thread"] = {"value": n1 / d1, "unit": "pairs/s", "cores": 1, "sample": "the first %d pairs of the same files, -t 1, %.1f s wall" % (n1, d1)}
    return out, same


def annotate_pass_c2(device, pairs, clones, steps):
    """The data-parallel pass alone over a resident C2-sized batch: stage-1 rough annotation of every read
    (main.cpp:1084-1120), inputs 2-bit packed in HBM before the timed region, results left on the device."""
    import t4libs
    import trust4_amd
    eng = trust4_amd.Engine(device)
    ref = eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(t4libs.REF_FA).commit()
    reads = t4libs.Synth(clones, 1).next_reads(pairs)
    n_reads = reads.shape[0]
    batch = eng.upload(reads)
    ref.annotate_rough(batch, fetch=False)
    eng.check(eng.lib.t4_sync(eng.h))
    t0 = time.perf_counter()
    kms, hits = [], 0
    for _ in range(steps):
        ref.annotate_rough(batch, fetch=False)
        st = eng.stats()
        kms.append(st["chain_kernel_ms"])
        hits = st["total_hits"]
    eng.check(eng.lib.t4_sync(eng.h))
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(kms))
    alg = annotate_bytes(n_reads, 150, hits)
    return {"workload": "C2: %d pairs (%d reads), %d clones, seed 1; rough annotation of every read, inputs resident, results on the device" % (pairs, n_reads, clones),
            "reads_per_s": n_reads * steps / dt, "ms_per_pass": dt / steps * 1e3, "kernel_ms": k_ms, "hits_per_read": hits / n_reads,
            "tier_reads": eng.stats()["tier_reads"],
            "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "t4k::queryKernel<.., 0> (all tiers of one pass)", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg}}


def file_md5(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def add_roofline(aq, note=""):
    """roofline block of the AddRead query launches of one whole stage 1, from the engine's own counters (stats JSON)"""
    alg = add_bytes(aq["reads_queried"], 150, aq["hits"])
    alg1 = add_bytes(aq["reads_served"], 150, int(aq["hits"] * aq["reads_served"] / max(1, aq["reads_queried"])))
    sec = max(aq["kernel_ms"], 1e-6) * 1e-3
    return {"bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / sec / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "kernel": "t4k::queryKernel<8192, 512, 512, 1> mode 4 (AddRead query: seed->sort->chain->score->ExtendOverlap) + the wide query's kernels "
                      "(t4k::wide{Scatter,Sort,Stats,Chain,Merge}Kernel, reads beyond the LDS tier spread over the chip) + t4k::extendKernel behind them "
                      "(kernel_ms brackets a round's launches), all %d query rounds of one step" % aq["rounds"],
            "kernel_ms": aq["kernel_ms"], "launch_ms_avg": aq["kernel_ms"] / max(1, aq["rounds"]),
            "algorithmic_bytes_per_step": alg, "reads_queried": aq["reads_queried"], "reads_served": aq["reads_served"], "hits": aq["hits"],
            "algorithmic_bytes_per_step_reads_served_only": alg1, "frac_reads_served_only": alg1 / sec / 1e9 / HBM_PEAK_GBS,
            "note": "a latency-bound chain of dependent launches (DESIGN 3b): bytes / kernel time says how little of the HBM rate a dependent round can use; "
                    "`achieved` counts every query the engine ran (re-queries of invalidated window entries included), `frac_reads_served_only` scales the "
                    "bytes to one query per served read as the reference does" + note}


def chain_block(st):
    """The chain of dependent query rounds of one stage 1 (stats JSON `chain`, t4_assembler_chain_stats) with its floor: what the
    Add pass would cost if EVERY round were as short as the shortest ones are (5th percentile of launch call -> results on the host)
    -- rounds only end when a commit has changed what the next read must be matched against, so this is the part of the pass that no
    amount of chip can shorten at bit-identity; the distance between the pass and it is what the engine still owes."""
    ch = st.get("chain")
    if not ch:
        return None
    ch = dict(ch)
    ch["floor_s"] = ch["rounds"] * ch["round_wall_ms_p05"] * 1e-3
    ch["addread_pass_s"] = st["phases_s"]["assembled"] - st["phases_s"]["trimmed_ready"]
    ch["whole_queries_per_served_read"] = ch["whole_queries"] / max(1, st["add_query"]["reads_served"])
    return ch


def config_leg(name, threads, device, mode="skipMateExtension", keep=None, cpu_prefix_pairs=0):
    """One whole stage 1 through trust4-hip on a BASELINE config itself (C2 = 1 M pairs, 20 k clones, seed 1; `c3p*` = a stated
    prefix of C3's read stream), under this run's clock, outputs compared with the md5 sums of the REFERENCE's outputs on the same
    files (tests/golden/c2_digests.json, produced by tools/c2_digests.py from oracle/_ref/trust4; the input files are
    regenerated here and their md5 sums are checked too)."""
    name, _, variant = name.partition(":")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import c2_digests as _cd
    if name in _cd.CELL_CONFIGS:
        return cell_config_leg(name, device)
    if variant == "dropin":   # the reference's main.cpp bound to the C ABI (integration/), run-trust4's DEFAULT options: mate-pair extension tail included
        mode = "default"
    tmp = keep or tempfile.mkdtemp(prefix="t4%s_" % name)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import c2_digests
        golden = json.load(open(c2_digests.OUT)).get(name)
        fa, f1, f2, n = c2_digests.make_inputs(tmp, name)
        pairs, clones, seed, prefix = c2_digests.CONFIGS[name]
        out = {"workload": "config %s: %d synthetic 150 bp PE pairs%s, %d clones, seed %d, -f hg38_bcrtcr.fa, bulk mode; whole stage 1 through trust4-hip -t %d%s, "
                           "FASTQ files in -> three files out, process start to exit" % (name.upper(), n, " (the first %d of the config's %d)" % (n, pairs) if prefix else "",
                                                                                          clones, seed, threads, " --skipMateExtension" if mode == "skipMateExtension" else ""),
               "pairs": n}
        inputs_ok = golden is not None and [file_md5(f1), file_md5(f2)] == golden["inputs_md5"]
        mine, stats_path = os.path.join(tmp, "mine"), os.path.join(tmp, "stats.json")
        env = dict(os.environ, T4_DEVICE=str(device), T4_STATS_JSON=stats_path)
        exe = os.path.join(ROOT, "oracle", "_ref", "trust4-dropin") if variant == "dropin" else DRIVER
        if variant == "dropin":
            out["workload"] = out["workload"].replace("through trust4-hip", "through oracle/_ref/trust4-dropin (the reference's main.cpp with its three hot loops bound to libt4hip.so, integration/; default options: mate-pair extension tail on the host)")
        argv = [exe, "-t", str(threads)] + (["--skipMateExtension"] if mode == "skipMateExtension" else []) + ["-f", fa, "-1", f1, "-2", f2, "-o", mine]
        t0 = time.perf_counter()
        p = subprocess.run(argv, env=env, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        if p.returncode:
            out["error"] = "%s exit %d: %s" % (os.path.basename(exe), p.returncode, " | ".join(p.stderr.strip().split("\n")[-3:]))[:400]
            return out
        md5s = {x: file_md5(mine + x) for x in OUT_SUFFIXES}
        out.update({"seconds": dt, "pairs_per_s": n / dt, "md5": md5s})
        if variant != "dropin":
            st = json.load(open(stats_path))
            aq = st["add_query"]
            out.update({"contigs": st["contigs"], "assembled_reads": st["assembled_reads"],
                        "rounds": aq["rounds"], "reads_queried": aq["reads_queried"], "reads_served": aq["reads_served"], "invalidations": aq["invalidations"],
                        "kernel_ms": aq["kernel_ms"], "launch_ms_avg": aq["kernel_ms"] / max(1, aq["rounds"]), "hits": aq["hits"],
                        "host_wait_for_queries_s": aq.get("host_wait_for_queries_s"), "wide": aq.get("wide"),
                        "addread_pass_s": st["phases_s"]["assembled"] - st["phases_s"]["trimmed_ready"], "phases_s": st["phases_s"],
                        "roofline": add_roofline(aq), "chain": chain_block(st)})
            out["files"] = (fa, f1, f2, mine, stats_path)
        if golden is None or mode not in golden.get("modes", {}):
            out["identical"] = None
            out["note"] = "no reference digests committed for this config / mode"
        else:
            g = golden["modes"][mode]
            out["identical"] = bool(inputs_ok and all(md5s[x] == g["md5"][x] for x in OUT_SUFFIXES))
            out["inputs_identical"] = bool(inputs_ok)
            out["reference_digest_run"] = {"seconds": g["reference_seconds"], "threads": g["reference_threads"], "pairs_per_s": n / g["reference_seconds"],
                                           "where": "the builder's container (tools/c2_digests.py), not this box: a digest, not a baseline"}
        if cpu_prefix_pairs > 0 and os.path.exists(REF_BIN):   # the reference on THIS box, on a stated prefix of the same files
            np_ = min(cpu_prefix_pairs, n)
            h1, h2 = os.path.join(tmp, "cpu_1.fq"), os.path.join(tmp, "cpu_2.fq")
            head_fastq(f1, h1, np_)
            head_fastq(f2, h2, np_)
            cores = host_cores()
            t0 = time.perf_counter()
            subprocess.run([REF_BIN, "-t", str(cores), "--skipMateExtension", "-f", fa, "-1", h1, "-2", h2, "-o", os.path.join(tmp, "cpuref")], check=True, stderr=subprocess.DEVNULL)
            dc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": np_ / dc, "unit": "pairs/s", "cores": cores, "kind": "reference", "seconds": dc,
                                   "sample": "oracle/_ref/trust4 -t %d --skipMateExtension on the first %d pairs of config %s's own files (%s, %s), on this box, %.1f s wall"
                                             % (cores, np_, name.upper(), os.path.basename(f1), os.path.basename(f2), dc)}
        return out
    except Exception as e:   # noqa: BLE001  (a side leg never takes the bench line down)
        return {"error": repr(e)[:300]}
    finally:
        if not keep:
            shutil.rmtree(tmp, ignore_errors=True)


def cell_config_leg(name, device, threads=32):
    """A barcode-mode config (tools/c2_digests.py CELL_CONFIGS: the C5 recipe at a stated size) through trust4-hip on one GPU, outputs
    against the md5 sums of the reference's run on the same files."""
    import c2_digests
    tmp = tempfile.mkdtemp(prefix="t4%s_" % name)
    try:
        golden = json.load(open(c2_digests.OUT)).get(name)
        fa, f1, f2, bc, umi, n = c2_digests.make_cell_inputs(tmp, name)
        cells = c2_digests.CELL_CONFIGS[name][1]
        out = {"workload": "C5 recipe at %d synthetic 150 bp PE pairs, %d cells x 2 clones, barcode + UMI files; whole stage 1 through trust4-hip -t %d, process start to exit" % (n, cells, threads), "pairs": n}
        inputs_ok = golden is not None and [file_md5(x) for x in (f1, f2, bc, umi)] == golden["inputs_md5"]
        mine, stats_path = os.path.join(tmp, "mine"), os.path.join(tmp, "stats.json")
        t0 = time.perf_counter()
        p = subprocess.run([DRIVER, "-t", str(threads), "-f", fa, "-1", f1, "-2", f2, "--barcode", bc, "--UMI", umi, "-o", mine],
                           env=dict(os.environ, T4_DEVICE=str(device), T4_STATS_JSON=stats_path), stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        if p.returncode:
            out["error"] = "trust4-hip exit %d: %s" % (p.returncode, " | ".join(p.stderr.strip().split("\n")[-3:]))[:400]
            return out
        md5s = {x: file_md5(mine + x) for x in OUT_SUFFIXES}
        out.update({"seconds": dt, "pairs_per_s": n / dt, "md5": md5s, "phases_s": json.load(open(stats_path)).get("phases_s")})
        if golden:
            g = golden["modes"]["barcode"]
            out["identical"] = bool(inputs_ok and all(md5s[x] == g["md5"][x] for x in OUT_SUFFIXES))
            out["reference_digest_run"] = {"seconds": g["reference_seconds"], "threads": g["reference_threads"], "pairs_per_s": n / g["reference_seconds"],
                                           "where": "the builder's container (tools/c2_digests.py --config %s), not this box: a digest, not a baseline" % name}
        return out
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


ADD_KERNELS = ("queryKernel<8192, 512, 512, 1>", "wideSeedKernel", "wideScatterKernel", "wideSortKernel", "wideStatsKernel", "wideChainKernel", "wideMergeKernel", "extendKernel")   # == tools/pmc_c2_summary.py


def pmc_traffic(fa, f1, f2, threads, device, tmp, kernels=ADD_KERNELS, limit_s=None):
    """HBM-side bytes of the dominant kernels of ONE step, from rocprofv3's TCC counters: the step's own command run twice more
    under `rocprofv3 --pmc <C> --kernel-trace` (FETCH_SIZE and WRITE_SIZE in separate passes, as the counter slots demand),
    the counter summed over every launch of the query kernels. Correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts
    128-byte requests at 64 bytes on gfx950, so it is doubled; WRITE_SIZE is taken as is (uncalibrated for this access pattern).
    Counter units are KB. Infinity-Cache hits are counted too: this is L2 <-> fabric traffic, an upper bound of DRAM traffic."""
    import csv
    import glob
    import re
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, "pmc_" + counter)
        cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               DRIVER, "-t", str(threads), "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", os.path.join(tmp, "pmcrun")]
        try:
            p = subprocess.run(cmd, env=dict(os.environ, T4_DEVICE=str(device), TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                               timeout=limit_s / 2 if limit_s else None)
        except subprocess.TimeoutExpired:
            return None, {"error": "rocprofv3 --pmc %s did not end within the %.0f s the run's budget left for it" % (counter, limit_s / 2)}
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode or not files:
            return None, {"error": "rocprofv3 --pmc %s failed (%d): %s" % (counter, p.returncode, p.stderr.strip().split("\n")[-1][:200])}
        tot, launches = 0.0, 0
        per_kernel = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"])
                acc = per_kernel.setdefault(name, [0, 0.0])
                acc[0] += 1
                acc[1] += float(row["Counter_Value"])
                if any(k in name for k in kernels):
                    tot += float(row["Counter_Value"])
                    launches += 1
        raw[counter] = {"counter_units_KB": tot, "launches": launches}
        keep = os.environ.get("T4_BENCH_PMC_DIR")   # the per-kernel sums this number comes from, kept for profiles/
        if keep:
            os.makedirs(keep, exist_ok=True)
            with open(os.path.join(keep, "step_pmc_%s.txt" % counter), "w") as g:
                g.write("# rocprofv3 --pmc %s --kernel-trace on one step of bench.py (trust4-hip, the bench batch); counter units: KB; per kernel: launches, sum\n" % counter)
                for name, (cnt, val) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
                    g.write("%-110s %8d %16.0f\n" % (name, cnt, val))
        shutil.rmtree(d, ignore_errors=True)
    traffic = (2.0 * raw["FETCH_SIZE"]["counter_units_KB"] + raw["WRITE_SIZE"]["counter_units_KB"]) * 1024.0
    return traffic, {"raw": raw, "fetch_bytes_corrected": 2.0 * raw["FETCH_SIZE"]["counter_units_KB"] * 1024.0, "write_bytes": raw["WRITE_SIZE"]["counter_units_KB"] * 1024.0,
                     "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes) on the step's command; FETCH_SIZE x 2 (gfx950), units KB; the AddRead query launches of one step (the query kernel, the wide query's six kernels, extendKernel)"}


def stage1_cells(pairs, cells):
    """Whole stage 1 in barcode mode through trust4-hip vs oracle/_ref/trust4 (when it travelled) on the same files."""
    tmp = tempfile.mkdtemp()
    try:
        import t4libs
        fa = os.path.join(tmp, "ref.fa")
        with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
            shutil.copyfileobj(f, g)
        pre = os.path.join(tmp, "c5")
        subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", "4", pre, "--cells", str(cells)], check=True, stdout=subprocess.DEVNULL)
        argv = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
        cores = min(8, host_cores())   # same host-thread budget for both programs
        t0 = time.perf_counter()
        subprocess.run([DRIVER, "-t", str(cores)] + argv + ["-o", os.path.join(tmp, "mine")], check=True, stderr=subprocess.DEVNULL)
        t_mine = time.perf_counter() - t0
        out = {"workload": "C5 recipe sample: %d synthetic 150 bp PE pairs, %d cells x 2 clones, barcode + UMI files; FASTQ in -> _raw.out/_final.out/_assembled_reads.fa out" % (pairs, cells),
               "pairs_per_s": pairs / t_mine, "seconds": t_mine, "host_threads": cores}
        if os.path.exists(REF_BIN):
            t0 = time.perf_counter()
            subprocess.run([REF_BIN, "-t", str(cores)] + argv + ["-o", os.path.join(tmp, "ref")], check=True, stderr=subprocess.DEVNULL)
            t_ref = time.perf_counter() - t0
            out.update({"reference_pairs_per_s": pairs / t_ref, "reference_seconds": t_ref, "reference_threads": cores,
                        "identical": all(filecmp.cmp(os.path.join(tmp, "ref" + x), os.path.join(tmp, "mine" + x), shallow=False) for x in OUT_SUFFIXES)})
        return out
    except Exception as e:   # noqa: BLE001  (a side leg never takes the bench line down)
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def stage1_cells_1m():
    """Barcode mode at the size VERDICT r3 #5 sets its bar on: the C5 recipe at 1 M pairs / 10 k cells through trust4-hip -t 32, files
    against the digests of the reference's run on that sample (tests/golden/c2_digests.json: c5_1m; the reference itself takes 80 s
    at -t 32 and is not run again here)."""
    tmp = tempfile.mkdtemp()
    try:
        import t4libs
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "c2_digests.json")))["c5_1m"]
        fa = os.path.join(tmp, "ref.fa")
        with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
            shutil.copyfileobj(f, g)
        pre = os.path.join(tmp, "c5")
        subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(golden["pairs"]), "0", str(golden["seed"]), pre, "--cells", str(golden["cells"])], check=True, stdout=subprocess.DEVNULL)
        argv = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
        cores = min(32, host_cores())
        stats = os.path.join(tmp, "stats.json")
        runs = []
        for _ in range(2):   # (the first run of a box pays for the first 13 GB table and the first launches)
            t0 = time.perf_counter()
            subprocess.run([DRIVER, "-t", str(cores)] + argv + ["-o", os.path.join(tmp, "mine")], check=True, stderr=subprocess.DEVNULL, env=dict(os.environ, T4_STATS_JSON=stats))
            runs.append(time.perf_counter() - t0)
        ph = json.load(open(stats))["phases_s"]
        return {"workload": "C5 recipe sample: %d synthetic 150 bp PE pairs, %d cells x 2 clones, barcode + UMI files; trust4-hip -t %d, FASTQ in -> three files out, process start to exit" % (golden["pairs"], golden["cells"], cores),
                "seconds": min(runs), "seconds_first_run": runs[0], "pairs_per_s": golden["pairs"] / min(runs), "host_threads": cores,
                "identical": all(file_md5(os.path.join(tmp, "mine" + x)) == golden["md5"][x] for x in OUT_SUFFIXES),
                "phases_s": {"parse_processread_21mers": ph["input_processed_counted"], "sort": ph["sorted"] - ph["input_processed_counted"],
                             "rough_annotation": ph["rough_annotation"] - ph["sorted"], "barcode_counts_trim": ph["trimmed_ready"] - ph["rough_annotation"],
                             "cell_pass": ph["assembled"] - ph["trimmed_ready"], "outputs": ph["outputs_written"] - ph["assembled"]},
                "reference_seconds_round3": 79.8}
    except Exception as e:   # noqa: BLE001  (a side leg never takes the bench line down)
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def stage0_e2e(pairs, receptor_fraction=0.02):
    """Stage-0 candidate filter (fastq-extractor-hip) vs oracle/_ref/fastq-extractor (when it travelled) on the same FASTQ files."""
    tmp = tempfile.mkdtemp()
    try:
        import t4libs
        fa = os.path.join(tmp, "ref.fa")
        with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
            shutil.copyfileobj(f, g)
        nrec = int(pairs * receptor_fraction)
        r1, r2 = t4libs.Synth(2000, 1).next_pairs(nrec)
        rnd = np.random.RandomState(5)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        gens = [acgt[rnd.randint(0, 4, size=(pairs - nrec, 150))] for _ in range(2)]
        order = rnd.permutation(pairs)
        for name, rec, gen in (("in_1.fq", r1, gens[0]), ("in_2.fq", r2, gens[1])):
            rows = [bytes(x[:150]) for x in rec] + [x.tobytes() for x in gen]
            with open(os.path.join(tmp, name), "wb") as f:
                q = b"F" * 150
                for i in order:
                    f.write(b"@q%d\n%s\n+\n%s\n" % (i, rows[i], q))
        argv = ["-f", fa, "-1", os.path.join(tmp, "in_1.fq"), "-2", os.path.join(tmp, "in_2.fq")]
        t0 = time.perf_counter()
        subprocess.run([os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip")] + argv + ["-o", os.path.join(tmp, "mine")], check=True, stderr=subprocess.DEVNULL)
        t_mine = time.perf_counter() - t0
        out = {"workload": "%d synthetic 150 bp PE pairs, %.0f %% receptor pairs in random pairs; FASTQ in -> candidate _1.fq/_2.fq out" % (pairs, 100 * receptor_fraction),
               "pairs_per_s": pairs / t_mine, "seconds": t_mine, "host_threads": 1}
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "fastq-extractor")
        if os.path.exists(ref_bin):
            cores = min(8, host_cores())
            t0 = time.perf_counter()
            subprocess.run([ref_bin, "-t", str(cores)] + argv + ["-o", os.path.join(tmp, "ref")], check=True, stderr=subprocess.DEVNULL)
            t_ref = time.perf_counter() - t0
            out.update({"reference_pairs_per_s": pairs / t_ref, "reference_seconds": t_ref, "reference_threads": cores,
                        "identical": all(filecmp.cmp(os.path.join(tmp, "ref" + x), os.path.join(tmp, "mine" + x), shallow=False) for x in ("_1.fq", "_2.fq"))})
        return out
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sharded_cells(args, rank, local_rank, world, dist):
    """N > 1: one C5-recipe sample, cells sharded by rank, one RCCL all-gather inside the engine at the end (strong scaling)."""
    import torch
    import trust4_amd.dist as t4dist
    dry = bool(os.environ.get("T4_BENCH_CPU_DRYRUN"))   # tests/test_dist_gloo.py: the same plumbing over gloo, the emulated driver and the file transport
    dev = "cpu" if dry else "cuda"
    driver = os.environ.get("T4_DRIVER", DRIVER) if dry else DRIVER

    def sync():
        if not dry:
            torch.cuda.synchronize()
    pairs, cells = args.cells_pairs * world, (args.cells_total if args.cells_total > 0 else args.cells * world)
    tmp = os.path.join(tempfile.gettempdir(), "t4bench_cells_%s" % os.environ.get("MASTER_PORT", "0"))
    fa, pre = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "c5")
    if rank == 0:
        import t4libs
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
            shutil.copyfileobj(f, g)
        subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", "4", pre, "--cells", str(cells)], check=True, stdout=subprocess.DEVNULL)
    dist.barrier()
    argv = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
    threads = max(1, min(args.cells_threads, host_cores() // world))
    out = os.path.join(tmp, "sharded")
    step_no = [0]

    def one_step(timed):
        # a fresh directory per step for the communicator id, the status files and (should RCCL not come up on every rank: the engine's
        # automatic fall-back, trust4_main.cpp commUp) the files of the exchange: nothing of an earlier step can be taken for this one's
        xdir = os.path.join(tmp, "xfer%d" % step_no[0])
        step_no[0] += 1
        if rank == 0:
            shutil.rmtree(xdir, ignore_errors=True)
            os.makedirs(xdir)
        sync()
        dist.barrier()
        stats = os.path.join(tmp, "stats_rank%d.json" % rank)
        env = dict(os.environ, T4_DEVICE="0" if dry else str(local_rank), T4_STATS_JSON=stats)
        # (the CPU dry run takes the same command line: the emulated engine has no RCCL, so its t4_comm_init fails on every rank and the
        # run goes through the fall-back -- which is what the dry run is there to cover)
        transport = ["--rcclId", os.path.join(xdir, "rcclid")]
        t0 = time.perf_counter()
        p = subprocess.run([driver, "-t", str(threads)] + argv + ["-o", out, "--cellShard", "%d/%d" % (rank, world)] + transport, env=env, stderr=subprocess.PIPE, text=True)
        if p.returncode:
            raise SystemExit("rank %d: trust4-hip --cellShard failed (%d): %s" % (rank, p.returncode, " | ".join(p.stderr.strip().split("\n")[-4:])))
        sync()
        dist.barrier()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        one_step(False)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    sync()
    dist.barrier()
    dt = t4dist.max_over_ranks(dist, time.perf_counter() - t0, dev)
    # this rank's seconds, gathered for the report: up to the point where the sample's 21-mer counts are complete on the rank (the text of
    # the whole sample split into records on every rank, ProcessRead and the counts of its own cells, the exchange of the ranks' tables:
    # `replicated_phases` -- only the parse still is), the phases before the cell pass on its own cells (count statistics,
    # sort, rough annotation, barcode-wise counts, trimming) and the cell pass itself
    rep = own = add = 0.0
    try:
        ph = json.load(open(os.path.join(tmp, "stats_rank%d.json" % rank)))["phases_s"]
        rep = ph.get("counted_all_reads", ph["trimmed_ready"])
        own, add = ph["trimmed_ready"] - rep, ph["assembled"] - ph["trimmed_ready"]
    except Exception:   # noqa: BLE001
        pass
    t = torch.tensor([rep, add, own], dtype=torch.float64, device=dev)
    parts = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(parts, t)
    if rank == 0:
        line = {"metric": "stage-1 assembly read pairs/sec (150 bp PE), whole stage 1, barcode mode, cells sharded over the GPUs", "value": pairs * args.steps / dt,
                "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "int32/u64", "data": "synthetic",
                "config": {"workload": "C5 recipe sample: %d synthetic 150 bp PE pairs, %d cells x 2 clones, barcode + UMI files; ONE sample over %d GPUs: every rank runs trust4-hip -t %d "
                                       "--cellShard R/%d --rcclId (every rank splits the text of the whole sample into records; ProcessRead, the 21-mer counts -- the ranks' tables put together in one all-gather --, "
                                       "count statistics, sort, rough annotation, barcode-wise counts and the Add pass on a contiguous range of cells per rank; the contig records gathered to rank 0 "
                                       "inside the engine, every rank writes its slice of the reads); "
                                       "process start to exit of the slowest rank"
                                       % (pairs, cells, world, threads, world),
                           "pairs": pairs, "cells": cells, "host_threads_per_rank": threads,
                           "per_rank_s": {"replicated_phases": [float(x[0]) for x in parts], "own_cells_before_the_add_pass": [float(x[2]) for x in parts],
                                          "add_pass_of_its_cells": [float(x[1]) for x in parts]}}}
        try:   # "RCCL", or "files (fallback from RCCL: <first rank's reason>)" when the communicator did not come up on every rank
            line["config"]["transport"] = open(os.path.join(tmp, "stats_rank0.json.transport")).read().strip()
        except OSError:
            line["config"]["transport"] = None
        md5s = {x: file_md5(out + x) for x in OUT_SUFFIXES}
        one = os.path.join(tmp, "one_rank")
        t1 = time.perf_counter()
        p = subprocess.run([driver, "-t", str(threads)] + argv + ["-o", one], env=dict(os.environ, T4_DEVICE="0" if dry else str(local_rank)), stderr=subprocess.PIPE, text=True)
        d1 = time.perf_counter() - t1
        if p.returncode:
            line["one_rank"] = {"error": p.stderr.strip().split("\n")[-1][:300]}
        else:
            line["one_rank"] = {"seconds": d1, "pairs_per_s": pairs / d1, "identical": all(file_md5(one + x) == md5s[x] for x in OUT_SUFFIXES),
                                "speedup_of_the_sharded_run": d1 / (dt / args.steps), "note": "the same sample through one trust4-hip on GPU 0 after the timed region; md5 of the three output files compared"}
        line["md5"] = md5s
        print(json.dumps(line))
    dist.barrier()
    if rank == 0:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=0, help="read pairs per step: 0 = config C2 itself (1 M pairs) when steps + warmup runs of it fit the budget, else the C2 recipe at 100 k pairs; "
                                                         "a number forces the C2 recipe at that size")
    ap.add_argument("--fallback-pairs", type=int, default=100000, help="pairs of the C2-recipe batch the steps are timed on when C2 itself does not fit")
    ap.add_argument("--clones", type=int, default=0, help="clones of the batch (default: pairs / 50, the C2 ratio)")
    ap.add_argument("--threads", type=int, default=8, help="host threads of trust4-hip (-t)")
    ap.add_argument("--budget", type=float, default=float(os.environ.get("T4_BENCH_BUDGET_S", "1770")),
                    help="seconds the WHOLE run may take, process start to the printed line (the driver stops bench.py after 1800 s): decides whether C2 itself is the workload and which side legs run")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 = skip the reference legs")
    ap.add_argument("--cpu-single-pairs", type=int, default=20000, help="prefix timed with the reference's -t 1 (0 = skip)")
    ap.add_argument("--cpu-c2-pairs", type=int, default=100000, help="prefix of C2's files the reference is timed on, on this box (0 = skip)")
    ap.add_argument("--side-legs", type=int, default=1, help="0 = skip passes.rough_annotation_c2 / stage1_cells / stage0_e2e")
    ap.add_argument("--traffic", type=int, default=1, help="0 = skip the two rocprofv3 PMC passes that measure roofline.traffic")
    ap.add_argument("--c2", type=int, default=1, help="0 = skip the run of config C2 itself (then the steps are timed on the 100 k-pair batch)")
    ap.add_argument("--config-leg", default="", help="additional config legs, comma separated (c3p05, c3p2, c3p5: prefixes of C3; c5m5 / c5m20: the C5 recipe at 5 M / 20 M pairs over 50 k cells, barcode mode; c2:dropin = C2 with default options through the reference's main.cpp bound to the C ABI)")
    ap.add_argument("--cells-pairs", type=int, default=250000, help="N > 1: pairs of the barcode-mode sample per GPU of the job")
    ap.add_argument("--cells", type=int, default=2500, help="N > 1: cells of the sample per GPU of the job")
    ap.add_argument("--cells-total", type=int, default=0, help="N > 1: cells of the whole sample instead of --cells per GPU (fewer cells than ranks: ranks without a cell)")
    ap.add_argument("--cells-threads", type=int, default=16, help="N > 1: host threads per rank")
    args = ap.parse_args()

    import torch
    import trust4_amd.build
    import trust4_amd.dist as t4dist
    import t4libs
    rank, local_rank, world = t4dist.env_rank()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # T4_BENCH_CPU_DRYRUN: the plumbing without GPUs (CPU test suite, never a measurement) -- N > 1: gloo, the emulated driver, the file
    # transport; N = 1: the emulated driver ($T4_DRIVER) on a stand-in for C2, so that the choice of the workload and the order of the legs are tested
    dry = bool(os.environ.get("T4_BENCH_CPU_DRYRUN"))
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    if not dry:
        torch.cuda.set_device(local_rank)
    elif world == 1:
        global DRIVER
        DRIVER = os.environ["T4_DRIVER"]
        args.side_legs = 0
        if not os.environ.get("T4_BENCH_DRY_TRAFFIC"):   # (set: the PMC legs are walked too -- rocprofv3 fails without a GPU, which is a path of its own)
            args.traffic = 0
    dist = t4dist.init("gloo" if dry else "nccl")   # RCCL: barrier + max-reduce of the timed interval (the data-path collective of the sharded run is inside the engine)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    if rank == 0 and not dry:
        trust4_amd.build.build()
    if rank == 0:
        t4libs.build_checkers()
    if dist:
        dist.barrier()
    if world > 1:
        sharded_cells(args, rank, local_rank, world, dist)
        dist.barrier()
        dist.destroy_process_group()
        return

    tmp = tempfile.mkdtemp(prefix="t4bench_")
    try:
        threads = max(1, min(args.threads, host_cores()))

        def spent():
            return time.time() - T_PROC0
        c2 = None
        use_c2 = False
        c2_cpu = None
        c2name = "c2"
        if args.pairs == 0 and args.c2:
            # config C2 itself, once: digest check, the first warm-up step if C2 is the workload, the `c2` record either way
            c2dir = os.path.join(tmp, "c2")
            os.makedirs(c2dir)
            c2name = os.environ.get("T4_BENCH_C2_STANDIN", "c2")   # testing aid: "c2mini" walks the same code in seconds (the line then says it is not config C2)
            c2 = config_leg(c2name, threads, local_rank, keep=c2dir)
            # the reference on this box on a prefix of C2's own files, BEFORE the steps: the line carries its cpu_baseline whatever the
            # steps leave of the budget (about 30 s; nothing else runs beside it or beside a timed step)
            if "files" in c2 and args.cpu_baseline and os.path.exists(REF_BIN) and args.cpu_c2_pairs > 0:
                try:
                    c2_cpu = config_leg_cpu_only(c2["files"], c2name, min(args.cpu_c2_pairs, c2["pairs"]), limit_s=max(30.0, min(300.0, args.budget - spent() - 200)))
                except Exception as e:   # noqa: BLE001
                    c2_cpu = {"value": None, "unit": "pairs/s", "cores": host_cores(), "kind": "reference", "sample": "not measured: %s" % repr(e)[:200]}
            # C2 is the workload when the remaining steps + warm-up runs of it (1.5 % slack; runs of one box differ by less), and 20 s for what
            # must follow them (md5 sums of the outputs, the line), end inside the budget
            need = c2.get("seconds", 1e9) * (args.steps + args.warmup - 1) * 1.015 + 20
            use_c2 = "seconds" in c2 and args.warmup >= 1 and spent() + need <= args.budget
            c2["workload_decision"] = {"spent_s_before_the_steps": spent(), "needed_s": need, "budget_s": args.budget}

        def pmc_on_small_batch(limit_s):
            """roofline.traffic when C2 is the workload: the two PMC passes run on the C2 recipe at 100 k pairs (a C2 run under the
            counters would take a quarter of an hour), plus one plain run of that batch for its rounds and algorithmic bytes"""
            pf = make_batch(tmp, 100000, 2000, 1)
            tr, detail = pmc_traffic(pf[0], pf[1], pf[2], threads, local_rank, tmp, limit_s=limit_s)
            if not tr:
                return None, detail, None
            run_stage1(pf[0], pf[1], pf[2], os.path.join(tmp, "pmcb"), threads, local_rank, stats=os.path.join(tmp, "pmcb.json"))
            return tr, detail, json.load(open(os.path.join(tmp, "pmcb.json")))["add_query"]
        # barcode mode at 1 M pairs / 10 k cells (about 40 s with the synthesis of its files), BEFORE the steps when they leave room:
        # the driver's line then carries `stage1_cells_1m` -- seconds and `identical` -- whatever the steps leave of the budget (VERDICT r5 #6b)
        cells_early = None
        if use_c2 and args.side_legs and spent() + need + 60 <= args.budget:
            cells_early = stage1_cells_1m()
            c2["workload_decision"]["stage1_cells_1m_before_the_steps_s"] = spent() - c2["workload_decision"]["spent_s_before_the_steps"]
        pmc_early = None
        if use_c2 and args.traffic and spent() + need + 110 <= args.budget:   # with time to spare the passes come first too: the line has its traffic whatever the steps leave
            try:
                pmc_early = pmc_on_small_batch(100.0)
                c2["workload_decision"]["pmc_passes_before_the_steps_s"] = spent() - c2["workload_decision"]["spent_s_before_the_steps"]
            except Exception as e:   # noqa: BLE001
                pmc_early = (None, {"error": repr(e)[:300]}, None)
        pairs = c2["pairs"] if use_c2 else (args.pairs if args.pairs > 0 else args.fallback_pairs)
        clones = args.clones if args.clones > 0 else max(1, pairs // 50)
        if use_c2:
            fa, f1, f2, _, _ = c2["files"]
        else:
            fa, f1, f2 = make_batch(tmp, pairs, clones, 1)
        mine = os.path.join(tmp, "mine")
        stats_path = os.path.join(tmp, "stats.json")

        for _ in range(args.warmup - (1 if use_c2 else 0)):
            run_stage1(fa, f1, f2, mine, threads, local_rank)
        sync()
        t0 = time.perf_counter()
        steps_done = 0
        for _ in range(args.steps):
            # last resort, never expected (the decision above leaves slack): a box that turned so much slower during the steps that the
            # next one would run into the driver's limit -- a line over the steps that were timed, saying so, instead of no line at all
            if use_c2 and steps_done >= 1 and spent() + (time.perf_counter() - t0) / steps_done * 1.03 > args.budget + 15:
                break
            run_stage1(fa, f1, f2, mine, threads, local_rank, stats=stats_path)
            steps_done += 1
        sync()
        dt = time.perf_counter() - t0
        steps_asked, args.steps = args.steps, steps_done

        st = json.load(open(stats_path))
        ph = st["phases_s"]
        aq, ra = st["add_query"], st["rough_annotation"]
        alg_ann = annotate_bytes(ra["reads"], 150, ra["hits"])
        roof = add_roofline(aq)
        workload = (("config C2 itself: 1000000 synthetic 150 bp PE pairs (20000 clones, seed 1)" if c2name == "c2" else "NOT config C2 -- the stand-in %s of T4_BENCH_C2_STANDIN: %d pairs (%d clones, seed 1)" % (c2name, pairs, clones)) if use_c2 else
                    "C2 recipe, %d synthetic 150 bp PE pairs (%d clones, seed 1)%s" % (pairs, clones,
                    ": config C2 itself takes %.1f s per step here, %d steps + %d warm-up do not fit the budget of %.0f s -- C2 is the `c2` record of this line"
                    % (c2["seconds"], args.steps, args.warmup, args.budget) if c2 and "seconds" in c2 else ""))
        out = {
            "metric": "stage-1 assembly read pairs/sec (150 bp PE), whole stage 1",
            "value": pairs * args.steps / dt,
            "unit": "pairs/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": None, "vs_baseline": None,
            "dtype": "int32/u64", "data": "synthetic",
            "config": {"workload": workload + ", -f hg38_bcrtcr.fa, k=9, bulk mode; one step = whole stage 1 through trust4-hip -t %d --skipMateExtension, FASTQ files in -> "
                                              "_raw.out / _assembled_reads.fa / _final.out out, process start to exit" % threads,
                       "pairs_per_step": pairs, "is_baseline_config_c2": bool(use_c2 and c2name == "c2"), "host_threads": threads, "contigs": st["contigs"], "assembled_reads": st["assembled_reads"],
                       "sharding": "N = 1: bulk-mode stage 1 is one ordered chain; N > 1 measures the barcode-mode sample sharded by cells (see the docstring)",
                       "phases_s": {"parse_processread_21mers": ph["input_processed_counted"], "sort": ph["sorted"] - ph["input_processed_counted"],
                                    "rough_annotation": ph["rough_annotation"] - ph["sorted"], "trim": ph["trimmed_ready"] - ph["rough_annotation"],
                                    "addread_pass": ph["assembled"] - ph["trimmed_ready"], "outputs": ph["outputs_written"] - ph["assembled"]}},
            "roofline": roof, "chain": chain_block(st),
            "passes": {"add_query": aq,
                       "rough_annotation_in_step": {"reads": ra["reads"], "hits": ra["hits"], "kernel_ms": ra["kernel_ms"],
                                                    "achieved_GBs": alg_ann / (max(ra["kernel_ms"], 1e-6) * 1e-3) / 1e9}},
        }
        if steps_done != steps_asked:
            out["steps_cut_short"] = "%d of the %d steps asked for were timed: the next one would have ended beyond the %d s the whole run may take" % (steps_done, steps_asked, int(args.budget))
        if use_c2:
            out["parity_on_bench_batch"] = bool(c2.get("identical")) and all(file_md5(mine + x) == c2["md5"][x] for x in OUT_SUFFIXES)
        # the reference on this box
        if args.cpu_baseline and os.path.exists(REF_BIN):
            if use_c2:
                out["cpu_baseline"] = c2_cpu
            else:
                out["cpu_baseline"], out["parity_on_bench_batch"] = cpu_baseline(tmp, fa, f1, f2, pairs, mine, args.cpu_single_pairs)
                if c2 and c2_cpu:
                    c2["cpu_baseline"] = c2_cpu
        left = args.budget + 10 - spent()   # what follows is left out when it could run into the driver's limit
        if args.traffic and pmc_early is None and left < 150:
            out["roofline"]["traffic_detail"] = {"skipped": "%.0f s left of the run's budget: the two PMC passes are in the line of `python bench.py` with its default steps (profiles/)" % left}
        elif args.traffic:
            aqb = None
            try:
                if pmc_early is not None:
                    tr, detail, aqb = pmc_early
                elif use_c2:
                    tr, detail, aqb = pmc_on_small_batch(max(60.0, left - 50))
                else:
                    tr, detail = pmc_traffic(fa, f1, f2, threads, local_rank, tmp, limit_s=max(60.0, left - 40))
            except Exception as e:   # noqa: BLE001
                tr, detail = None, {"error": repr(e)[:300]}
            out["roofline"]["traffic_detail"] = detail
            if tr and not use_c2:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_over_algorithmic"] = tr / roof["algorithmic_bytes_per_step"]
            elif tr and aqb:
                # measured on the 100 k-pair batch of the same recipe, live in this run: a block of its OWN -- traffic, algorithmic bytes and their
                # ratio all of THAT batch (a number scaled to C2's rounds beside the small batch's ratio contradicted
                # itself; C2's rounds are three times heavier). `roofline.traffic` stays null: it is not this workload's traffic.
                algb = add_bytes(aqb["reads_queried"], 150, aqb["hits"])
                out["roofline"]["traffic_small_batch"] = {"workload": "C2 recipe at 100 k pairs (2 000 clones, seed 1), one whole run under each counter, this box, this run",
                                                          "traffic": tr, "algorithmic_bytes": algb, "traffic_over_algorithmic": tr / algb, "rounds": aqb["rounds"],
                                                          "traffic_per_round": tr / max(1, aqb["rounds"]), "algorithmic_bytes_per_round": algb / max(1, aqb["rounds"])}
        # config C2's own traffic: two PMC passes over a whole C2 run take seven minutes, more than a bench run can spare beside 25 timed
        # C2 steps -- measured in the builder's GPU session with this engine and committed (tools/pmc_c2_summary.py); cited, with its
        # source, not measured by this run
        if use_c2 and c2name == "c2":
            try:
                cand = sorted(x for x in os.listdir(os.path.join(ROOT, "profiles")) if x.startswith("r06") and x.endswith("_c2_pmc_summary.json"))
                prof = json.load(open(os.path.join(ROOT, "profiles", cand[-1])))
                out["roofline"]["traffic_c2_profile"] = {"source": "profiles/" + cand[-1] + " (committed measurement of a builder session, NOT of this run)", "traffic": prof["traffic_bytes"],
                                                         "algorithmic_bytes": prof["algorithmic_bytes"], "traffic_over_algorithmic": prof["traffic_over_algorithmic"],
                                                         "fetch_bytes_corrected": prof["fetch_bytes_corrected"], "write_bytes": prof["write_bytes"], "note": prof.get("note", "")}
            except Exception:   # noqa: BLE001
                pass
        if c2 is not None:
            c2.pop("files", None)
            out["c2"] = c2
        if cells_early is not None:
            out["stage1_cells_1m"] = cells_early
        if args.side_legs:
            for extra in [x for x in args.config_leg.split(",") if x]:
                out[extra.replace(":", "_")] = config_leg(extra, threads, local_rank)
                out[extra.replace(":", "_")].pop("files", None)
            left = args.budget + 10 - spent()   # side legs only while the whole run stays inside the driver's limit
            if left > 300:
                try:
                    out["passes"]["rough_annotation_c2"] = annotate_pass_c2(local_rank, 1000000, 20000, 3)
                except Exception as e:   # noqa: BLE001
                    out["passes"]["rough_annotation_c2"] = {"error": repr(e)[:300]}
                out["stage1_cells"] = stage1_cells(100000, 1000)
                out["stage0_e2e"] = stage0_e2e(400000)
                if cells_early is None and args.budget + 10 - spent() > 150:
                    out["stage1_cells_1m"] = stage1_cells_1m()
            else:
                out["side_legs_skipped"] = "%.0f s left of the run's budget" % left
        print(json.dumps(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config_leg_cpu_only(files, name, prefix_pairs, limit_s=None):
    """the reference on THIS box on the first prefix_pairs pairs of a config's own files"""
    fa, f1, f2 = files[0], files[1], files[2]
    tmp = os.path.dirname(f1)
    h1, h2 = os.path.join(tmp, "cpu_1.fq"), os.path.join(tmp, "cpu_2.fq")
    head_fastq(f1, h1, prefix_pairs)
    head_fastq(f2, h2, prefix_pairs)
    cores = host_cores()
    t0 = time.perf_counter()
    subprocess.run([REF_BIN, "-t", str(cores), "--skipMateExtension", "-f", fa, "-1", h1, "-2", h2, "-o", os.path.join(tmp, "cpuref")], check=True, stderr=subprocess.DEVNULL, timeout=limit_s)
    dc = time.perf_counter() - t0
    return {"value": prefix_pairs / dc, "unit": "pairs/s", "cores": cores, "kind": "reference", "seconds": dc,
            "sample": "oracle/_ref/trust4 -t %d --skipMateExtension on the first %d pairs of config %s's own files (%s, %s), on this box, %.1f s wall"
                      % (cores, prefix_pairs, name.upper(), os.path.basename(f1), os.path.basename(f2), dc)}


if __name__ == "__main__":
    main()

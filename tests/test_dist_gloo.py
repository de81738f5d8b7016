"""N > 1 paths on CPU: world_size 2 (and 5, 8) over gloo or through the engine's file transport. Covers bench.py's launcher plumbing
(trust4_amd/dist.py), the sharded barcode mode and the sharded rough-annotation pass of the driver (emulated engine; the GPU engine
itself is exercised by the -m gpu suite)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import trust4_amd.dist as t4dist
    dist = t4dist.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    dist.barrier()
    mx = t4dist.max_over_ranks(dist, float(rank + 1), "cpu")
    q.put((rank, mx))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_plumbing():
    """what bench.py --gpus N takes from trust4_amd/dist.py: the process group from the launcher's environment, barriers, and the
    timed interval max-reduced over the ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 2.0), (1, 2.0)]   # MAX over ranks, on both


def _stage1_worker(rank, world, port, argv, prefix, driver, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      T4_DEVICE_OVERRIDE="0", T4_DIST_BACKEND="gloo")
    import trust4_amd.stage1_dist as sd
    rc = sd.main(argv + ["-o", prefix], driver=driver)
    q.put((rank, rc))


def run_stage1_two_ranks(tmp_path, driver, pairs, cells, seed, world=2):
    """barcode-mode stage 1 on 2 ranks (trust4_amd/stage1_dist.py: cells sharded by rank, contig records all-gathered and
    renumbered on rank 0) must reproduce the single-process outputs byte for byte"""
    import filecmp
    import gzip
    import shutil
    import subprocess
    import t4libs
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", str(seed), pre, "--cells", str(cells)], check=True)
    argv = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
    single = str(tmp_path / "single")
    subprocess.run([driver] + argv + ["-o", single], check=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    merged = str(tmp_path / "merged")
    procs = [ctx.Process(target=_stage1_worker, args=(r, world, port, argv, merged, driver, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, 0) for r in range(world)]
    for suffix in ("_raw.out", "_final.out", "_assembled_reads.fa"):
        assert filecmp.cmp(single + suffix, merged + suffix, shallow=False), suffix
    assert open(single + "_raw.out").read().count(">") >= cells


def run_engine_merge(tmp_path, driver, pairs, cells, seed, world, env=None, extra=(), expect_log=None, common=()):
    """The merge trust4-hip itself runs on a multi-GPU node (trust4_main.cpp: shard headers all-gathered, every rank renumbers its own
    contig records, records gathered to rank 0, every rank writes its slice of _assembled_reads.fa at its offset), with the file
    transport (--gatherDir) in place of RCCL: `world` processes of the driver, no Python in the exchange. Outputs must equal the
    single-process run's byte for byte -- with world > 1 every rank beyond the first renumbers with base > 0."""
    import filecmp
    import gzip
    import shutil
    import subprocess
    import t4libs
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", str(seed), pre, "--cells", str(cells)], check=True)
    argv = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"] + list(common)   # (common: options of the single-process run too)
    e = dict(os.environ)
    e.update(env or {})
    single = str(tmp_path / "single")
    subprocess.run([driver] + argv + ["-o", single], check=True, env=e)
    gdir = tmp_path / "gather"
    gdir.mkdir()
    merged = str(tmp_path / "merged")
    procs = [subprocess.Popen([driver] + argv + ["-o", merged, "--cellShard", "%d/%d" % (r, world), "--gatherDir", str(gdir)] + list(extra), env=e, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=900)[1] for p in procs]
    assert [p.returncode for p in procs] == [0] * world, logs
    assert "Gathered %d shards over files" % world in logs[0]
    if expect_log:
        for l in logs:
            assert expect_log in l, l[-600:]
    for suffix in ("_raw.out", "_final.out", "_assembled_reads.fa"):
        assert filecmp.cmp(single + suffix, merged + suffix, shallow=False), suffix
    n = open(single + "_raw.out").read().count(">")
    assert n >= (0 if common else cells)   # (options like --contigMinCov drop contigs, all of them at times)
    return n


def test_input_dealt_out_by_cells(tmp_path):
    """--cellShard with a transport, round 5: a rank builds, ProcessReads and counts the pairs of ITS cells only (the cells follow from
    the barcode file alone), and the 21-mer counts of the whole sample are put together through one exchange of the ranks' tables
    (t4_kmer_count_export / _merge on the device, the host threads' maps under T4_GPU_KMERCOUNT=0). (a) device counts, 3 ranks;
    (b) host counts; (c) more ranks than cells (ranks without a pair); (d) T4_SHARD_INPUT=0: round 4's way (input phases over the
    whole sample on every rank); (e) --contigMinCov (pair counts per barcode come from the whole barcode file). All: the
    single-process files byte for byte -- the read statistics of `_assembled_reads.fa` (min / median count per read) are where a
    count that missed another rank's reads would show."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    env = {"HIPEMU_THREADS": "2", "T4_THREADS": "2"}
    for tag in "abcde":
        (tmp_path / tag).mkdir()
    dealt = "their pairs alone are processed and counted here"
    run_engine_merge(tmp_path / "a", exe, 120, 7, 21, 3, env=env, expect_log=dealt)
    run_engine_merge(tmp_path / "b", exe, 100, 6, 22, 3, env=dict(env, T4_GPU_KMERCOUNT="0"), expect_log="pairs of this rank's table went to the other ranks")
    run_engine_merge(tmp_path / "c", exe, 50, 3, 23, 5, env={"HIPEMU_THREADS": "1"}, expect_log=dealt)
    run_engine_merge(tmp_path / "d", exe, 100, 6, 21, 2, env=dict(env, T4_SHARD_INPUT="0"), expect_log="are this rank's (")
    run_engine_merge(tmp_path / "e", exe, 120, 7, 24, 2, env=env, common=["--contigMinCov", "8"], expect_log=dealt)
    # negative control: without the other ranks' counts the files differ (the comparison above does see the counts)
    (tmp_path / "f").mkdir()
    with pytest.raises(AssertionError, match="_assembled_reads.fa|_raw.out"):
        run_engine_merge(tmp_path / "f", exe, 120, 7, 21, 3, env=dict(env, T4_TEST_NO_COUNT_MERGE="1"))


def test_barcode_file_that_cannot_be_read_twice(tmp_path):
    """with the input dealt out by cells every rank reads the barcode file once AHEAD of the input loop -- a FIFO (process
    substitution, /dev/stdin) would be drained by that. Whether every barcode file is a regular file follows from argv, so every rank
    decides alike: the input is not dealt out, every rank reads the whole sample once (round 4's way), and says so. Two ranks, each
    with its own FIFO in place of the barcode file: the single-process files byte for byte."""
    import filecmp
    import gzip
    import shutil
    import subprocess
    import threading
    import t4libs
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "100", "0", "31", pre, "--cells", "6"], check=True)
    env = dict(os.environ, HIPEMU_THREADS="2", T4_THREADS="2")
    base = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--UMI", pre + "_umi.fa"]
    single = str(tmp_path / "single")
    subprocess.run([exe] + base + ["--barcode", pre + "_bc.fa", "-o", single], check=True, env=env, stderr=subprocess.DEVNULL)
    gdir = tmp_path / "gather"
    gdir.mkdir()
    merged = str(tmp_path / "merged")
    procs, feeders = [], []
    for r in range(2):
        fifo = str(tmp_path / ("bc%d.fifo" % r))
        os.mkfifo(fifo)

        def feed(path=fifo):
            with open(pre + "_bc.fa", "rb") as src, open(path, "wb") as dst:   # (blocks until the driver opens its end)
                shutil.copyfileobj(src, dst)
        t = threading.Thread(target=feed, daemon=True)
        t.start()
        feeders.append(t)
        procs.append(subprocess.Popen([exe] + base + ["--barcode", fifo, "-o", merged, "--cellShard", "%d/2" % r, "--gatherDir", str(gdir)], env=env, stderr=subprocess.PIPE, text=True))
    logs = [p.communicate(timeout=600)[1] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], logs
    for l in logs:
        assert "cannot be read twice" in l and "their pairs alone are processed and counted here" not in l, l[-600:]
    for suffix in ("_raw.out", "_final.out", "_assembled_reads.fa"):
        assert filecmp.cmp(single + suffix, merged + suffix, shallow=False), suffix


def test_early_shard_paths(tmp_path):
    """--cellShard with a transport (round 4): every rank lets go of the other ranks' reads once the sample's 21-mers are counted and
    runs statistics, sort, rough annotation, barcode-wise counts and the cell pass on its own cells only. (a) the default, logged per
    rank; (b) more ranks than cells: ranks without a read; (c) --lateShard: every rank keeps every read up to the cell pass (round 3's
    way); (d) the fallback: identical reads either side of a rank boundary (forced here) -> every rank starts a --lateShard run of its
    own and passes its status on. All: the single-process files byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    env = {"HIPEMU_THREADS": "2", "T4_THREADS": "2"}
    for tag in "abcd":
        (tmp_path / tag).mkdir()
    run_engine_merge(tmp_path / "a", exe, 120, 7, 9, 2, env=env, expect_log="of 7 are this rank's")
    run_engine_merge(tmp_path / "b", exe, 60, 3, 11, 5, env={"HIPEMU_THREADS": "1"}, expect_log="are this rank's")
    run_engine_merge(tmp_path / "c", exe, 120, 7, 9, 2, env=env, extra=["--lateShard"])
    run_engine_merge(tmp_path / "d", exe, 120, 7, 9, 2, env=dict(env, T4_TEST_FORCE_COUPLED="1"), expect_log="starting over with --lateShard")


def test_engine_merge_two_and_eight_ranks(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    (tmp_path / "w2").mkdir()
    (tmp_path / "w8").mkdir()
    run_engine_merge(tmp_path / "w2", exe, 120, 7, 9, 2, env={"T4_CELL_GROUPS": "2", "T4_THREADS": "2", "HIPEMU_THREADS": "2"})   # (each rank: its cells in two groups)
    run_engine_merge(tmp_path / "w8", exe, 100, 11, 10, 8, env={"HIPEMU_THREADS": "1"})


def run_read_shard(tmp_path, driver, pairs, clones, seed, world, env=None):
    """--readShard R/N (bulk mode): one sample's rough-annotation pass cut into `world` ranges of the sorted distinct reads,
    a process per range, the 160-byte records exchanged through --gatherDir; rank 0 runs the ordered pass alone and must write the
    single-process files byte for byte, the other ranks end with status 0 after the exchange and write nothing."""
    import filecmp
    import gzip
    import shutil
    import subprocess
    import t4libs
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "bulk")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), str(clones), str(seed), pre], check=True)
    argv = ["--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    e = dict(os.environ)
    e.update(env or {})
    single = str(tmp_path / "single")
    subprocess.run([driver] + argv + ["-o", single], check=True, env=e)
    gdir = tmp_path / "gather"
    gdir.mkdir()
    outs = [str(tmp_path / ("rank%d" % r)) for r in range(world)]
    procs = [subprocess.Popen([driver] + argv + ["-o", outs[r], "--readShard", "%d/%d" % (r, world), "--gatherDir", str(gdir)], env=e, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=900)[1] for p in procs]
    assert [p.returncode for p in procs] == [0] * world, logs
    for r in range(world):
        assert "Rough annotations of %d read ranges exchanged" % world in logs[r]
    for suffix in ("_raw.out", "_final.out", "_assembled_reads.fa"):
        assert filecmp.cmp(single + suffix, outs[0] + suffix, shallow=False), suffix
        for r in range(1, world):
            assert not os.path.exists(outs[r] + suffix)
    assert open(single + "_raw.out").read().count(">") > 0
    return logs


def test_read_shard_two_and_three_ranks(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    (tmp_path / "w2").mkdir()
    (tmp_path / "w3").mkdir()
    run_read_shard(tmp_path / "w2", exe, 300, 12, 21, 2)
    logs = run_read_shard(tmp_path / "w3", exe, 200, 8, 22, 3, env={"HIPEMU_THREADS": "2"})
    import re
    spans = sorted(tuple(int(x) for x in re.search(r"this rank: reads (\d+)-(\d+) of (\d+)", l).groups()) for l in logs)
    assert spans[0][0] == 0 and spans[-1][1] == spans[-1][2] and all(spans[i][1] == spans[i + 1][0] for i in range(2))   # the ranges tile the read list
    assert all(b > a for a, b, _ in spans)


def test_read_shard_refuses_without_transport_or_with_cell_shard(tmp_path):
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    r = subprocess.run([exe, "--skipMateExtension", "-f", "x.fa", "-1", "a", "-2", "b", "--readShard", "0/2"], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "--readShard needs" in r.stderr
    r = subprocess.run([exe, "-f", "x.fa", "-1", "a", "-2", "b", "--readShard", "2/2", "--gatherDir", str(tmp_path)], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "R/N" in r.stderr


@pytest.mark.gpu
def test_read_shard_one_rank_rccl_gpu(tmp_path):
    """`--readShard 0/1 --rcclId FILE`: the annotation records go through t4_comm_allgather_bytes (RCCL bound by dlopen, ncclCommInitRank,
    two ncclAllGather) -- with one rank, a box has one GPU -- and the outputs are the plain run's."""
    import filecmp
    import gzip
    import shutil
    import subprocess
    import t4libs
    import trust4_amd.build as b
    b.build()
    exe = os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip")
    fa = str(tmp_path / "ref.fa")
    with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "bulk")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "5000", "100", "24", pre], check=True, stdout=subprocess.DEVNULL)
    argv = ["--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    subprocess.run([exe] + argv + ["-o", str(tmp_path / "plain")], check=True, stderr=subprocess.DEVNULL)
    p = subprocess.run([exe] + argv + ["-o", str(tmp_path / "rccl"), "--readShard", "0/1", "--rcclId", str(tmp_path / "id")], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "read ranges exchanged over RCCL" in p.stderr, p.stderr[-800:]
    for suffix in ("_raw.out", "_final.out", "_assembled_reads.fa"):
        assert filecmp.cmp(str(tmp_path / "plain") + suffix, str(tmp_path / "rccl") + suffix, shallow=False), suffix


@pytest.mark.gpu
def test_read_shard_four_ranks_gpu(tmp_path):
    """the same with the real engine: four ranks share the box's one GPU, file transport"""
    import trust4_amd.build as b
    b.build()
    run_read_shard(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip"), 20000, 400, 23, 4)


@pytest.mark.gpu
def test_engine_merge_four_ranks_gpu(tmp_path):
    """the same with the real engine: four ranks share the box's one GPU, file transport"""
    import trust4_amd.build as b
    b.build()
    run_engine_merge(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip"), 4000, 60, 12, 4)


def test_bench_gpus2_plumbing_dry_run(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment), without
    GPUs: T4_BENCH_CPU_DRYRUN swaps RCCL for gloo and trust4-hip for the emulated driver; the command line keeps --rcclId, and since the
    emulated engine has no RCCL its t4_comm_init fails on every rank: the run takes the engine's automatic fall-back to the file
    transport (every rank publishes how its communicator came up, all switch together), which the line must report. Everything
    else is the code an 8-GPU node runs: the sample, the sharded steps, max over ranks, the per-rank phase report, the one-rank run and
    the md5 comparison of its files with the sharded run's."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    port = 29700 + (os.getpid() % 200)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   T4_BENCH_CPU_DRYRUN="1", T4_DRIVER=exe, HIPEMU_THREADS="2", TMPDIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--cells-pairs", "60", "--cells", "4",
                                       "--cells-threads", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-600:] for o in outs]
    line = json.loads([x for x in outs[0][0].strip().split("\n") if x.startswith("{")][-1])
    assert not [x for x in outs[1][0].split("\n") if x.startswith("{")]          # rank 0 alone prints the line
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["warmup"] == 1 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["pairs"] == 120 and line["config"]["cells"] == 8
    assert line["one_rank"]["identical"] is True
    assert line["config"]["transport"].startswith("files (fallback from RCCL: rank 0: fail"), line["config"]["transport"]
    per = line["config"]["per_rank_s"]
    assert len(per["replicated_phases"]) == 2 and min(per["replicated_phases"]) > 0
    assert len(per["own_cells_before_the_add_pass"]) == 2 and min(per["own_cells_before_the_add_pass"]) > 0   # (the early shard: a mark of its own in the stats)


def test_bench_gpus8_plumbing_dry_run(tmp_path):
    """`bench.py --gpus 8` as the driver will launch it on the first 8-GPU node, without GPUs: eight ranks over gloo, the
    emulated driver, FIVE cells of uneven sizes for eight ranks -- so some ranks own no cell at all and the others own one or two --, the
    engine's agreed fall-back from RCCL to the file transport on every rank, and the line a first SCALE record will be read from:
    `config.per_rank_s` with eight entries per phase, `one_rank.identical` and `one_rank.speedup_of_the_sharded_run`."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    exe = _emulated_driver()
    port = 29950 + (os.getpid() % 40)
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   T4_BENCH_CPU_DRYRUN="1", T4_DRIVER=exe, HIPEMU_THREADS="1", OMP_NUM_THREADS="1", TMPDIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--cells-pairs", "15", "--cells-total", "5",
                                       "--cells-threads", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    assert [p.returncode for p in procs] == [0] * 8, [o[1][-400:] for o in outs]
    line = json.loads([x for x in outs[0][0].strip().split("\n") if x.startswith("{")][-1])
    for r in range(1, 8):
        assert not [x for x in outs[r][0].split("\n") if x.startswith("{")]          # rank 0 alone prints the line
    assert line["n_gpus"] == 8 and line["steps"] == 1 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["pairs"] == 120 and line["config"]["cells"] == 5
    assert line["one_rank"]["identical"] is True and line["one_rank"]["speedup_of_the_sharded_run"] > 0
    assert line["config"]["transport"].startswith("files (fallback from RCCL: rank 0: fail"), line["config"]["transport"]
    per = line["config"]["per_rank_s"]
    for key in ("replicated_phases", "own_cells_before_the_add_pass", "add_pass_of_its_cells"):
        assert len(per[key]) == 8, key
    assert min(per["replicated_phases"]) > 0


def test_two_rank_barcode_stage1(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    run_stage1_two_ranks(tmp_path, _emulated_driver(), 120, 7, 9)


def test_eight_rank_barcode_stage1(tmp_path):
    """world size 8 (one node's worth of ranks; more ranks than some shards have cells): the merged outputs are still the
    single-process ones byte for byte"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    os.environ["HIPEMU_THREADS"] = "1"   # eight emulated ranks share this host's cores
    try:
        run_stage1_two_ranks(tmp_path, _emulated_driver(), 100, 11, 10, world=8)
    finally:
        os.environ.pop("HIPEMU_THREADS", None)


@pytest.mark.gpu
def test_two_rank_barcode_stage1_gpu(tmp_path):
    """the same on the real engine: two ranks share the box's one GPU (collective over gloo; on a multi-GPU node the
    launcher uses one GPU per rank and RCCL)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import trust4_amd.build as b
    b.build()
    run_stage1_two_ranks(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip"), 2000, 50, 11)


def _bench_single_dry_run(tmp_path, extra):
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_stage1_e2e import _emulated_driver
    env = dict(os.environ, T4_BENCH_CPU_DRYRUN="1", T4_DRIVER=_emulated_driver(), T4_BENCH_C2_STANDIN="c2micro", HIPEMU_THREADS="4", TMPDIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-c2-pairs", "200", "--cpu-single-pairs", "50"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-800:]
    return json.loads([x for x in p.stdout.strip().split("\n") if x.startswith("{")][-1])


def test_bench_single_gpu_workload_choice_dry_run(tmp_path):
    """`python bench.py --steps K --warmup W` at N = 1 without a GPU (T4_BENCH_CPU_DRYRUN: the emulated driver, a 400-pair stand-in for
    config C2): the run of "C2" comes first, the reference's timing on a prefix of its files before the steps, then the choice of the
    workload from what the WHOLE run may take -- inside the budget the steps are timed on the "C2" files (and the line says whether
    that was config C2: here it was not), outside it on the fallback batch with the "C2" run as the `c2` record."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "trust4")):
        pytest.skip("oracle/_ref/trust4 not built")
    line = _bench_single_dry_run(tmp_path, [])
    assert line["steps"] == 1 and line["warmup"] == 1 and line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] is None
    assert line["config"]["pairs_per_step"] == 400 and line["config"]["is_baseline_config_c2"] is False and "stand-in c2micro" in line["config"]["workload"]
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["value"] > 0 and "first 200 pairs" in line["cpu_baseline"]["sample"]
    assert line["c2"]["workload_decision"]["budget_s"] == 1770 and line["c2"]["workload_decision"]["spent_s_before_the_steps"] > 0
    assert "steps_cut_short" not in line and line["roofline"]["traffic"] is None
    # the whole run may take 1 s: "C2" cannot be the workload
    line = _bench_single_dry_run(tmp_path, ["--budget", "1", "--fallback-pairs", "300"])
    assert line["steps"] == 1 and line["config"]["pairs_per_step"] == 300 and line["config"]["is_baseline_config_c2"] is False
    assert "do not fit the budget" in line["config"]["workload"]
    assert line["parity_on_bench_batch"] is True and line["cpu_baseline"]["value"] > 0 and "whole batch" in line["cpu_baseline"]["sample"]
    assert line["c2"]["pairs"] == 400 and line["c2"]["cpu_baseline"]["value"] > 0

"""Barcode-mode stage 1 on N GPUs of one node: one process per GPU, launched as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m trust4_amd.stage1_dist \
        -f ref.fa -1 r_1.fq -2 r_2.fq --barcode bc.fa [--UMI umi.fa] -o PREFIX
Every rank runs `trust4-hip --cellShard RANK/N` on its own GPU: input, 21-mer counts and the rough annotation are replicated
(they define the global read order), the order-dependent Add pass -- the part that does not parallelise inside a cell --
runs on a contiguous range of cells per rank with no exchange. The one collective is at the end: the contig records of
every rank (text of its shard `_raw.out`, ids local to the shard) are all-gathered together with the contig-slot counts -- on GPUs
by the engine itself (`trust4-hip --rcclId`: t4_comm, ncclAllGather over xGMI from C++), in the CPU tests by torch.distributed
over gloo; rank 0 shifts every id by the slots of the earlier ranks --
the numbering the reference's cell-after-cell pass produces -- and writes PREFIX_raw.out / _final.out /
_assembled_reads.fa byte-identical to a single-process run."""
import os
import subprocess
import sys

import numpy as np

from . import dist as t4dist

ROOT = os.path.dirname(os.path.abspath(__file__))


def shift_ids(raw_text, base):
    """`>BARCODE_<id> name` header lines of a shard's Output (SeqSet.hpp:10951) with id += base"""
    if base == 0:
        return raw_text
    out = []
    for line in raw_text.split(b"\n"):
        if line.startswith(b">"):
            head, sep, rest = line.partition(b" ")
            bc, _, idx = head.rpartition(b"_")
            line = bc + b"_" + str(int(idx) + base).encode() + sep + rest
        out.append(line)
    return b"\n".join(out)


def all_gather_bytes(dist, payload, device):
    """variable-size all-gather of one bytes object per rank -> list of bytes (rank order)"""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if payload:
        buf[: len(payload)] = torch.from_numpy(np.frombuffer(payload, dtype=np.uint8).copy()).to(device)
    parts = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(parts, buf)
    return [bytes(p[:s].cpu().numpy().tobytes()) for p, s in zip(parts, sizes)]


def merge(shards, prefix):
    """shards: list (rank order) of dicts raw / main / rescue (bytes) and slots (int)"""
    base = 0
    raws = []
    for sh in shards:
        raws.append(shift_ids(sh["raw"], base))
        base += sh["slots"]
    raw = b"".join(raws)
    for suffix in ("_raw.out", "_final.out"):   # with barcodes _final.out is a second dump of the raw set (main.cpp:2018-2036)
        with open(prefix + suffix, "wb") as f:
            f.write(raw)
    with open(prefix + "_assembled_reads.fa", "wb") as f:
        for sh in shards:
            f.write(sh["main"])
        for sh in shards:
            f.write(sh["rescue"])
    return base


def main(argv=None, driver=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    rank, local_rank, world = t4dist.env_rank()
    prefix = "trust"
    if "-o" in argv:
        i = argv.index("-o")
        prefix = argv[i + 1]
        del argv[i: i + 2]
    driver = driver or os.environ.get("T4_DRIVER") or os.path.join(ROOT, "bin", "trust4-hip")
    import torch
    on_gpu = torch.cuda.is_available() and os.environ.get("T4_DIST_BACKEND", "nccl") == "nccl"   # gloo: CPU tests, or several ranks on one GPU
    dist = t4dist.init("nccl" if on_gpu else "gloo")
    device = "cuda:%d" % local_rank if on_gpu else "cpu"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if on_gpu and world > 1 and not os.environ.get("T4_DIST_PY"):
        # the exchange happens inside the engine: every rank's trust4-hip joins one RCCL communicator (bootstrapped through a file
        # that rank 0 creates) and all-gathers the shard results itself; rank 0 of the engine writes PREFIX_*.  Nothing goes
        # through Python or torch (T4_DIST_PY=1 keeps the torch.distributed path below, which is also what the gloo tests run).
        # (a directory of this run's own for the id, the ranks' status files and -- should RCCL not come up on every rank -- the files of
        # the engine's fall-back transport: a stale id or file of an earlier run can never be read)
        import shutil
        xdir = "%s.xfer.%s" % (prefix, os.environ.get("MASTER_PORT", "0"))
        if rank == 0:
            shutil.rmtree(xdir, ignore_errors=True)
            os.makedirs(xdir)
        dist.barrier()
        env = dict(os.environ)
        env["T4_DEVICE"] = str(local_rank)
        subprocess.run([driver] + argv + ["-o", prefix, "--cellShard", "%d/%d" % (rank, world), "--rcclId", os.path.join(xdir, "rcclid")], check=True, env=env)
        dist.barrier()
        if rank == 0:
            shutil.rmtree(xdir, ignore_errors=True)
        dist.destroy_process_group()
        return 0
    shard_prefix = "%s.shard%d" % (prefix, rank)
    env = dict(os.environ)
    env["T4_DEVICE"] = os.environ.get("T4_DEVICE_OVERRIDE", str(local_rank))   # the override is for the one-device CPU tests
    cmd = [driver] + argv + ["-o", shard_prefix]
    if world > 1:
        cmd += ["--cellShard", "%d/%d" % (rank, world)]
    env["T4_STATS_JSON"] = shard_prefix + "_stats.json"
    import time
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True, env=env)
    t_run = time.perf_counter() - t0
    try:   # where this rank's time went (phases of trust4-hip; everything before "trimmed_ready" is replicated on every rank)
        import json
        ph = json.load(open(shard_prefix + "_stats.json"))["phases_s"]
        sys.stderr.write("stage1_dist rank %d/%d: %.2f s; replicated phases (parse, counts, sort, rough annotation, trim) %.2f s, Add pass of this rank's cells %.2f s, outputs %.2f s\n"
                         % (rank, world, t_run, ph["trimmed_ready"], ph["assembled"] - ph["trimmed_ready"], ph["outputs_written"] - ph["assembled"]))
        os.remove(shard_prefix + "_stats.json")
    except Exception:   # noqa: BLE001  (reporting only)
        pass
    if world == 1:
        for suffix in ("_raw.out", "_final.out", "_assembled_reads.fa"):
            os.replace(shard_prefix + suffix, prefix + suffix)
        return 0

    def rd(suffix):
        with open(shard_prefix + suffix, "rb") as f:
            return f.read()
    meta = dict(line.split()[:2] for line in rd("_shard.meta").decode().strip().split("\n"))
    mine = {"raw": rd("_raw.out"), "main": rd("_assembled_reads.fa"), "rescue": rd("_assembled_reads_rescue.fa"), "slots": int(meta["contig_slots"])}
    gathered = {k: all_gather_bytes(dist, mine[k], device) for k in ("raw", "main", "rescue")}
    slots = all_gather_bytes(dist, str(mine["slots"]).encode(), device)
    if rank == 0:
        shards = [{"raw": gathered["raw"][r], "main": gathered["main"][r], "rescue": gathered["rescue"][r], "slots": int(slots[r])} for r in range(world)]
        total = merge(shards, prefix)
        sys.stderr.write("stage1_dist: %d ranks, %d contig slots\n" % (world, total))
    dist.barrier()
    for suffix in ("_raw.out", "_assembled_reads.fa", "_assembled_reads_rescue.fa", "_shard.meta"):
        try:
            os.remove(shard_prefix + suffix)
        except OSError:
            pass
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

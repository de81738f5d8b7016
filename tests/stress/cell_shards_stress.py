#!/usr/bin/env python3
"""Stress runs of barcode mode over ranks on the EMULATED engine (test infrastructure, run by hand; not collected by pytest): random
numbers of ranks, cells and pairs, host / device 21-mer counts, small first count tables, cell groups, --contigMinCov -- the merged
files of `trust4-hip --cellShard R/N --gatherDir` (input dealt out by cells, count tables put together through one exchange)
against one process, byte for byte (tests/test_dist_gloo.py::run_engine_merge).

    python tests/stress/cell_shards_stress.py SEED RUNS

Round 5: 40 + 60 runs, all identical."""
import os
import pathlib
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_dist_gloo import run_engine_merge   # noqa: E402
from test_stage1_e2e import _emulated_driver   # noqa: E402


def main():
    exe = _emulated_driver()
    rnd = random.Random(int(sys.argv[1]))
    bad = 0
    for it in range(int(sys.argv[2])):
        world, cells, pairs, seed = rnd.choice([2, 3, 4, 6]), rnd.choice([1, 2, 3, 5, 8, 13]), rnd.choice([20, 60, 120, 200]), rnd.randrange(1, 10000)
        env = {"HIPEMU_THREADS": "1", "T4_THREADS": str(rnd.choice([1, 2, 3]))}
        if rnd.random() < 0.3:
            env["T4_GPU_KMERCOUNT"] = "0"
        if rnd.random() < 0.2:
            env["T4_KC_SLOTS"] = "1024"
        if rnd.random() < 0.2:
            env["T4_CELL_GROUPS"] = str(rnd.choice([1, 2, 3]))
        common = ["--contigMinCov", str(rnd.choice([2, 5, 9]))] if rnd.random() < 0.25 else []
        with tempfile.TemporaryDirectory() as d:
            try:
                n = run_engine_merge(pathlib.Path(d), exe, pairs, cells, seed, world, env=env, common=common, expect_log="their pairs alone are processed")
                print("ok", it, world, cells, pairs, seed, env, common, "contigs", n, flush=True)
            except BaseException as e:   # noqa: BLE001
                bad += 1
                print("FAILED", it, world, cells, pairs, seed, env, common, repr(e)[:300], flush=True)
    print("runs that differed:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

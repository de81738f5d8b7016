"""Generates tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/libt4ref.so, built from
/root/reference by oracle/Makefile). Run in the build container only:  python tests/golden/make_golden.py
The vectors pin: sorted hit lists, GetOverlapsFromRead results and AnnotateRead(level 0) gene overlaps
for seeded synthetic reads (tools/t4synth.c) plus hand-made edge cases; and gap-DP known answers."""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from t4libs import REF_FA, Ref, Synth, rows_to_strs  # noqa: E402


def edge_reads(rnd, base):
    out = []

    def mut(x):
        x = list(x)
        for _ in range(rnd.randint(1, 12)):
            x[rnd.randrange(len(x))] = "N"
        return "".join(x)

    out += [mut(x) for x in base[:40]]
    out += ["".join(rnd.choice("ACGT") for _ in range(150)) for _ in range(20)]
    out += [x[: rnd.randint(0, 149)] for x in base[:20]]
    out += ["A" * 150, "ACGT" * 30, "N" * 40, "ACGTACGTA", "ACGTACGT", "TG" * 60 + "N" * 9 + "CA" * 30, ""]
    out += [x[:70] + "N" * rnd.randint(5, 12) + x[80:] for x in base[:30]]
    out += [base[i] + base[i + 1][20:150] for i in range(0, 40, 2)]
    return out


def main():
    r = Ref(9, REF_FA, 17)
    rnd = random.Random(2024)
    base = rows_to_strs(Synth(500, 11).next_reads(400))
    reads = base + edge_reads(rnd, base)
    ann = np.zeros((len(reads), 4, 9), dtype=np.float64)
    ovs, ovoff = [], [0]
    hits, hoff = [], [0]
    for i, rd in enumerate(reads):
        _, g = r.annotate_read0(rd)
        for t in range(4):
            ann[i, t] = g[t] if g[t][0] != -1 else (-1, 0, 0, 0, 0, 0, 0, 0, 0)
        ret, lst = r.overlaps_from_read(rd)
        ovs += [list(x) for x in lst]
        ovoff.append(len(ovs))
        if i % 8 == 0:
            h = r.hits(rd)
            h = h[np.lexsort((h[:, 1], h[:, 2], h[:, 0], h[:, 3]))]
            hits.append(h)
            hoff.append(hoff[-1] + len(h))
        else:
            hoff.append(hoff[-1])
    np.savez_compressed(os.path.join(HERE, "ref_query_k9.npz"), reads=np.array(reads), annotate=ann,
                        overlaps=np.array(ovs, dtype=np.float64), overlap_off=np.array(ovoff),
                        hits=np.concatenate(hits).astype(np.int32), hit_off=np.array(hoff))
    # DP known answers
    dps = []
    for it in range(400):
        lt, lp = rnd.randint(0, 30), rnd.randint(0, 30)
        t = "".join(rnd.choice("ACGT") for _ in range(lt))
        p = list(t)
        for _ in range(rnd.randint(0, 4)):
            if p and rnd.random() < 0.5:
                del p[rnd.randrange(len(p))]
            else:
                p.insert(rnd.randint(0, len(p)), rnd.choice("ACGT"))
        p = "".join(p) if rnd.random() < 0.8 else "".join(rnd.choice("ACGT") for _ in range(lp))
        sc, al = r.global_alignment(t, p)
        dps.append((t, p, sc, "".join(map(str, al))))
    np.savez_compressed(os.path.join(HERE, "ref_dp_affine.npz"), t=np.array([d[0] for d in dps]), p=np.array([d[1] for d in dps]),
                        score=np.array([d[2] for d in dps]), align=np.array([d[3] for d in dps]))
    print("golden written:", len(reads), "reads,", len(ovs), "overlaps,", hoff[-1], "hits")


if __name__ == "__main__":
    main()

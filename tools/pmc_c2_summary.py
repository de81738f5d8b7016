"""HBM-side traffic of the AddRead query kernels over ONE WHOLE RUN of config C2, from the two rocprofv3 PMC passes of
profiles/scripts/gpu_r6*.sh (per-kernel sums of FETCH_SIZE and of WRITE_SIZE, counter units KB) and the stats JSON of a plain run of
the same binary: what bench.py cites as `roofline.traffic_c2_profile`.
usage: pmc_c2_summary.py <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> <stats.json> <out.json> [note]"""
import json
import sys

ADD_KERNELS = ("queryKernel<8192, 512, 512, 1>", "wideSeedKernel", "wideScatterKernel", "wideSortKernel", "wideStatsKernel", "wideChainKernel", "wideMergeKernel", "extendKernel")


def table(path):
    out = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        parts = line.rsplit(None, 2)
        out[parts[0].strip()] = (int(parts[1]), float(parts[2]))
    return out


def add_bytes(reads, length, hits):   # == bench.py add_bytes
    return reads * ((length + 3) // 4 + (length + 7) // 8 + 128 + 16 * length) + 8 * hits


fetch, write = table(sys.argv[1]), table(sys.argv[2])
st = json.load(open(sys.argv[3]))
aq = st["add_query"]
per = {}
f_kb = w_kb = 0.0
for name in sorted(set(fetch) | set(write)):
    if not any(k in name for k in ADD_KERNELS):
        continue
    f, w = fetch.get(name, (0, 0.0)), write.get(name, (0, 0.0))
    per[name] = {"launches": f[0] or w[0], "fetch_KB": f[1], "write_KB": w[1]}
    f_kb += f[1]
    w_kb += w[1]
alg = add_bytes(aq["reads_queried"], 150, aq["hits"])
traffic = (2.0 * f_kb + w_kb) * 1024.0
out = {"workload": "config C2 itself: 1 M synthetic 150 bp PE pairs (20 k clones, seed 1), one whole run of trust4-hip -t 8 --skipMateExtension under rocprofv3 --pmc <C> --kernel-trace (FETCH_SIZE and WRITE_SIZE in separate passes)",
       "kernels": "the AddRead query launches: " + ", ".join(ADD_KERNELS),
       "per_kernel": per, "fetch_bytes_corrected": 2.0 * f_kb * 1024.0, "write_bytes": w_kb * 1024.0, "traffic_bytes": traffic,
       "algorithmic_bytes": alg, "traffic_over_algorithmic": traffic / alg,
       "rounds": aq["rounds"], "reads_queried": aq["reads_queried"], "hits": aq["hits"], "kernel_ms_plain_run": aq["kernel_ms"],
       "method": "MI355X_MICROARCH.md HBM section: FETCH_SIZE x 2 on gfx950 (taken as the upper figure: the guide calibrates the factor on wide coalesced streams, these kernels gather 8-byte postings), WRITE_SIZE as counted, units KB; "
                 "L2 <-> fabric traffic (Infinity-Cache hits included): an upper bound of DRAM traffic",
       "note": sys.argv[5] if len(sys.argv) > 5 else ""}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fetch_bytes_corrected", "write_bytes", "traffic_bytes", "algorithmic_bytes", "traffic_over_algorithmic")}))

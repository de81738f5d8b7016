"""Summarise the rocprofv3 PMC passes of bench.py (FETCH_SIZE / WRITE_SIZE, separate runs) into profiles/<tag>_pmc_summary.json.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv, json, re, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(float)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
            acc[name] += float(row["Counter_Value"])
    return dict(acc)


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
qf = sum(v for k, v in fetch.items() if "queryKernel" in k) * 1024.0
qw = sum(v for k, v in write.items() if "queryKernel" in k) * 1024.0
out = {
    "command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-pairs 0 (C = FETCH_SIZE, WRITE_SIZE in separate passes)",
    "workload": "C2: 1M pairs / 2M reads per pass",
    "unit": "KB (counter units), summed over the tier launches of one pass",
    "FETCH_SIZE": fetch, "WRITE_SIZE": write,
    "query_kernels_fetch_bytes_raw": qf, "query_kernels_write_bytes_raw": qw,
    "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced streams by exactly 2x; the access pattern here is 8-byte posting gathers + 4/8-byte table probes, for which the guide gives no calibration, so `traffic_bytes` applies the 2x correction to FETCH_SIZE as the conservative upper figure and WRITE_SIZE is taken as is. Infinity-Cache hits are counted by these counters, so this is L2<->fabric traffic, an upper bound of HBM traffic (index + postings = 4 MB are cache resident).",
    "traffic_bytes": 2.0 * qf + qw,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("query_kernels_fetch_bytes_raw", "query_kernels_write_bytes_raw", "traffic_bytes")}))

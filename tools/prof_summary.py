#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small text/JSON files kept under profiles/.

  python tools/prof_summary.py stats  <dir-with-*kernel_stats.csv>                 -> kernel table on stdout
  python tools/prof_summary.py pmc    <dir-with-*counter_collection.csv> [...]     -> per-kernel mean counter per dispatch
  python tools/prof_summary.py traffic <fetch-dir> <write-dir> <out.json>          -> HBM bytes per launch per kernel

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) gfx950's
FETCH_SIZE tallies 128-byte requests as 64 bytes, so reads are doubled; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    # avg_wo_max drops the single longest launch: the first launch of a kernel in a process pays one-off costs (code object load,
    # first touch of the device-mapped host blocks: 30 ms for k_inter_ctu) that sit in bench.py's untimed warm-up step
    print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'avg_wo_max':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        n, tot, mx = int(r['Calls']), int(r['TotalDurationNs']), int(r['MaxNs'])
        wo = (tot - mx) / max(1, n - 1) / 1e3
        print(f"{short(r['Name'])[:60]:60s} {n:7d} {tot / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {wo:10.2f} {mx / 1e3:10.1f} {float(r['Percentage']):6.2f}")


def pmc_means(d):
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = out[short(r["Kernel_Name"])][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    return out


def pmc(dirs):
    for d in dirs:
        for k, cs in pmc_means(d).items():
            if "mihevc" not in k:
                continue
            n = max(v[1] for v in cs.values())
            print(k, f"dispatches={n}", " ".join(f"{c}={v[0] / max(1, v[1]):.4g}" for c, v in cs.items()))


def traffic(fd, wd, outp):
    rd, wr = pmc_means(fd), pmc_means(wd)
    res = {}
    for k in rd:
        if "mihevc" not in k:
            continue
        f = rd[k].get("FETCH_SIZE")
        w = wr.get(k, {}).get("WRITE_SIZE")
        if not f or not w:
            continue
        fetch = f[0] / f[1] * 1024 * 2      # KiB -> bytes, gfx950 half-count correction
        write = w[0] / w[1] * 1024
        res[k.split("::")[-1].split("<")[0]] = {"dispatches": f[1], "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                                                 "hbm_bytes_per_launch": round(fetch + write)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE x2 per the gfx950 correction",
               "kernels": res}, open(outp, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "stats":
        stats(sys.argv[2])
    elif cmd == "pmc":
        pmc(sys.argv[2:])
    elif cmd == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])

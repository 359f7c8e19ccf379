#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small text/JSON files kept under profiles/.

  python tools/prof_summary.py stats  <dir-with-*kernel_stats.csv>                 -> kernel table on stdout
  python tools/prof_summary.py pmc    <dir-with-*counter_collection.csv> [...]     -> per-kernel mean counter per dispatch
  python tools/prof_summary.py traffic <fetch-dir> <write-dir> <out.json>          -> HBM bytes per launch per kernel
  python tools/prof_summary.py pmcjson <out.json> <dir> [...]                      -> per-kernel VALU issue share (bench.py reads it)

Normalisation of the SQ counters (stated because the raw numbers cannot be read without it): every value printed by `pmc` is the MEAN
PER DISPATCH of what rocprofv3 wrote for that kernel.  On this pool rocprofv3 reports the SQ block of a subset of the shader engines
(SQ_WAVES comes out near 1/20 of the waves a grid launches), so absolute values are a sample; ratios of two SQ counters of the same
pass are not affected, and only such ratios are used: valu_active_per_wave = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both quad-cycles
summed over the same sampled waves) = share of a wave's resident time in which it issues vector ALU work; valu_issue_frac = that
times the waves resident per SIMD (workgroups per CU from the kernel's __launch_bounds__ x 4 waves / 4 SIMDs), capped at 1 = share
of a SIMD's issue slots the kernel fills.  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier).

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


def waves_per_simd_from_source():
    """workgroups per CU (= waves per SIMD: 256 threads are 4 waves over the CU's 4 SIMDs) each kernel is BUILT for: the second argument of its
    __launch_bounds__ in csrc/device.hip, read from the source so the table cannot go stale (round 2 kept it by hand and k_sao_decide was wrong).
    Registers are allocated to fit that many and the LDS footprints are sized for it; kernels without the argument: 8 (the VGPR limit
    of 64 at wave64 x 8 waves)."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hevc_amd", "csrc", "device.hip")).read()
    src = re.sub(r"#define INTRA_OCC (\d+)", "", src)
    occ = {}
    for m in re.finditer(r"__launch_bounds__\((.*)\)\s*void\s+(k_\w+)", src):      # one kernel header per line
        args = m.group(1).split(",", 1)
        n = 8
        if len(args) > 1:
            d = re.findall(r"\d+", args[1].replace("INTRA_OCC", "3"))
            if "sizeof" in args[1]:
                d = d[1:]              # (sizeof(T) == 1 ? A : B): the 8-bit instantiation's A
            n = int(d[0]) if d else 8
        occ[m.group(2)] = n
    return occ


def pmcjson(outp, dirs):
    acc = {}
    for d in dirs:
        for k, cs in pmc_means(d).items():
            if "mihevc" not in k or "unsigned short" in k:
                continue
            name = k.split("::")[-1].split("<")[0]
            acc.setdefault(name, {}).update({c: v[0] / max(1, v[1]) for c, v in cs.items()})
    res = {}
    occ_table = waves_per_simd_from_source()
    for name, c in acc.items():
        if not c.get("SQ_WAVE_CYCLES") or "SQ_ACTIVE_INST_VALU" not in c:
            continue
        per_wave = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
        occ = occ_table.get(name, 8)
        res[name] = {"valu_active_per_wave": round(per_wave, 4), "waves_per_simd": occ, "valu_issue_frac": round(min(1.0, per_wave * occ), 4),
                     "wait_frac": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in c else None,
                     "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1) if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c else None}
    json.dump({"source": "rocprofv3 --pmc SQ_* passes of the bench command `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras` (tools/pmc_kernels.sh); ratios of counters of one pass, waves per SIMD from the kernels' __launch_bounds__ in csrc/device.hip, see tools/prof_summary.py",
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
    elif cmd == "pmcjson":
        pmcjson(sys.argv[2], sys.argv[3:])

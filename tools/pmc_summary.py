"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out dir> [tag]

Writes r01_pmc_fetch_by_kernel.csv, r01_pmc_write_by_kernel.csv and pmc_traffic.json into the
out dir.  HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) 1024: on gfx950 FETCH_SIZE counts
half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is
exact; narrow / gathered reads are uncalibrated, so the figures are upper estimates.
"""
import collections
import csv
import glob
import json
import os
import sys


def by_kernel(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        nm = r["Kernel_Name"].split("(")[0].replace("kktdev::", "").replace("stg::", "").replace("void ", "")
        nm = nm.split("<")[0]
        tot[nm] += float(r["Counter_Value"])
        cnt[nm].add(r["Dispatch_Id"])
    return {k: (tot[k], len(cnt[k])) for k in tot}


def main():
    fd, wd, out = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r01"
    fetch, write = by_kernel(fd, "FETCH_SIZE"), by_kernel(wd, "WRITE_SIZE")
    for name, data, col in ((tag + "_pmc_fetch_by_kernel.csv", fetch, "FETCH_SIZE_KB"), (tag + "_pmc_write_by_kernel.csv", write, "WRITE_SIZE_KB")):
        with open(os.path.join(out, name), "w") as g:
            g.write(f"kernel,launches,{col}_total,{col}_per_launch\n")
            for k, (t, n) in sorted(data.items(), key=lambda kv: -kv[1][0]):
                g.write(f"{k},{n},{t:.1f},{t / max(n, 1):.3f}\n")
    res = {}
    for k in fetch:
        ft, fn = fetch[k]
        wt, wn = write.get(k, (0.0, fn))
        res[k] = {"launches": fn, "FETCH_SIZE_KB_per_launch": ft / max(fn, 1),
                  "WRITE_SIZE_KB_per_launch": wt / max(wn, 1),
                  "hbm_bytes_per_launch": (2 * ft / max(fn, 1) + wt / max(wn, 1)) * 1024}
    # which kernel sources these figures belong to (bench.py quotes them only for the same sources)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["_kernel_source_sha16"] = bench.kernel_source_sha16()
    json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

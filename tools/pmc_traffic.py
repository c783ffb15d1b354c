"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  The act_fwd kernel (reads n*2 bytes, writes n*2 bytes, 16 B per lane) is the
in-run calibration: its corrected fetch must equal its write.
"""
import collections
import csv
import json
import re
import sys


def short(name: str) -> str:
    # gathered-GEMM instantiations keep their template arguments (they are different kernels: bench.py names them the same way)
    m = re.match(r"void (conv_gemm_kernel<[^>]*>)", name)
    if m:
        return m.group(1).replace(" ", "")
    m = re.match(r"_Z\d+([a-z0-9_]+_kernel)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:60]


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for k in fetch:
        if k not in write or fetch[k][0] != write[k][0]:
            continue
        n = fetch[k][0]
        rd = 2.0 * 1024.0 * fetch[k][1] / n
        wr = 1024.0 * write[k][1] / n
        res[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    cal = res.get("act_fwd_kernel")
    meta = {"correction": "FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE KiB x1",
            "calibration_act_fwd_read_over_write": (cal["read_bytes_per_launch"] / cal["write_bytes_per_launch"]) if cal else None}
    json.dump({"meta": meta, "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))},
              open(out, "w"), indent=1)
    print(json.dumps(meta))
    for k, v in list(res.items())[:0]:
        print(k, v)


if __name__ == "__main__":
    main()

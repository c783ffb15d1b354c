"""Summarise one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE) into per-kernel MFMA utilisation.

usage: python tools/pmc_mfma.py <counter_collection.csv> <out.json>

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per
v_mfma_f32_32x32x16_bf16: /opt/skills/guides/MI355X_MICROARCH.md, cycle constants and DVFS sections).  A kernel that kept
every matrix pipe busy all the time would read 1.0 whatever clock the chip held; `eff_clock_ghz` = kernel cycles / kernel
time is only available where the trace carries timestamps.
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
    m = re.match(r"void ([a-z0-9_]+_kernel)", name)
    if m:
        return m.group(1)
    m = re.match(r"_Z\d+([a-z0-9_]+_kernel)", name)
    if m:
        return m.group(1)
    m = re.match(r"([a-z0-9_]+_kernel)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:60]


def main():
    path, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    res = {}
    for k, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0:
            continue
        cycles = gui / 8.0
        res[k] = {"launches": len(disp[k]), "kernel_cycles_total": cycles, "mfma_busy_cycles_total": mfma,
                  "mfma_busy_frac": mfma / (cycles * 256 * 4)}
        if "SQ_BUSY_CU_CYCLES" in c:
            res[k]["sq_busy_cu_cycles_total"] = c["SQ_BUSY_CU_CYCLES"]
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_total"]))
    json.dump({"meta": {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)", "source": path}, "kernels": res},
              open(out, "w"), indent=1)
    for k, v in list(res.items())[:8]:
        print(f"{k:40s} launches {v['launches']:5d}  mfma_busy_frac {v['mfma_busy_frac']:.3f}")


if __name__ == "__main__":
    main()

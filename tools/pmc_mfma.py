"""Per-kernel SQ counters of the bench workload from one rocprofv3 PMC pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE [+ GRBM_GUI_ACTIVE]).
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts
cycles.  Derived per kernel (launch-averaged): mfma_busy_of_wave_cycles = MFMA_BUSY / (4 x WAVE_CYCLES) (share of the time
waves are resident during which the matrix pipe works for them), lds_conflict_frac = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE.
    python tools/pmc_mfma.py counter_collection.csv > profiles/rNN_pmc_mfma.json"""
import collections, csv, json, sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    d = {"launches": n}
    d.update({c: round(v, 1) for c, v in m.items()})
    if m.get("SQ_WAVE_CYCLES"):
        d["mfma_busy_of_wave_cycles"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_WAVE_CYCLES"]), 4)
        d["wait_any_frac"] = round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
        d["wait_inst_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
        d["active_inst_frac"] = round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if m.get("GRBM_GUI_ACTIVE"):
        # MFMA utilisation against the PIPE's peak: busy cycles summed over the chip's 1024 SIMDs / (1024 x the kernel's
        # duration in shader cycles).  rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked against the kernel
        # trace: 4.93 M counts for a 270.8-us launch = 8 x 2.27 GHz), so duration x clock = GRBM_GUI_ACTIVE / 8 and the
        # denominator is 128 x GRBM_GUI_ACTIVE.  (Cross-check: the grouped weight gradient's 13.8 M MFMAs x 16 cycles =
        # 221 M busy cycles, exactly the counter; 0.35 here = the 0.34 - 0.35 of 2.5 PF that bench.py times with HIP events.)
        d["mfma_util_of_pipe_peak"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * m["GRBM_GUI_ACTIVE"]), 4)
    out[k] = d
order = sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0) * out[k]["launches"])
json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ counters> -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline "
                     "--no-profile --no-eval-leg (train mode; all launches of the run, tuning launches included)",
           "kernels": {k: out[k] for k in order}}, sys.stdout, indent=1)

"""Turn one rocprofv3 SQ-counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU GRBM_GUI_ACTIVE, with --kernel-trace only) into profiles/<round>_sq_counters.json: per kernel tag, the MFMA-pipe busy fraction
and where the wave cycles went.

    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)      (busy cycles are summed over the chip's SIMDs;
                                                                                          GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    valu_active / wait_any / wait_inst = SQ_ACTIVE_INST_VALU, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (all in quad-cycles)
usage: python tools/sq_counters.py <pass_dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import tag_of

if __name__ == "__main__":
    d, out = sys.argv[1:3]
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            t = tag_of(row["Kernel_Name"])
            per[t][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[t].add(row["Dispatch_Id"])
    res = {}
    for t, c in per.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if gui <= 0:
            continue
        res[t] = {"dispatches": len(disp[t]),
                  "mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0), 4),
                  "valu_active_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4) if wc else None,
                  "wait_any_of_wave_cycles": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4) if wc else None,
                  "wait_inst_of_wave_cycles": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4) if wc else None,
                  "valu_insts_per_dispatch": round(c.get("SQ_INSTS_VALU", 0.0) / max(len(disp[t]), 1)),
                  "gpu_cycles_per_dispatch": round(gui / 8.0 / max(len(disp[t]), 1))}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["gpu_cycles_per_dispatch"] * kv[1]["dispatches"]))
    json.dump({"_method": __doc__.split("usage")[0].strip(), "kernels": res}, open(out, "w"), indent=1)
    for t, k in list(res.items())[:16]:
        print(f"{t:34s} x{k['dispatches']:4d}  mfma busy {k['mfma_busy_frac']:.3f}  valu {k['valu_active_of_wave_cycles']}  wait {k['wait_any_of_wave_cycles']}")

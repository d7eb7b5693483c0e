"""Per-kernel SQ counters from one rocprofv3 --pmc pass (counter_collection.csv) -> JSON + a short table.
    python tools/pmc_counters.py p_counter_collection.csv out.json
Derived: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) with cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports the sum over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'); lds_conflict_frac =
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles)."""
import collections, csv, json, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import short

agg = collections.defaultdict(lambda: collections.defaultdict(float))
launch = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launch[k].add(r["Dispatch_Id"])
out = {}
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
    n = max(1, len(launch[k]))
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    rec = {"launches": n, "per_launch": {name: v / n for name, v in c.items()}}
    if cyc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        rec["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
        rec["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVE_CYCLES", 0.0) > 0:
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if name in c:
                rec[name.lower() + "_share_of_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
    out[k] = rec
json.dump({"source": "rocprofv3 --pmc (one pass, SQ block + GRBM_GUI_ACTIVE), sums over all launches of the bench run",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k in list(out)[:8]:
    r = out[k]
    print(f"{k[:70]:70s} x{r['launches']:<4d} mfma_busy {r.get('mfma_busy_frac', float('nan')):.3f}  lds_conflict {r.get('lds_conflict_frac', float('nan')):.4f}")

"""Copy the summaries of one tools/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/ under per-round names.
    python tools/profile_collect.py r03"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")


def first(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


def copy(pattern, name):
    f = first(pattern)
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}"))
        print("copied", name)
    else:
        print("MISSING", pattern)


def bench_line(path, name):
    if os.path.exists(path):
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, f"{tag}_{name}"), "w"), indent=1)
            print("wrote", name)


copy("stats/**/*kernel_stats.csv", "bench_default_kernel_stats.csv")
bench_line(os.path.join(src, "bench_under_stats.json"), "bench_default_under_rocprof.json")
f, w = first("fetch/**/*counter_collection.csv"), first("write/**/*counter_collection.csv")
if f and w:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), f, w, os.path.join(dst, f"{tag}_hbm_traffic.json")])
sq = first("sq/**/*counter_collection.csv")
if sq:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_counters.py"), sq, os.path.join(dst, f"{tag}_mfma_lds_counters.json")])
for m in ("x3", "dec_f16"):
    copy(f"mode_{m}/**/*kernel_stats.csv", f"mode_{m}_kernel_stats.csv")
    sq = first(f"mode_{m}_sq/**/*counter_collection.csv")
    if sq:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_counters.py"), sq, os.path.join(dst, f"{tag}_mode_{m}_mfma_lds_counters.json")])
for k in ("dct", "blur"):
    copy(f"{k}/**/*kernel_stats.csv", f"slot_{k}_kernel_stats.csv")
    bench_line(os.path.join(src, f"slot_{k}.json"), f"slot_{k}.json")

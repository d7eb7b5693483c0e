"""Per-kernel HBM traffic from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs,
MI355X_MICROARCH.md §HBM):
    python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for
wide (16 B/lane) reads, so it is doubled as the guide prescribes.  Output: bytes per launch, per kernel."""
import collections, csv, json, re, sys


def short(name: str) -> str:
    """Kernel name as bench.py / elvis_conv_kernel_name print it, from rocprofv3's mangled or demangled form."""
    m = re.search(r"(conv3x3_halo_kernel|conv_igemm_kernel)I(DF16_|f)((?:L[ib]\d+E)+)", name)
    if m:
        args = re.findall(r"L([ib])(\d+)E", m.group(3))
        vals = [("true" if v == "1" else "false") if k == "b" else v for k, v in args]
        return f"{m.group(1)}<{'half' if m.group(2) == 'DF16_' else 'float'},{','.join(vals)}>"
    m = re.search(r"\d\d([a-z]\w*?_kernel)I((?:L[ib]\d+E)+)", name)   # other mangled templates: integer / bool arguments only
    if m:
        args = re.findall(r"L([ib])(\d+)E", m.group(2))
        vals = [("true" if v == "1" else "false") if k == "b" else v for k, v in args]
        return f"{m.group(1)}<{','.join(vals)}>"
    d = re.search(r"(\w+_kernel\w*)<([^>]*)>", name)   # demangled form
    if d:
        return f"{d.group(1)}<{d.group(2).replace(' ', '')}>"
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([a-z]\w*?_kernel)", name)    # mangled, non-integer template arguments
    if m:
        return m.group(1)
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:80]


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 1])[0] + write.get(k, [0, 1])[0])):
        f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
        out[k] = {"launches": max(f[1], w[1]),
                  "fetch_bytes_per_launch": 2.0 * 1024.0 * f[0] / max(f[1], 1),   # x2: gfx950 FETCH_SIZE correction
                  "write_bytes_per_launch": 1024.0 * w[0] / max(w[1], 1)}
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE doubled (gfx950)",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k in list(out)[:8]:
        print(k, out[k])


if __name__ == "__main__":
    main()

"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a small text report."""
import csv, glob, os, sys, collections

def short(name):
    """'void svgf::(anonymous namespace)::atrous_lds_kernel<0, 4>(svgf::Geo, ...)' or its mangled form -> 'atrous_lds_kernel<ST=0,S=4>'."""
    import re
    for k in ("temporal_kernel", "moments_young_kernel", "moments_lds_kernel", "moments_kernel", "atrous_lds_kernel", "atrous_fused12_kernel", "atrous_direct_kernel"):
        if k in name:
            m = re.search(k + r"<(\d+)(?:, (\d+))?", name) or re.search(k + r"ILi(\d+)E(?:Li(\d+)E)?", name)
            if m and k == "atrous_lds_kernel" and m.group(2) is not None:
                return f"{k}<ST={m.group(1)},S={m.group(2)}>"
            return f"{k}<ST={m.group(1)}>" if m else k
    return name[:60]

def main(d):
    out = []
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        out.append(f"== kernel stats ({os.path.relpath(f, d)})")
        for r in csv.DictReader(open(f)):
            out.append(f"{short(r['Name']):44s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {float(r['AverageNs']):>12.0f} pct {r['Percentage']}")
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out.append(f"== counters ({os.path.relpath(f, d)}) mean per dispatch")
            for k in sorted(agg):
                if "svgf" not in k and "kernel" not in k:
                    continue
                out.append(f"{k:44s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(agg[k].items())) + f"  (n={len(next(iter(agg[k].values())))})")
    print("\n".join(out))

if __name__ == "__main__":
    main(sys.argv[1])

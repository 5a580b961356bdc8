"""Builds profiles/hbm_traffic.json from the PMC passes of tools/rocprof_bench.sh (gpurun_out/prof_<tag>/pmc_*).
usage: python tools/hbm_traffic.py <tag> [round-name]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else tag


def mean_counter(kind, mode):
    vals = []
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "pmc_%s_%s" % (kind, mode), "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "ssqp_solve" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals)


out = {"config": "cfg4", "nprob": 1024, "round": rnd,
       "correction": "FETCH_SIZE x2 on gfx950 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); "
                     "WRITE_SIZE exact; separate --pmc passes (tools/rocprof_bench.sh); KiB units"}
for mode, key in (("default", "default_formulation"), ("dense", "dense_formulation")):
    f, w = mean_counter("fetch", mode), mean_counter("write", mode)
    out[key] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
out["hbm_bytes_per_launch"] = out["default_formulation"]["hbm_bytes_per_launch"]
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

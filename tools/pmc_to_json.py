"""Builds gpurun_out/prof_<tag>/pmc_counters.json from the PMC passes of tools/profile_round.sh: per-launch means
of every counter, summed over the two solve kernels of a launch (wavefront kernel + workgroup kernel), stamped with
the sha256 of the kernel sources so that bench.py only quotes them for the code they were taken on.
usage: python tools/pmc_to_json.py <tag>   (copy the result to profiles/pmc_counters.json)"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)


def per_launch(passdir, only=None):
    """{counter: mean over launches of the sum over the solve kernels of one launch}; only = substring a kernel name
    must contain"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))   # counter -> kernel -> values per dispatch
    for f in glob.glob(os.path.join(base, passdir, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if only is not None and only not in kn:
                continue
            if "ssqp_solve_kernel" in kn or "ssqp_wave_kernel" in kn:
                # one class per kernel of the hand-over chain (each is launched once per batch, possibly on an empty list)
                cls = kn.split("(ssqp::SolveParams")[0].split("ssqp_")[-1]
                # (the lean and the accounting build of a wave kernel are ONE class: bench.py runs one accounting launch and
                #  times the lean ones -- the counters of a launch are their mean, not their sum)
                cls = cls.replace(", true>", ">").replace(", false>", ">")
                acc[r["Counter_Name"]][cls].append(float(r["Counter_Value"]))
    out = {}
    for c, d in acc.items():
        out[c] = sum(sum(v) / len(v) for v in d.values())
    return out


res = {"config": "cfg4", "nprob": 1024, "tag": tag, "kernel_source_sha256": bench.kernel_source_hash(),
       "correction": "FETCH_SIZE x2 on gfx950 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); "
                     "WRITE_SIZE exact; separate --pmc passes (tools/profile_round.sh); rocprofv3 reports KiB; "
                     "figures are per launch = wavefront kernel + workgroup kernel of one batch"}
for mode, key in (("default", "default_formulation"), ("dense", "dense_formulation")):
    f = per_launch("pmc_fetch_" + mode).get("FETCH_SIZE")
    w = per_launch("pmc_write_" + mode).get("WRITE_SIZE")
    if f is None or w is None:
        continue
    res[key] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
sq = {}
for p in ("pmc_sq1_default", "pmc_sq2_default", "pmc_sq3_default"):
    sq.update(per_launch(p))
if sq and "default_formulation" in res:
    res["default_formulation"]["sq"] = sq
# the eight-per-CU build (what the launch lanes of the timed region run), from the lanes passes
f8 = per_launch("pmc_fetch_lanes", "ssqp_wave_kernel<2, true, 2,").get("FETCH_SIZE")
w8 = per_launch("pmc_write_lanes", "ssqp_wave_kernel<2, true, 2,").get("WRITE_SIZE")
if f8 is not None and w8 is not None:
    res["eight_per_cu_build"] = {"FETCH_SIZE_KiB": f8, "WRITE_SIZE_KiB": w8, "hbm_bytes_per_launch": (2.0 * f8 + w8) * 1024.0,
                                 "sq": per_launch("pmc_sq1_lanes", "ssqp_wave_kernel<2, true, 2,")}
# cfg3 (big-factor build): written beside, keyed the same way (bench.py --config cfg3 --pmc-json <that file>)
f3 = per_launch("pmc_fetch_cfg3").get("FETCH_SIZE")
w3 = per_launch("pmc_write_cfg3").get("WRITE_SIZE")
if f3 is not None and w3 is not None:
    res3 = {"config": "cfg3", "nprob": 1024, "tag": tag, "kernel_source_sha256": bench.kernel_source_hash(),
            "correction": res["correction"],
            "default_formulation": {"FETCH_SIZE_KiB": f3, "WRITE_SIZE_KiB": w3, "hbm_bytes_per_launch": (2.0 * f3 + w3) * 1024.0,
                                    "sq": {**per_launch("pmc_sq1_cfg3"), **per_launch("pmc_sq3_cfg3")}}}
    json.dump(res3, open(os.path.join(base, "pmc_counters_cfg3.json"), "w"), indent=1)
    print(json.dumps(res3, indent=1))
# the large-K path (blocked LDL' with f64 MFMA tiles) on the K -> 320 workload: MFMA counters of the workgroup kernel
mf = per_launch("pmc_mfma_k320", "ssqp_solve_kernel")
if mf:
    log = open(os.path.join(base, "mfma_k320.log")).read() if os.path.exists(os.path.join(base, "mfma_k320.log")) else ""
    json.dump({"workload": "gen:320,1,4,640,0.5,0.0,1.0,0.0 x 256 QPs, wave_kernel=0", "tag": tag,
               "kernel_source_sha256": bench.kernel_source_hash(), "counters_per_launch": mf,
               "mfma_busy_over_cu_busy": mf.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(mf.get("SQ_BUSY_CU_CYCLES", 1), 1),
               "debug_parity_line": [l for l in log.splitlines() if l.startswith("gen:")][-1:]},
              open(os.path.join(base, "pmc_mfma_k320.json"), "w"), indent=1)
json.dump(res, open(os.path.join(base, "pmc_counters.json"), "w"), indent=1)
print(json.dumps(res, indent=1))

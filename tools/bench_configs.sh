#!/bin/bash
# GPU box: one bench.py line per BASELINE.json configuration (the parity-test configurations; cfg4 is the headline)
mkdir -p gpurun_out/configs
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > gpurun_out/configs/$name.json 2> gpurun_out/configs/$name.err || echo "FAILED $name: $(tail -2 gpurun_out/configs/$name.err)"; }
run cfg1 --config cfg1 --nprob 1024 --steps 10 --warmup 2 --cpu-seconds 4
run cfg2 --config cfg2 --nprob 1 --steps 10 --warmup 2 --cpu-seconds 3
run cfg3 --config cfg3 --nprob 1024 --steps 6 --warmup 2 --cpu-seconds 6 --pmc-json profiles/pmc_counters_cfg3.json
# (the same with the caller's option for workloads known to end with large free sets: the big-factor build from the first pass)
run cfg3_wave_kernel2 --config cfg3 --nprob 1024 --steps 6 --warmup 2 --no-cpu --skip-dense --option wave_kernel=2
run cfg4 --config cfg4 --nprob 1024 --steps 20 --warmup 3 --cpu-seconds 8
run cfg4_serial --config cfg4 --nprob 1024 --steps 6 --warmup 1 --streams 1 --no-cpu --skip-dense
run cfg5 --config cfg5 --nprob 1 --steps 3 --warmup 1 --cpu-seconds 3
# BASELINE config 4 whole on ONE GPU (8,192 QPs, 16 GiB of V per batch): the strong-scaling anchor; one lane and the default
run cfg4_8192_serial --config cfg4 --nprob 8192 --steps 3 --warmup 1 --streams 1 --no-cpu --skip-dense --repeats 3
run cfg4_8192 --config cfg4 --nprob 8192 --steps 4 --warmup 2 --streams 2 --no-cpu --skip-dense --repeats 3
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "no json", e); continue
    cb=d.get("cpu_baseline") or {}; cl=d.get("cpu_baseline_lapack") or {}
    e2=d.get("end_to_end_solveQP") or {}
    print("%-12s value %10.1f QPs/s  ms/step %8.3f  mode %s  serial qps %.0f  kernel_ms %.3f  iters %.1f  conv %s  wave %.2f ho %d big %d | cpu port %.1f lapack %.1f (cores %s) | e2e %s | roofline frac %.3f traffic %s" % (
        os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["pipeline"]["mode"], d["roofline_serial"]["qps"], d["roofline"]["kernel_ms"],
        d["iters_to_kkt"]["mean"], d["all_converged"], d["kernels"]["wavefront_kernel_share"], d["kernels"]["handed_over"], d["kernels"]["continued_in_big_factor_wave_kernel"],
        cb.get("value",0), cl.get("value",0), cb.get("cores"), e2.get("qps", e2.get("error")), d["roofline"]["frac"], d["roofline"]["traffic"]))
PY

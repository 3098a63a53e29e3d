#!/bin/bash
# HBM traffic of the GCN propagate kernel (cache-resident and HBM-resident batch) from PMC counters, separate passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for B in 256 2048; do
  B=$B N=10 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_pf$B -- python3 scripts/prof_one.py prop > /dev/null 2>&1 &&
  B=$B N=10 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_pw$B -- python3 scripts/prof_one.py prop > /dev/null 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json
out = {}
for B in (256, 2048):
    def med(d, c):
        f = glob.glob("gpurun_out/%s/*/*_counter_collection.csv" % d)[0]
        v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gcn_propagate" in r["Kernel_Name"] and r["Counter_Name"] == c)
        return v[len(v) // 2]
    fe, wr = med("pmc_pf%d" % B, "FETCH_SIZE"), med("pmc_pw%d" % B, "WRITE_SIZE")
    out["dd_b%d_f128" % B] = {"FETCH_SIZE_KiB_raw": fe, "WRITE_SIZE_KiB_raw": wr, "hbm_read_bytes": 2 * fe * 1024,
                              "hbm_write_bytes": wr * 1024, "traffic_bytes_per_launch": 2 * fe * 1024 + wr * 1024}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/prop_traffic.json", "w"), indent=1)
PY

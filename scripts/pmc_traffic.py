#!/usr/bin/env python3
"""Parse gpurun_out/pmc_* (scripts/pmc_traffic.sh) -> profiles/r01/agg_traffic.json.
gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half of the
bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections, csv, glob, json, os

def med(dirn, counter, kernel="spmm"):
    f = glob.glob("gpurun_out/%s/*/*_counter_collection.csv" % dirn)[0]
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter)
    return v[len(v) // 2]

out = {}
for tag, fd, wd in (("dd_b32_rows9151_f128", "pmc_f", "pmc_w"), ("dd_b2048_f128", "pmc_f2k", "pmc_w2k")):
    fetch, write = med(fd, "FETCH_SIZE"), med(wd, "WRITE_SIZE")
    out[tag] = {"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
                "hbm_read_bytes": 2 * fetch * 1024, "hbm_write_bytes": write * 1024,
                "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024}
if glob.glob("gpurun_out/pmc_gf/*/*_counter_collection.csv"):
    fetch, write = med("pmc_gf", "FETCH_SIZE", "rowgemm_"), med("pmc_gw", "WRITE_SIZE", "rowgemm_")      # the gather kernel (rowgemm_gather_ks2_kernel)
    out["dd_b32_gather_rowgemm_k128_n128"] = {"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
                                              "hbm_read_bytes": 2 * fetch * 1024, "hbm_write_bytes": write * 1024,
                                              "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024}
os.makedirs("profiles/r01", exist_ok=True)
json.dump(out, open("profiles/r01/agg_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))

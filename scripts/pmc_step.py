#!/usr/bin/env python3
"""Parse gpurun_out/pmc_step_{f,w} (scripts/pmc_step.sh) -> profiles/<round>/step_traffic.json: HBM bytes per launch of each kernel of
the timed step.  gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports exactly half
of the bytes of wide (16 B / lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections, csv, glob, json, os, re, sys
ROUND = os.environ.get("TSGNN_ROUND", "r04")
PREFIX = sys.argv[1] if len(sys.argv) > 1 else "pmc_step"            # gpurun_out/<PREFIX>_{f,w}
OUTNAME = sys.argv[2] if len(sys.argv) > 2 else "step_traffic.json"

def per_kernel(dirn, counter):
    f = glob.glob("gpurun_out/%s/*/*_counter_collection.csv" % dirn)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}

def short(name):
    """'void (anonymous namespace)::sage_layer_fwd_kernel<true>(...)' -> 'sage_layer_fwd_kernel<true>'"""
    m = re.search(r"\(anonymous namespace\)::([A-Za-z0-9_]+(<[^(]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name

F, W = per_kernel(PREFIX + "_f", "FETCH_SIZE"), per_kernel(PREFIX + "_w", "WRITE_SIZE")
out = {}
for k in F:
    if "anonymous namespace" not in k:
        continue
    fetch, write = F[k], W.get(k, 0.0)
    out[short(k)] = {"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write, "hbm_read_bytes": 2 * fetch * 1024,
                     "hbm_write_bytes": write * 1024, "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024}
os.makedirs("profiles/" + ROUND, exist_ok=True)
json.dump(out, open("profiles/%s/%s" % (ROUND, OUTNAME), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

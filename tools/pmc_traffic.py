"""HBM-side traffic per launch from two rocprofv3 PMC passes of bench.py (one with --pmc FETCH_SIZE, one with --pmc
WRITE_SIZE; the TCC block cannot hold both at once).  Units and corrections as MI355X_MICROARCH.md prescribes:
FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) reads, so it is
doubled; WRITE_SIZE is used as reported (checked on opt_kernel: 2 x 1.84 GB fetched vs 16 B/param x 235 M params,
3.22 GB written vs 14 B/param).     python tools/pmc_traffic.py fetch.csv write.csv > profiles/rNN_pmc.json"""
import collections, csv, json, sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
        agg[name].append(float(r["Counter_Value"]) * 1024.0)
    return agg


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in f:
    nf, nw = len(f[k]), len(w.get(k, []))
    fetch = 2.0 * sum(f[k]) / nf
    write = sum(w[k]) / nw if nw else 0.0
    out[k] = {"launches": nf, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
              "hbm_bytes_per_launch": round(fetch + write)}
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                     "--warmup 2 --no-cpu-baseline --no-profile (train mode; all launches of the run averaged, tuning launches "
                     "included); FETCH_SIZE doubled (gfx950 correction), KB -> bytes",
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}, sys.stdout, indent=1)

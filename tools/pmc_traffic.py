"""HBM-side traffic per launch from two rocprofv3 PMC passes of bench.py (one with --pmc FETCH_SIZE, one with --pmc
WRITE_SIZE; the TCC block cannot hold both at once).  Units and corrections as MI355X_MICROARCH.md prescribes:
FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) reads, so it is
doubled; WRITE_SIZE is used as reported (checked on opt_kernel: 2 x 1.84 GB fetched vs 16 B/param x 235 M params,
3.22 GB written vs 14 B/param).
    python tools/pmc_traffic.py fetch.csv write.csv [gemm_bytes_log.json] > profiles/rNN_pmc.json
gemm_bytes_log.json (round 4): what the FETCH pass's process wrote under SMX_GEMM_BYTES_LOG - the ALGORITHMIC bytes of every bf16
GEMM launch in launch order (speechmix_amd/ops.py).  Each such launch dispatches exactly one `gemm_bf16_*` kernel, so the list is
matched to the pass's `gemm_bf16_*` rows in dispatch order and every GEMM instantiation gets `algorithmic_bytes_per_launch` and
`traffic_over_algorithmic` beside its measured bytes (values well above 1: operand panels re-read from HBM / the fabric)."""
import collections, csv, json, sys


def rows(path):
    out = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
        out.append((int(r.get("Dispatch_Id", len(out))), name, float(r["Counter_Value"]) * 1024.0))
    out.sort()
    return out


def load(path):
    agg = collections.defaultdict(list)
    for _, name, v in rows(path):
        agg[name].append(v)
    return agg


f, w = load(sys.argv[1]), load(sys.argv[2])
alg = {}
note = ""
if len(sys.argv) > 3:
    log = json.load(open(sys.argv[3]))
    gem = [(d, n) for d, n, _ in rows(sys.argv[1]) if n.startswith("gemm_bf16_")]
    if len(gem) == len(log):
        agg = collections.defaultdict(list)
        for (_, n), b in zip(gem, log):
            agg[n].append(b)
        alg = {k: sum(v) / len(v) for k, v in agg.items()}
        note = "; algorithmic bytes: SMX_GEMM_BYTES_LOG of the FETCH pass matched to its gemm_bf16_* dispatches in order"
    else:
        note = f"; algorithmic bytes NOT attached: {len(log)} logged launches vs {len(gem)} gemm_bf16_* dispatches"
out = {}
for k in f:
    nf, nw = len(f[k]), len(w.get(k, []))
    fetch = 2.0 * sum(f[k]) / nf
    write = sum(w[k]) / nw if nw else 0.0
    out[k] = {"launches": nf, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
              "hbm_bytes_per_launch": round(fetch + write)}
    if k in alg:
        out[k]["algorithmic_bytes_per_launch"] = round(alg[k])
        out[k]["traffic_over_algorithmic"] = round((fetch + write) / max(alg[k], 1.0), 3)
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                     "--warmup 2 --no-cpu-baseline --no-profile (train mode; all launches of the run averaged, tuning launches "
                     "included); FETCH_SIZE doubled (gfx950 correction), KB -> bytes" + note,
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}, sys.stdout, indent=1)

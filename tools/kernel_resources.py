#!/usr/bin/env python3
"""Register / scratch / LDS footprint of every kernel of the library (hipcc -S of each .hip, AMDGPU metadata):
    python tools/kernel_resources.py [out.json]            # dump
    python tools/kernel_resources.py --diff a.json b.json   # what changed
A source edit that pushes a hand-scheduled kernel into scratch or over an occupancy step shows up here before a GPU run."""
import concurrent.futures as cf
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "speechmix_amd", "csrc")


def one(f):
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-I", SRC, "-S",
                          "--cuda-device-only", os.path.join(SRC, f), "-o", "-"], capture_output=True, text=True, check=True).stdout
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, flags=re.S):
        name, body = m.group(1), m.group(2)
        def g(key):
            mm = re.search(r"\.amdhsa_" + key + r"\s+(\S+)", body)
            return mm.group(1) if mm else None
        out[name] = dict(vgpr=g("next_free_vgpr"), sgpr=g("next_free_sgpr"), scratch=g("private_segment_fixed_size"),
                         lds=g("group_segment_fixed_size"), accum_offset=g("accum_offset"))
    return f, out


def dump(path):
    files = sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))
    res = {}
    with cf.ThreadPoolExecutor(6) as ex:
        for f, out in ex.map(one, files):
            res.update({f + ":" + k: v for k, v in out.items()})
    json.dump(res, open(path, "w"), indent=0, sort_keys=True)
    print(len(res), "kernels ->", path)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--diff"]:
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print(k, a.get(k), "->", b.get(k))
    else:
        dump(sys.argv[1] if len(sys.argv) > 1 else "/tmp/kernel_resources.json")

"""Build libspeechmix_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python speechmix_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU.  One object per .hip file (parallel), linked into one shared object
next to the package so it travels to the GPU box with the source snapshot.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libspeechmix_hip.so")
OBJ = os.path.join(HERE, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
if os.environ.get("SMX_DEFS"):        # extra -D switches for A/B builds, comma separated
    FLAGS += ["-D" + d for d in os.environ["SMX_DEFS"].split(",")]
if os.environ.get("SMX_PP_LAB"):      # ablation variants of the ping-pong GEMM (lab builds)
    FLAGS.append("-DSMX_PP_LAB")


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs += [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "speechmix_hip.h")]
    hdrs = [h for h in hdrs if os.path.exists(h)]
    jobs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer(src, obj, hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        # SMX_TU: the file's stem - names the translation unit's step-key word (smx_common.h)
        cmd = ["hipcc"] + FLAGS + ["-DSMX_TU=" + os.path.basename(src)[:-4], "-I", HERE, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, log in ex.map(cc, jobs):
                if verbose:
                    print(f"[hipcc] {os.path.basename(src)} rc={rc}")
                if rc != 0:
                    sys.stderr.write(log)
                    raise RuntimeError(f"hipcc failed on {src}")
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if jobs or not os.path.exists(OUT):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("[link]", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

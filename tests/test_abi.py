"""CPU checks of the C-ABI boundary: the shared object builds/loads, exports every symbol that
include/speechmix_hip.h declares, and its struct layouts agree with the ctypes mirrors and with the header
as seen by a plain C compiler.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "speechmix_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from speechmix_amd import _lib as L
    return L.lib()


def _declared():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"^\s*(?:int|double|long long|void|size_t)\s+(smx_\w+)\s*\(", src, flags=re.M)))


def test_every_exported_symbol_is_declared(lib):
    """The other direction: nothing the library exports under the smx_ prefix may be missing from the header."""
    from speechmix_amd import _lib as L
    path = L.lib()._name
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("smx_")})
    assert len(exported) >= 80
    missing = [n for n in exported if n not in set(_declared())]
    assert not missing, f"exported but not declared in include/speechmix_hip.h: {missing}"


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/speechmix_hip.h but not exported"


def test_struct_layouts_match_ctypes(lib):
    from speechmix_amd import _lib as L
    for name, st in (("SmxGemmParams", L.GemmParams), ("SmxNormParams", L.NormParams), ("SmxNormBwdParams", L.NormBwdParams),
                     ("SmxAttnParams", L.AttnParams), ("SmxConv0Params", L.Conv0Params), ("SmxCEParams", L.CEParams),
                     ("SmxOptParams", L.OptParams), ("SmxWsumParams", L.WsumParams), ("SmxAfParams", L.AfParams),
                     ("SmxAfTensor", L.AfTensor), ("SmxAfTile", L.AfTile), ("SmxAfSeg", L.AfSeg), ("SmxFoldTable", L.FoldTable),
                     ("SmxTrTable", L.TrTable)):
        assert getattr(lib, "smx_sizeof_" + name)() == C.sizeof(st), name
    assert lib.smx_fold_max() == L.FOLD_MAX
    assert lib.smx_tr_max() == L.TR_MAX


def test_header_is_plain_c_and_agrees_with_library(lib, tmp_path):
    """The header must be consumable from C (the drop-in boundary is a C ABI) and give the same sizes."""
    prog = tmp_path / "sz.c"
    prog.write_text('#define __HIP_PLATFORM_AMD__ 1\n#include <stdio.h>\n#include "speechmix_hip.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(SmxGemmParams), sizeof(SmxNormParams),'
                    ' sizeof(SmxNormBwdParams), sizeof(SmxAttnParams), sizeof(SmxConv0Params), sizeof(SmxCEParams),'
                    ' sizeof(SmxOptParams), sizeof(SmxWsumParams), sizeof(SmxAfParams), sizeof(SmxAfTensor), sizeof(SmxAfTile), sizeof(SmxAfSeg));'
                    ' return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", str(prog), "-o", str(exe)],
                   check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [getattr(lib, "smx_sizeof_" + n)() for n in ("SmxGemmParams", "SmxNormParams", "SmxNormBwdParams", "SmxAttnParams",
                                                         "SmxConv0Params", "SmxCEParams", "SmxOptParams", "SmxWsumParams", "SmxAfParams",
                                                         "SmxAfTensor", "SmxAfTile", "SmxAfSeg")]
    assert got == want


def test_product_path_fails_loudly_without_gpu_or_library():
    import torch
    from speechmix_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    t = torch.zeros(8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(t, t, t, 8, 8, 8, ops.F32)


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "speechmix_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_pingpong_gemm_isa_has_no_spills_and_no_queue_drain():
    """The ping-pong GEMM's schedule depends on two properties of the generated code that a source edit can silently
    break: no register spills (a scratch reload makes hipcc wait vmcnt(0), draining the LDS-DMA prefetch
    every K tile) and no compiler-inserted vmcnt wait in front of the K loop's first LDS read (any VMEM operation hipcc
    still tracks there - an epilogue load consumed on only some paths - has the same effect).  Checked on the ISA of every
    instantiation (hipcc cross-compiles without a GPU)."""
    src = os.path.join(ROOT, "speechmix_amd", "csrc", "gemm_pp.hip")
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-I", os.path.dirname(src),
                          "-S", "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True, check=True).stdout
    kernels = re.findall(r"^(_Z19gemm_bf16_pp_kernel\w+):.*?s_endpgm", asm, flags=re.M | re.S)
    bodies = re.findall(r"^_Z19gemm_bf16_pp_kernel\w+:.*?s_endpgm", asm, flags=re.M | re.S)
    assert len(bodies) >= 9, len(bodies)
    for name, body in zip(kernels, bodies):
        assert "scratch_" not in body, f"{name}: register spill"
        lines = body.splitlines()
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        lo, hi = max(0, mf[0] - 120), mf[-1]               # the K loop (its cold work-list path included)
        in_asm, bad = False, []
        for l in lines[lo:hi]:
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif not in_asm and "s_waitcnt" in l and "vmcnt" in l:
                bad.append(l.strip())
        assert not bad, f"{name}: compiler-inserted VMEM wait inside the K loop: {bad}"


def test_allreduce_bucket_entry_validates_and_reports_missing_rccl_without_crashing(lib):
    """smx_allreduce_bucket: exported, validates its arguments, and an empty bucket is a no-op.
    (A real collective needs >= 1 GPU and a communicator: tests/test_gpu_r2.py drives it through RCCL on the GPU box.)"""
    fn = lib.smx_allreduce_bucket
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert fn(None, None, 16, 0, None) == -22                    # no communicator / buffer
    assert fn(C.c_void_p(1), C.c_void_p(1), 0, 0, None) == 0      # empty bucket
    assert fn(C.c_void_p(1), C.c_void_p(1), 16, 7, None) == -22   # unknown dtype

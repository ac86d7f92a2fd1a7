"""ctypes binding of libspeechmix_hip.so (the C-ABI drop-in boundary, see include/speechmix_hip.h).

The product path has NO fallback: if the shared object is missing or a symbol is absent this module
raises, and every op wrapper raises RuntimeError on a non-zero return code.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported BEFORE the shared object: one HIP runtime per process - torch's)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMX_LIB") or os.path.join(_HERE, "libspeechmix_hip.so")     # SMX_LIB: A/B builds (tools/lab)

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


class RowView(C.Structure):
    _fields_ = [("batch_stride", C.c_longlong), ("ld", C.c_longlong), ("off", C.c_longlong),
                ("rows_per_batch", C.c_int), ("_pad", C.c_int)]


class GemmParams(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("resid", C.c_void_p), ("aux_out", C.c_void_p), ("aux_in", C.c_void_p),
                ("a", RowView), ("b", RowView), ("c", RowView), ("e", RowView),
                ("batch_a", C.c_longlong), ("batch_b", C.c_longlong), ("batch_c", C.c_longlong),
                ("batch_bias", C.c_longlong), ("batch_e", C.c_longlong),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("a_rc", C.c_int), ("b_rc", C.c_int),
                ("act", C.c_int), ("out_f32", C.c_int), ("atomic", C.c_int), ("nbatch", C.c_int),
                ("split_k", C.c_int), ("tr_mode", C.c_int), ("alpha", C.c_float),
                ("split_stride", C.c_longlong), ("drop_p", C.c_float), ("drop_seed", C.c_uint)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python speechmix_amd/csrc/build.py` "
                "(speechmix_amd has no CPU or PyTorch fallback for its kernels)")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f"speechmix_hip: {name} failed with code {rc}")


class NormParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("pos", C.c_void_p), ("xsum_out", C.c_void_p), ("y", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("M", C.c_int), ("D", C.c_int), ("pos_period", C.c_int), ("pos_offset", C.c_int),
                ("rms", C.c_int), ("act", C.c_int), ("eps", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint)]


class NormBwdParams(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dres", C.c_void_p), ("dx", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dpos", C.c_void_p), ("partials", C.c_void_p),
                ("M", C.c_int), ("D", C.c_int), ("pos_period", C.c_int), ("pos_offset", C.c_int),
                ("rms", C.c_int), ("act", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_uint), ("defer_fold", C.c_int),
                ("dx_drop", C.c_void_p), ("drop2_p", C.c_float), ("drop2_seed", C.c_uint)]


FOLD_MAX = 48


class FoldEntry(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dst", C.c_void_p), ("nrows", C.c_int), ("ncols", C.c_int), ("ld", C.c_longlong),
                ("alpha", C.c_float), ("pad", C.c_int)]


class FoldTable(C.Structure):
    _fields_ = [("n", C.c_int), ("pad", C.c_int), ("e", FoldEntry * FOLD_MAX)]


TR_MAX = 64


class TrEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("tile0", C.c_int), ("tcols", C.c_int)]


class TrTable(C.Structure):
    _fields_ = [("n", C.c_int), ("tiles", C.c_int), ("e", TrEntry * TR_MAX)]


class AttnParams(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p), ("lse", C.c_void_p),
                ("bias", C.c_void_p), ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
                ("delta", C.c_void_p), ("dbias", C.c_void_p),
                ("q_bs", C.c_longlong), ("q_ld", C.c_longlong), ("k_bs", C.c_longlong), ("k_ld", C.c_longlong),
                ("v_bs", C.c_longlong), ("v_ld", C.c_longlong), ("o_bs", C.c_longlong), ("o_ld", C.c_longlong),
                ("dq_bs", C.c_longlong), ("dq_ld", C.c_longlong), ("dk_bs", C.c_longlong), ("dk_ld", C.c_longlong),
                ("dv_bs", C.c_longlong), ("dv_ld", C.c_longlong), ("do_bs", C.c_longlong), ("do_ld", C.c_longlong),
                ("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("D", C.c_int),
                ("causal", C.c_int), ("scale", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint),
                ("mask_q", C.c_void_p), ("mask_k", C.c_void_p), ("klen", C.c_void_p)]


class Conv0Params(C.Structure):
    _fields_ = [("wave", C.c_void_p), ("w", C.c_void_p), ("cbias", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("stats", C.c_void_p), ("y", C.c_void_p), ("dy", C.c_void_p),
                ("bstats", C.c_void_p), ("dw", C.c_void_p), ("dcbias", C.c_void_p), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p),
                ("B", C.c_int), ("N", C.c_int), ("C", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("T0", C.c_int),
                ("group", C.c_int), ("eps", C.c_float), ("tiles_per_block", C.c_int), ("partials", C.c_void_p), ("nb", C.c_int)]


class CEParams(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("labels", C.c_void_p), ("loss", C.c_void_p), ("argmax", C.c_void_p),
                ("dlogits", C.c_void_p), ("lse", C.c_void_p), ("M", C.c_int), ("V", C.c_int),
                ("ldl", C.c_longlong), ("ldd", C.c_longlong), ("gscale", C.c_float),
                ("logits_t", C.c_void_p), ("kld", C.c_void_p), ("kld_scale", C.c_float),
                ("count_labels", C.c_void_p), ("count_M", C.c_int)]


class OptParams(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("shadow", C.c_void_p),
                ("gnorm_sq", C.c_void_p), ("n", C.c_longlong),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("bias_c1", C.c_float), ("bias_c2", C.c_float),
                ("grad_scale", C.c_float), ("max_grad_norm", C.c_float), ("kind", C.c_int)]


class AfTensor(C.Structure):
    _fields_ = [("off", C.c_longlong), ("nb", C.c_int), ("R", C.c_int), ("C", C.c_int), ("row_off", C.c_int),
                ("col_off", C.c_int), ("rm_off", C.c_int), ("factored", C.c_int), ("tile0", C.c_int), ("ntile", C.c_int),
                ("_pad", C.c_int)]


class AfTile(C.Structure):
    _fields_ = [("tensor", C.c_int), ("b", C.c_int), ("r0", C.c_int), ("nr", C.c_int), ("c0", C.c_int), ("nc", C.c_int),
                ("full_rows", C.c_int), ("full_cols", C.c_int), ("cp_off", C.c_int), ("rp_off", C.c_int)]


class AfSeg(C.Structure):
    _fields_ = [("tensor", C.c_int), ("b", C.c_int), ("cp_off", C.c_int), ("n_rt", C.c_int), ("rp_off", C.c_int), ("n_ct", C.c_int)]


class AfParams(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("shadow", C.c_void_p), ("tensors", C.c_void_p), ("tiles", C.c_void_p),
                ("segs", C.c_void_p), ("row", C.c_void_p), ("col", C.c_void_p), ("racc", C.c_void_p), ("cacc", C.c_void_p),
                ("rmean", C.c_void_p), ("usq", C.c_void_p), ("usq_part", C.c_void_p), ("cpart", C.c_void_p), ("beta2t", C.c_void_p),
                ("gn2", C.c_void_p), ("gsq_part", C.c_void_p),
                ("racc_n", C.c_longlong), ("cacc_n", C.c_longlong), ("ntensors", C.c_int), ("ntiles", C.c_int),
                ("nsegs", C.c_int), ("lr", C.c_float), ("eps1", C.c_float), ("clip_threshold", C.c_float),
                ("grad_scale", C.c_float), ("max_grad_norm", C.c_float)]


class WsumParams(C.Structure):
    _fields_ = [("h", C.c_void_p * 40), ("w", C.c_void_p), ("out", C.c_void_p), ("dy", C.c_void_p), ("dots", C.c_void_p),
                ("dw", C.c_void_p), ("sw", C.c_void_p), ("n", C.c_longlong), ("L1", C.c_int)]

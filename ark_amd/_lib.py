"""ctypes binding of the ark_amd C-ABI (include/ark_amd.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
There is deliberately NO fallback: if the library is missing, every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARK_AMD_LIB: explicit path of an alternative build of the SAME library (timing ablations, tools/ablate_diag.sh)
LIB_PATH = os.environ.get("ARK_AMD_LIB") or os.path.join(_HERE, "lib", "libark_amd.so")

PREC_F32, PREC_BF16, PREC_F16 = 0, 1, 2
LAY_KMAJ, LAY_MMAJ = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_MUL_DGELU, EPI_MUL_AUX, EPI_BIAS_RELU, EPI_MUL_RELU, EPI_ADD = 0, 1, 2, 3, 4, 5, 6, 7

_lib = None


class ArkError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ArkError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). No CPU fallback exists."
            )
        # torch ships its own copy of the HIP runtime; load it FIRST so that this library's libamdhip64
        # dependency binds to the same runtime instance (loading the system copy first leaves two runtimes in
        # the process and every launch from here fails with hipErrorNoDevice)
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def check(rc, what):
    if rc != 0:
        kind = "hipError_t" if rc > 0 else "argument error"
        raise ArkError(f"{what} failed: {kind} {rc}")


def ptr(t):
    """device pointer of a torch tensor (or None) as c_void_p"""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def i64(x):
    return ctypes.c_int64(int(x))


def i32(x):
    return ctypes.c_int(int(x))


def f32(x):
    return ctypes.c_float(float(x))


def u64(x):
    return ctypes.c_uint64(int(x) & 0xFFFFFFFFFFFFFFFF)


DIAG_MAX_ROLES = 4
WGRAD_MAX_GROUP = 12
ADAM_MAX_JOBS = 48


class GruDiagRole(ctypes.Structure):
    """ArkGruDiagRole of include/ark_amd.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in (
        "x16", "h_prev16", "w_ih16", "w_hh16", "b_ih", "b_hh", "y_prev_t", "y_out_t", "y16a", "y16b", "yd16a", "yd16b",
        "save_r", "save_z", "save_n", "save_hn")] + [
        ("drop_seed", ctypes.c_uint64), ("drop_base", ctypes.c_int64), ("drop_p", ctypes.c_float), ("pad_", ctypes.c_int),
        ("x_tab", ctypes.c_void_p), ("x_tok", ctypes.c_void_p)]


def dptr(t):
    """raw device address (0 for None) for struct fields"""
    return 0 if t is None else t.data_ptr()


class GruDiagBwdRole(ctypes.Structure):
    """ArkGruDiagBwdRole of include/ark_amd.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in (
        "dgi_up16", "w_ihT_up16", "dg_next16", "w_hhT16", "dy_t", "carry_t", "save_r", "save_z", "save_n", "save_hn",
        "y_prev_t", "dg16", "db_ih", "db_hh", "dh0")] + [
        ("drop_seed", ctypes.c_uint64), ("drop_base", ctypes.c_int64), ("drop_p", ctypes.c_float), ("first", ctypes.c_int)]


SWEEP_MAX_LAYERS = 4


class GruSweepLayer(ctypes.Structure):
    """ArkGruSweepLayer of include/ark_amd.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in (
        "w_ih16", "w_hh16", "b_ih", "b_hh", "y_t", "y16a", "y16b", "yd16a", "yd16b", "save_r", "save_z", "save_n", "save_hn")] + [
        ("drop_seed", ctypes.c_uint64), ("drop_p", ctypes.c_float), ("pad_", ctypes.c_int)]


class GruSweep(ctypes.Structure):
    """ArkGruSweep of include/ark_amd.h"""
    _fields_ = [("layer", GruSweepLayer * SWEEP_MAX_LAYERS), ("x0_16", ctypes.c_void_p), ("exch", ctypes.c_void_p),
                ("sync", ctypes.c_void_p), ("hyper", ctypes.c_void_p), ("n_layers", ctypes.c_int), ("B", ctypes.c_int),
                ("D", ctypes.c_int), ("L", ctypes.c_int), ("t0", ctypes.c_int), ("wg_slices", ctypes.c_int)]


class GruSweepBwdLayer(ctypes.Structure):
    """ArkGruSweepBwdLayer of include/ark_amd.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in (
        "w_hhT16", "w_ihT_up16", "save_r", "save_z", "save_n", "save_hn", "y_t", "dg16", "db_ih", "db_hh")] + [
        ("drop_seed", ctypes.c_uint64), ("drop_p", ctypes.c_float), ("pad_", ctypes.c_int)]


class GruSweepBwd(ctypes.Structure):
    """ArkGruSweepBwd of include/ark_amd.h"""
    _fields_ = [("layer", GruSweepBwdLayer * SWEEP_MAX_LAYERS), ("dy_t", ctypes.c_void_p), ("dh0", ctypes.c_void_p),
                ("exch", ctypes.c_void_p), ("sync", ctypes.c_void_p), ("hyper", ctypes.c_void_p), ("n_layers", ctypes.c_int),
                ("B", ctypes.c_int), ("D", ctypes.c_int), ("L", ctypes.c_int), ("wg_slices", ctypes.c_int), ("pad_", ctypes.c_int)]


class DiagTuning(ctypes.Structure):
    """ArkDiagTuning of include/ark_amd.h (speed-only tile / ring choices, passed per call)"""
    _fields_ = [(k, ctypes.c_int) for k in ("fwd_rows", "fwd_ki", "fwd_nbuf", "fwd_xcd", "fwd_units", "bwd_rows", "bwd_ki",
                                            "bwd_nbuf", "bwd_xcd_rows", "bwd_cols")]


class WgradTuning(ctypes.Structure):
    """ArkWgradTuning of include/ark_amd.h"""
    _fields_ = [(k, ctypes.c_int) for k in ("tile", "nbuf", "target_wgs", "balance", "waves")]


def sweep_sync_words(n_layers, B, Lq):
    """words of the `sync` workspace of ark_gru_sweep_fwd / _bwd"""
    f = lib().ark_gru_sweep_sync_words
    f.restype = ctypes.c_long
    return int(f(i32(n_layers), i32(B), i32(Lq)))


def wgrad_tuning(**kw):
    """the library's measured defaults with the given fields replaced (a ctypes byref, ready to pass)"""
    t = WgradTuning()
    lib().ark_wgrad_tuning_default(ctypes.byref(t))
    for k, v in kw.items():
        if k not in dict(WgradTuning._fields_):
            raise KeyError(k)
        setattr(t, k, int(v))
    return ctypes.byref(t)


def diag_tuning(**kw):
    """the library's measured defaults with the given fields replaced"""
    t = DiagTuning()
    lib().ark_diag_tuning_default(ctypes.byref(t))
    for k, v in kw.items():
        if k not in dict(DiagTuning._fields_):
            raise KeyError(k)
        setattr(t, k, int(v))
    return t

"""ctypes binding of libequiformer_hip.so (the C ABI declared in include/equiformer_hip.h).

The HIP library is THE compute path: there is no CPU or eager-PyTorch fallback.  If the shared object is missing
or a symbol cannot be resolved this module raises at import / first use, and every entry point raises on a
non-zero return code.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libequiformer_hip.so")
if os.environ.get("EQF_LIB_VARIANT"):  # development A/B builds of the same sources (equiformer_amd/build.py --variant NAME ...)
    LIB_PATH = os.path.join(_HERE, "libequiformer_hip_%s.so" % os.environ["EQF_LIB_VARIANT"])

EQF_MAX_SEG = 8
EQF_MAX_PATHS = 72

c_fp = ctypes.c_void_p  # device pointers travel as opaque addresses
c_int = ctypes.c_int


class EqfIrreps(ctypes.Structure):
    _fields_ = [("nseg", c_int), ("l", c_int * EQF_MAX_SEG), ("mul", c_int * EQF_MAX_SEG), ("odd", c_int * EQF_MAX_SEG)]


class EqfRows(ctypes.Structure):
    _fields_ = [("d", c_int), ("ld", c_int), ("inner", c_int)]


class EqfGemmDesc(ctypes.Structure):
    _fields_ = [("A", c_fp), ("B", c_fp), ("C", c_fp), ("bias", c_fp), ("ra", EqfRows), ("rc", EqfRows),
                ("ldb", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("accumulate", c_int), ("kind", c_int)]


class EqfGateIn(ctypes.Structure):
    _fields_ = [("S", c_int), ("G", c_int), ("c_silu", ctypes.c_float), ("c_sig", ctypes.c_float)]


_PA = c_int * EQF_MAX_PATHS


class EqfDtpPaths(ctypes.Structure):
    _fields_ = [("npaths", c_int), ("sh_dim", c_int), ("in_dim", c_int), ("out_dim", c_int), ("w_numel", c_int),
                ("m_numel", c_int), ("l1", _PA), ("l2", _PA), ("l3", _PA), ("mul", _PA), ("in_off", _PA),
                ("out_off", _PA), ("out_ch", _PA), ("out_k", _PA), ("w_off", _PA), ("cg_off", _PA), ("m_off", _PA)]


_P_IRR = ctypes.POINTER(EqfIrreps)
_P_PATHS = ctypes.POINTER(EqfDtpPaths)
_PP = ctypes.POINTER(ctypes.c_void_p)
_f = ctypes.c_float
_u64 = ctypes.c_ulonglong
_long = ctypes.c_long

# name -> argtypes, in the order of include/equiformer_hip.h
SIGNATURES = {
    "eqf_radius_graph_count": [c_fp, c_fp, c_int, _f, c_int, c_fp, c_fp],
    "eqf_radius_graph_fill": [c_fp, c_fp, c_int, _f, c_int, c_fp, c_fp, c_fp, c_fp],
    "eqf_radius_graph_pbc_count": [c_fp, c_fp, c_fp, c_int, _f, c_int, c_fp, c_fp, c_fp],
    "eqf_radius_graph_pbc_fill": [c_fp, c_fp, c_fp, c_int, _f, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "eqf_sumsq": [c_fp, ctypes.c_long, c_fp, c_fp],
    "eqf_adamw_step": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_long, _f, _f, _f, _f, c_int, _f, _f, c_fp],
    "eqf_adamw_step_dev": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_long, c_fp, _f, _f, _f, _f, _f, c_fp],
    "eqf_segment_ptr": [c_fp, c_int, c_int, c_fp, c_fp, c_fp],
    "eqf_exclusive_scan_i32": [c_fp, c_int, c_fp, c_fp, c_fp],
    "eqf_csr_by_source": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp],
    "eqf_edge_geom_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "eqf_edge_geom_bwd": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "eqf_rbf_gaussian_fwd": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, _f, c_fp, c_fp],
    "eqf_rbf_gaussian_bwd": [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, _f, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "eqf_rbf_expnorm_fwd": [c_fp, c_int, c_int, c_fp, c_fp, _f, _f, c_fp, c_fp],
    "eqf_rbf_expnorm_bwd": [c_fp, c_fp, c_int, c_int, c_fp, c_fp, _f, _f, c_fp, c_fp],
    "eqf_rbf_bessel_fwd": [c_fp, c_int, c_int, c_fp, _f, c_fp, c_fp],
    "eqf_rbf_bessel_bwd": [c_fp, c_fp, c_int, c_int, c_fp, _f, c_fp, c_fp, c_fp],
    "eqf_gemm_nn": [c_fp, EqfRows, c_fp, c_int, c_fp, EqfRows, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "eqf_gemm_nt": [c_fp, EqfRows, c_fp, c_int, c_fp, EqfRows, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "eqf_gemm_tn": [c_fp, EqfRows, c_fp, EqfRows, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "eqf_gemm_tn_colsum": [c_fp, EqfRows, c_fp, EqfRows, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "eqf_gemm_group": [ctypes.POINTER(EqfGemmDesc), c_int, c_fp],
    "eqf_gemmx_group": [ctypes.POINTER(EqfGemmDesc), c_int, c_int, c_fp],
    "eqf_colsum": [c_fp, EqfRows, c_int, c_int, c_fp, c_fp],
    "eqf_dtp_coupling_fwd": [c_fp, c_fp, _P_PATHS, c_fp, c_int, c_fp],
    "eqf_dtp_coupling_bwd": [c_fp, c_fp, _P_PATHS, c_fp, c_int, c_fp],
    "eqf_dtp_linear_fwd": [c_fp, c_fp, c_fp, _P_PATHS, _PP, c_fp, c_fp, _P_IRR, c_int, c_fp],
    "eqf_dtp_linear_wgrad": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, _P_IRR, _PP, c_int, c_fp],
    "eqf_sfc_fwd": [c_fp, c_fp, c_fp, _P_PATHS, _PP, c_fp, c_fp, c_fp, c_fp, _P_IRR, c_fp, c_int, c_int, c_fp],
    "eqf_sfc_bwd_data": [c_fp, c_fp, c_fp, _P_PATHS, _PP, c_fp, c_fp, _P_IRR, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_fp],
    "eqf_sfc_bwd_weight": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, _P_IRR, c_fp, c_int, _PP, c_fp, c_int, c_fp],
    "eqf_sfcx_packed_numel": [_P_PATHS, _P_IRR, c_int, c_int],
    "eqf_sfcx_supported": [_P_PATHS, _P_IRR, c_int, c_int],
    "eqf_sfcx_fwd_gated": [c_fp, ctypes.POINTER(EqfGateIn), c_fp, c_fp, _P_PATHS, c_fp, c_fp, c_fp, _P_IRR, c_int, c_int, c_fp],
    "eqf_sfcx_bwd_data_gated": [c_fp, ctypes.POINTER(EqfGateIn), c_fp, c_fp, _P_PATHS, c_fp, c_fp, _P_IRR, c_fp, c_fp, c_fp,
                                c_int, c_int, c_fp],
    "eqf_sfcx_bwd_weight_gated": [c_fp, ctypes.POINTER(EqfGateIn), c_fp, c_fp, _P_PATHS, c_fp, _P_IRR, _PP, c_int, c_int,
                                  c_fp],
    "eqf_sfcx_bwd_weight_bias": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, _P_IRR, c_fp, c_int, _PP, c_fp, c_fp, c_fp, c_int, c_int, c_fp],
    "eqf_sfcx_bwd_weight_gated_bias": [c_fp, ctypes.POINTER(EqfGateIn), c_fp, c_fp, _P_PATHS, c_fp, _P_IRR, _PP, c_fp, c_int,
                                       c_int, c_fp],
    "eqf_sfcx_pack": [_PP, c_fp, _P_PATHS, _P_IRR, c_int, c_int, c_fp, c_fp],
    "eqf_sfcx_fwd": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, c_fp, c_fp, c_fp, _P_IRR, c_fp, c_int, c_int, c_int, c_fp],
    "eqf_sfcx_bwd_data": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, c_fp, _P_IRR, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_fp],
    "eqf_sfcx_bwd_weight": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, _P_IRR, c_fp, c_int, _PP, c_fp, c_int, c_int, c_fp],
    "eqf_layernorm_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, _P_IRR, _f, c_fp],
    "eqf_layernorm_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, _P_IRR, c_fp],
    "eqf_add_layernorm_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, _P_IRR, _f, c_fp],
    "eqf_add_layernorm_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, _P_IRR, c_fp],
    "eqf_gate_fwd": [c_fp, c_fp, c_int, c_int, _P_IRR, _f, _f, c_fp],
    "eqf_gate_bwd": [c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _f, c_fp],
    "eqf_silu_fwd": [c_fp, c_fp, _long, _f, c_fp],
    "eqf_silu_bwd": [c_fp, c_fp, c_fp, _long, _f, c_fp],
    "eqf_lnsilu_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, _f, c_fp],
    "eqf_lnsilu_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _f, c_fp],
    "eqf_lnsilu_group_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, _f, c_fp],
    "eqf_lnsilu_group_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, _f, c_fp],
    "eqf_fold_weight_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp],
    "eqf_fold_weight_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp],
    "eqf_embed_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "eqf_embed_bwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "eqf_gather_add_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp],
    "eqf_segment_sum": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, _f, c_int, c_fp],
    "eqf_segment_bcast": [c_fp, c_fp, c_fp, c_int, c_int, _f, c_fp],
    "eqf_kv_split": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "eqf_kv_merge": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "eqf_dp_logits_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "eqf_dp_logits_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "eqf_vec_sh": [c_fp, c_fp, c_int, c_int, _f, c_fp, c_fp],
    "eqf_segment_scale": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp],
    "eqf_dtp_fwd": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, c_int, c_fp],
    "eqf_dtp_bwd": [c_fp, c_fp, c_fp, _P_PATHS, c_fp, c_fp, c_fp, c_fp, c_int, c_fp],
    "eqf_alpha_fwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, _f, c_fp],
    "eqf_alpha_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, _f, c_fp],
    "eqf_attn_aggregate_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _u64, c_fp],
    "eqf_attn_aggregate_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _u64, c_fp],
    "eqf_attn_aggregate_fwd_dseed": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _u64, c_fp, c_fp],
    "eqf_attn_aggregate_bwd_dseed": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _u64, c_fp, c_fp],
    "eqf_silu_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, _long, _f, c_fp],
    "eqf_gate_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _f, c_fp],
    "eqf_lnsilu_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _f, c_fp],
    "eqf_lnsilu_group_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, _f, c_fp],
    "eqf_layernorm_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, _P_IRR, _f, c_fp],
    "eqf_alpha_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, _f, c_fp],
    "eqf_attn_aggregate_bwd2": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, _P_IRR, _f, _u64,
                                c_fp],
    "eqf_rbf_expnorm_bwd2": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, _f, _f, c_fp, c_fp, c_fp],
    "eqf_rbf_gaussian_bwd2": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, _f, c_fp, c_fp, c_fp, c_fp, c_fp,
                              c_fp, c_fp],
    "eqf_rbf_bessel_bwd2": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, _f, c_fp, c_fp, c_fp, c_fp],
    "eqf_edge_geom_bwd2": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "eqf_prof_enable": [ctypes.c_char_p],
    "eqf_prof_report": [ctypes.c_char_p, c_int],
}

RESTYPES = {"eqf_sfcx_packed_numel": ctypes.c_long}  # everything else returns an int status

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and attach prototypes.  Raises HipLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "libequiformer_hip.so not found at %s -- build it with `python -m equiformer_amd.build` "
            "(there is no CPU fallback for the hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.eqf_version.restype = ctypes.c_char_p
    lib.eqf_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> missing symbol, loud
        fn.restype = RESTYPES.get(name, c_int)
        fn.argtypes = argtypes
    _check_build(lib)
    _lib = lib
    return lib


def _check_build(lib):
    """The library reports the hash of the sources it was compiled from (eqf_version: "... src=<hash>").  When the sources
    sit beside it (a checkout, a gpurun snapshot) and their hash differs, the binary is stale: measurements and parity
    claims would belong to other code.  EQF_ALLOW_STALE_LIB=1 turns the error into a warning (development builds)."""
    csrc = os.path.join(_HERE, "csrc")
    if not os.path.isdir(csrc):
        return
    ver = lib.eqf_version().decode()
    built = ver.split("src=")[-1] if "src=" in ver else "unknown"
    from . import build as _build
    want = _build.source_hash()
    if built != want:
        msg = ("libequiformer_hip.so was built from sources with hash %s, the sources beside it hash to %s -- rebuild with "
               "`python -m equiformer_amd.build`" % (built, want))
        if os.environ.get("EQF_ALLOW_STALE_LIB") == "1":
            import warnings
            warnings.warn(msg)
        else:
            raise HipLibraryError(msg)


def built_hash():
    """Hash of the sources the LOADED binary was compiled from."""
    ver = version()
    return ver.split("src=")[-1] if "src=" in ver else "unknown"


def version():
    return load().eqf_version().decode()


_fn_cache = {}


def call(name, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise HipLibraryError("%s failed with code %d" % (name, rc))


def prof_enable(filter_substring):
    """None disables; "" records every matrix-core launch; otherwise only kernels whose name contains the string."""
    load().eqf_prof_enable(None if filter_substring is None else filter_substring.encode())


def prof_report():
    """{kernel name: dict(launches, total_ms, flops, bytes)} for the launches recorded since the last call."""
    buf = ctypes.create_string_buffer(1 << 16)
    n = load().eqf_prof_report(buf, len(buf))
    out = {}
    if n > 0:
        for line in buf.value.decode().splitlines():
            name, cnt, ms, fl, by = line.split()
            out[name] = dict(launches=int(float(cnt)), total_ms=float(ms), flops=float(fl), bytes=float(by))
    return out


def make_irreps(segments, parities=None):
    """segments: iterable of (mul, l); parities: +1 / -1 per segment (default: all even)."""
    segments = list(segments)
    if len(segments) > EQF_MAX_SEG:
        raise ValueError("too many irreps segments")
    s = EqfIrreps()
    s.nseg = len(segments)
    for i, (mul, l) in enumerate(segments):
        s.l[i], s.mul[i] = int(l), int(mul)
        s.odd[i] = 1 if parities is not None and parities[i] == -1 else 0
    return s

"""Python side of the stand-in R runtime (rstub.c): builds R-like objects, calls the shim's `.Call` entry
points by name through the registered table, converts results back to numpy.  Test infrastructure."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_rstub  # noqa: E402

INTSXP, REALSXP, VECSXP, LGLSXP, NILSXP, EXTPTRSXP, STRSXP = 13, 14, 19, 10, 0, 22, 16


class RError(RuntimeError):
    pass


class R:
    def __init__(self):
        # the shim resolves libbigsnpr_hip's symbols: load that first, globally
        import bigsnpr_amd
        from bigsnpr_amd import _lib
        C.CDLL(os.environ.get("BSN_LIB_PATH") or _lib.LIB_PATH, mode=C.RTLD_GLOBAL)
        L = self.L = C.CDLL(build_rstub.build())
        vp = C.c_void_p
        for name, res, args in [
                ("rstub_nil", vp, []), ("rstub_int", vp, [C.POINTER(C.c_int), C.c_ssize_t]), ("rstub_lgl", vp, [C.c_int]),
                ("rstub_real", vp, [C.POINTER(C.c_double), C.c_ssize_t]),
                ("rstub_real_matrix", vp, [C.POINTER(C.c_double), C.c_int, C.c_int]), ("rstub_string", vp, [C.c_char_p]),
                ("rstub_raw_matrix", vp, [C.POINTER(C.c_ubyte), C.c_int, C.c_int]),
                ("rstub_env", vp, []), ("rstub_env_set", None, [vp, C.c_char_p, vp]), ("rstub_type", C.c_int, [vp]),
                ("rstub_length", C.c_ssize_t, [vp]), ("rstub_nrow", C.c_int, [vp]), ("rstub_ncol", C.c_int, [vp]),
                ("rstub_data", vp, [vp]), ("rstub_list_get", vp, [vp, C.c_ssize_t]),
                ("rstub_list_name", C.c_char_p, [vp, C.c_ssize_t]), ("rstub_last_error", C.c_char_p, []),
                ("rstub_warnings", C.c_char_p, []), ("rstub_call", vp, [C.c_char_p, C.c_int, C.POINTER(vp)]),
                ("rstub_reset", None, []), ("rstub_n_routines", C.c_int, []), ("rstub_routine_name", C.c_char_p, [C.c_int]),
                ("rstub_routine_nargs", C.c_int, [C.c_int]), ("R_init_bigsnprhip", None, [vp])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.R_init_bigsnprhip(None)
        self.warnings = ""

    def routines(self):
        return {self.L.rstub_routine_name(i).decode(): self.L.rstub_routine_nargs(i) for i in range(self.L.rstub_n_routines())}

    # ---- R objects from Python values ----
    def obj(self, v):
        L = self.L
        if v is None:
            return L.rstub_nil()
        if isinstance(v, (int, C.c_void_p)) and not isinstance(v, bool) and getattr(v, "_is_sexp", False):
            return v
        if isinstance(v, SEXP):
            return v.p
        if isinstance(v, bool):
            return L.rstub_lgl(int(v))
        if isinstance(v, str):
            return L.rstub_string(v.encode())
        if isinstance(v, int):
            a = np.array([v], dtype=np.int32)
            return L.rstub_int(a.ctypes.data_as(C.POINTER(C.c_int)), 1)
        if isinstance(v, float):
            a = np.array([v], dtype=np.float64)
            return L.rstub_real(a.ctypes.data_as(C.POINTER(C.c_double)), 1)
        a = np.asarray(v)
        if a.dtype == np.uint8 and a.ndim == 2:      # a raw matrix
            f = np.asfortranarray(a)
            return L.rstub_raw_matrix(f.ctypes.data_as(C.POINTER(C.c_ubyte)), a.shape[0], a.shape[1])
        if a.dtype.kind in "iu":
            a = np.ascontiguousarray(a, dtype=np.int32)
            return L.rstub_int(a.ctypes.data_as(C.POINTER(C.c_int)), a.size)
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 2:
            f = np.asfortranarray(a)
            return L.rstub_real_matrix(f.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], a.shape[1])
        a = np.ascontiguousarray(a)
        return L.rstub_real(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)

    def env(self, **fields):
        e = self.L.rstub_env()
        for k, v in fields.items():
            self.L.rstub_env_set(e, k.encode(), self.obj(v))
        return SEXP(e)

    # ---- .Call ----
    def call(self, name, *args):
        arr = (C.c_void_p * max(1, len(args)))(*[self.obj(a) for a in args])
        res = self.L.rstub_call(name.encode(), len(args), arr)
        self.warnings = self.L.rstub_warnings().decode()
        if not res:
            raise RError(self.L.rstub_last_error().decode())
        return self.value(res)

    def value(self, p):
        L = self.L
        t = L.rstub_type(p)
        n = L.rstub_length(p)
        if t == NILSXP:
            return None
        if t in (INTSXP, LGLSXP, REALSXP):
            ct, dt = (C.c_int, np.int32) if t != REALSXP else (C.c_double, np.float64)
            a = np.ctypeslib.as_array(C.cast(L.rstub_data(p), C.POINTER(ct)), shape=(n,)).astype(dt).copy() if n else np.empty(0, dt)
            if L.rstub_nrow(p) > 0:
                a = a.reshape((L.rstub_nrow(p), L.rstub_ncol(p)), order="F")
            return a
        if t == VECSXP:
            items = [self.value(L.rstub_list_get(p, i)) for i in range(n)]
            names = [L.rstub_list_name(p, i).decode() for i in range(n)]
            return dict(zip(names, items)) if any(names) else items
        return SEXP(p)   # external pointers and other opaque things

    def reset(self):
        self.L.rstub_reset()


class SEXP:
    def __init__(self, p):
        self.p = p

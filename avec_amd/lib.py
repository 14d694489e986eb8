"""ctypes binding of libavec_hip.so (the C-ABI drop-in boundary, include/avec_hip.h).

The prototypes are parsed from the header itself so the Python side cannot drift from the ABI.
There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "avec_hip.h")
LIB_PATH = os.environ.get("AVEC_LIB_PATH") or os.path.join(_HERE, "libavec_hip.so")     # AVEC_LIB_PATH: kernel experiments (tools/build_abl.sh)

F32, BF16 = 0, 1
ROWS_PLAIN, ROWS_CONV_FWD, ROWS_CONV_BWD, ROWS_STEM3D = 0, 1, 2, 3
ACT_NONE, ACT_SWISH, ACT_RELU = 0, 1, 2


class Rows(ctypes.Structure):
    _fields_ = [("ld", ctypes.c_longlong), ("rows_out", ctypes.c_int), ("rows_in", ctypes.c_int), ("step", ctypes.c_int),
                ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int), ("KH", ctypes.c_int), ("KW", ctypes.c_int),
                ("stride", ctypes.c_int), ("pad", ctypes.c_int), ("OH", ctypes.c_int), ("OW", ctypes.c_int), ("T3", ctypes.c_int)]


class Epilogue(ctypes.Structure):
    _fields_ = [("out", ctypes.c_void_p), ("ldo", ctypes.c_longlong), ("out_f32", ctypes.c_int),
                ("out_pre", ctypes.c_void_p), ("ldpre", ctypes.c_longlong),
                ("bias", ctypes.c_void_p), ("act", ctypes.c_int),
                ("drop_p", ctypes.c_float), ("rng", ctypes.c_void_p), ("rng_stream", ctypes.c_uint),
                ("res", ctypes.c_void_p), ("ldres", ctypes.c_longlong), ("alpha", ctypes.c_float), ("res_act", ctypes.c_int),
                ("dact_z", ctypes.c_void_p), ("ldz", ctypes.c_longlong), ("dact", ctypes.c_int),
                ("colsum", ctypes.c_void_p), ("stats", ctypes.c_void_p),
                ("bnb_y", ctypes.c_void_p), ("ldby", ctypes.c_longlong), ("bnb_ss", ctypes.c_void_p), ("bnb_mask", ctypes.c_int),
                ("res_cls0", ctypes.c_int), ("res_mask", ctypes.c_void_p)]


class Attn(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p), ("ld", ctypes.c_longlong),
                ("e", ctypes.c_void_p), ("lde", ctypes.c_longlong),
                ("lens", ctypes.c_void_p), ("len_div", ctypes.c_int), ("q_full", ctypes.c_int),
                ("mask", ctypes.c_void_p), ("mask_bstride", ctypes.c_longlong),
                ("o", ctypes.c_void_p), ("ldo", ctypes.c_longlong), ("lse", ctypes.c_void_p),
                ("dout", ctypes.c_void_p),
                ("dq", ctypes.c_void_p), ("dk", ctypes.c_void_p), ("dv", ctypes.c_void_p), ("lddq", ctypes.c_longlong), ("ldd", ctypes.c_longlong),
                ("de", ctypes.c_void_p), ("ldde", ctypes.c_longlong), ("pbuf", ctypes.c_void_p), ("dsbuf", ctypes.c_void_p), ("ldt", ctypes.c_longlong), ("dsrel", ctypes.c_void_p), ("ldr", ctypes.c_longlong),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("T", ctypes.c_int), ("d", ctypes.c_int), ("scale", ctypes.c_float), ("Tk", ctypes.c_int)]


class TnItem(ctypes.Structure):
    """avec_tn_item_t"""
    _fields_ = [("P", ctypes.c_void_p), ("Q", ctypes.c_void_p), ("O", ctypes.c_void_p), ("p_colsum", ctypes.c_void_p),
                ("ldp", ctypes.c_longlong), ("ldq", ctypes.c_longlong), ("ldo", ctypes.c_longlong), ("M", ctypes.c_longlong),
                ("I", ctypes.c_int), ("J", ctypes.c_int), ("q_rows_out", ctypes.c_int), ("q_rows_in", ctypes.c_int), ("q_step", ctypes.c_int),
                ("reserved", ctypes.c_int)]


class TnBatched(ctypes.Structure):
    """avec_tn_batched_t"""
    _fields_ = [("P", ctypes.c_void_p), ("ldp", ctypes.c_longlong), ("Q", ctypes.c_void_p), ("ldq", ctypes.c_longlong),
                ("O", ctypes.c_void_p), ("O_act", ctypes.c_void_p), ("ldo", ctypes.c_longlong),
                ("M", ctypes.c_longlong), ("I", ctypes.c_int), ("J", ctypes.c_int), ("nb_outer", ctypes.c_int), ("nb_inner", ctypes.c_int),
                ("strides6", ctypes.POINTER(ctypes.c_longlong))]


class LnItem(ctypes.Structure):
    """avec_ln_item_t"""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p), ("dgamma", ctypes.c_void_p),
                ("dbeta", ctypes.c_void_p), ("M", ctypes.c_longlong), ("D", ctypes.c_int), ("dy_f32", ctypes.c_int)]


class Fp8Item(ctypes.Structure):
    """avec_fp8_item_t"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("n", ctypes.c_longlong), ("slot", ctypes.c_int), ("K", ctypes.c_int), ("Kp", ctypes.c_int),
                ("reserved", ctypes.c_int)]


class WgradItem(ctypes.Structure):
    """avec_wgrad3x3_item_t"""
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("images", ctypes.c_longlong), ("C", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("reserved", ctypes.c_int)]


TN_GROUP_MAX, LN_GROUP_MAX, WGRAD_GROUP_MAX = 32, 40, 16
ABI_STRUCTS = (Rows, Epilogue, Attn, TnItem, TnBatched, LnItem, Fp8Item, WgradItem)      # index = `which` of avec_struct_size


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl or decl.startswith("hipStream_t"):
        return ctypes.c_void_p
    base = re.sub(r"\b[A-Za-z_][A-Za-z_0-9]*$", "", decl).strip() or decl  # drop the parameter name
    base = base.replace("const", "").strip()
    return {"int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float, "unsigned": ctypes.c_uint,
            "void": None}[base]


def declared_functions(header=HEADER):
    """{name: (restype, [argtypes])} for every `avec_*` function declared in the header."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*(const char\*|long long|int)\s+(avec_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        restype = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "const char*": ctypes.c_char_p}[ret]
        out[name] = (restype, argtypes)
    return out


class _Lib:
    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError("libavec_hip.so not found at %s -- build it with `python -m avec_amd.build` "
                                   "(there is no non-HIP fallback)" % LIB_PATH)
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in declared_functions().items():
                fn = getattr(dll, name)  # AttributeError if the .so does not export a declared symbol
                fn.restype = restype
                fn.argtypes = argtypes
            want = int(re.search(r"#define AVEC_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
            if dll.avec_version() != want:
                raise RuntimeError("libavec_hip.so ABI version %d, include/avec_hip.h declares %d: rebuild (python -m avec_amd.build)" % (dll.avec_version(), want))
            for which, st in enumerate(ABI_STRUCTS):          # a struct that grew on one side only would be read past its end
                if dll.avec_struct_size(which) != ctypes.sizeof(st):
                    raise RuntimeError("libavec_hip.so: sizeof(%s) is %d in the library, %d in avec_amd/lib.py" % (st.__name__, dll.avec_struct_size(which), ctypes.sizeof(st)))
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        fn = getattr(self.load(), "avec_" + name)

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise RuntimeError("avec_%s failed (%d): %s" % (name, rc, self._dll.avec_last_error().decode()))
        call.__name__ = name
        self.__dict__[name] = call
        return call

    def raw(self, name):
        return getattr(self.load(), name)


lib = _Lib()

"""ctypes binding of liblpg_hip.so, generated from include/lpg_hip.h.

The prototypes are parsed out of the header so the header stays the single source of truth for
the C ABI (tests/test_abi.py checks that every declared symbol is exported).  There is no CPU
fallback: if the shared library is missing the import of any op raises.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(ROOT, "include", "lpg_hip.h")
# LPG_LIB_VARIANT=tuning: the -DLPG_TUNING build (csrc/build.py --tuning) for the sweep scripts under profiles/
LIB_PATH = os.path.join(_PKG, "csrc", "liblpg_hip_tuning.so" if os.environ.get("LPG_LIB_VARIANT") == "tuning"
                        else "liblpg_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
    "float": ctypes.c_float, "uint8_t": ctypes.c_uint8, "uint64_t": ctypes.c_uint64, "uint16_t": ctypes.c_uint16,
}
_PROTO = re.compile(r"^\s*(const\s+char\s*\*|int64_t|int)\s+(lpg_\w+)\s*\(([^)]*)\)\s*;", re.M | re.S)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every `lpg_*` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for ret, name, args in _PROTO.findall(text):
        ret = ret.replace(" ", "")
        restype = ctypes.c_char_p if ret.startswith("constchar") else _CTYPES[ret]
        argtypes, argnames = [], []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                nm = re.findall(r"(\w+)$", a)[0]
                ty = a[: -len(nm)].strip()
                if "*" in ty:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[ty.replace("const ", "").strip()])
                argnames.append(nm)
        out[name] = (restype, argtypes, argnames)
    return out


class LpgError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LpgError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (or python link-prediction-gnn_amd/csrc/build.py). "
            "This package has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes, _) in parse_header().items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lpg_abi_version() != 1:
        raise LpgError("liblpg_hip.so ABI version mismatch")
    _lib = lib
    return lib


def set_option(name, value):
    """process-wide kernel-selection option (include/lpg_hip.h: lpg_set_option)"""
    check(load().lpg_set_option(name.encode(), int(value)), f"set_option({name})")


def get_option(name):
    return int(load().lpg_get_option(name.encode()))


class options:
    """with _lib.options(spmm_chunk=64, linear_wres=2): ...  -- sets the options, restores the old values on exit"""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


# ---- recording of launch sequences (lpg_twowl_forward_small's argument block) --------------------------------------
_REC_OPS = {"lpg_embedding_graphnorm_f32": (1, 12, (8,)), "lpg_small_linear_t_f32": (2, 27, (25,)),
            "lpg_small_gcn_f32": (3, 20, ()), "lpg_small_readout_f32": (4, 23, (16,)),
            "lpg_small_gcn_embed_f32": (5, 21, (12,)), "lpg_tuple_layer_f32": (6, 19, ()),
            "lpg_tuple_readout_f32": (7, 31, (16, 21)), "lpg_tuple_ax_small_f32": (8, 15, (12,))}
_rec = None            # while recording: {"words": [...], "keep": [...]}


def call(name, *args):
    """lib.<name>(*args) with the return code checked; while a recording is open the launch is also appended to it as
    { op, a_0 .. a_{k-1} } words (stream left out, floats as the bits of a double): include/lpg_hip.h,
    lpg_twowl_forward_small."""
    rc = check(getattr(load(), name)(*args), name)
    if _rec is not None:
        import struct
        op, k, floats = _REC_OPS[name]
        if len(args) != k + 1:
            raise LpgError(f"{name}: recorded with {len(args)} arguments, the block layout has {k} + stream")
        words = [op]
        for i, a in enumerate(args[:k]):
            if i in floats:
                words.append(struct.unpack("<q", struct.pack("<d", float(a)))[0])
            else:
                words.append(0 if a is None else int(a))
        _rec["words"].extend(words)
    return rc


def keepalive(*tensors):
    """tensors whose device pointers went into a recorded launch must outlive the recording"""
    if _rec is not None:
        _rec["keep"].extend(t for t in tensors if t is not None)


class recording:
    """with _lib.recording() as rec: ...   ->  rec["words"] (the block), rec["keep"] (tensors to hold on to)"""

    def __enter__(self):
        global _rec
        if _rec is not None:
            raise LpgError("a launch recording is already open")
        _rec = {"words": [], "keep": []}
        return _rec

    def __exit__(self, *exc):
        global _rec
        _rec = None
        return False


def check(rc, what=""):
    if rc < 0:
        msg = load().lpg_last_error()
        raise LpgError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
    return rc

"""ctypes binding of libartdeco_hip.so.

The prototypes are derived from include/artdeco_hip.h (single source of truth
for the C ABI), so a signature change in the header is a signature change here.
There is deliberately NO fallback: if the library is missing or a launch fails,
the caller gets an exception, never a silent CPU/eager path.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "artdeco_hip.h")
# ARTDECO_HIP_LIB: developer override used for A/B measurements of differently-compiled builds of the SAME sources
# (python -m artdeco_amd.build --variant NAME --extra-flags ...); the default is the in-tree library.
LIB_PATH = os.environ.get("ARTDECO_HIP_LIB") or os.path.join(_HERE, "lib", "libartdeco_hip.so")

_CTYPE = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "int32_t": ctypes.c_int32,
    "uint32_t": ctypes.c_uint32,
    "size_t": ctypes.c_size_t,
    "adk_stream_t": ctypes.c_void_p,
}

_ERRNAMES = {-1: "ADK_EINVAL (bad argument)", -2: "ADK_EWORKSPACE (workspace too small)",
             -3: "ADK_EUNSUPPORTED"}


class AdkError(RuntimeError):
    pass


def parse_header(path: str = HEADER) -> dict[str, list[tuple[str, str]]]:
    """Return {symbol: [(ctype_name_or_'ptr', arg_name), ...]} for every `int|int64_t adk_*(...)`.
    The return type is kept in RETURN_TYPES[symbol]."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out: dict[str, list[tuple[str, str]]] = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(adk_\w+)\s*\(([^)]*)\)\s*;", src):
        rtype, name, args = m.group(1), m.group(2), m.group(3).strip()
        RETURN_TYPES[name] = rtype
        parsed: list[tuple[str, str]] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    parsed.append(("ptr", a.split("*")[-1].strip()))
                else:
                    toks = a.replace("const ", "").split()
                    parsed.append((toks[0], toks[-1]))
        out[name] = parsed
    return out


RETURN_TYPES: dict[str, str] = {}

_lock = threading.Lock()
_lib = None
_protos = None


def is_loaded() -> bool:
    return _lib is not None


def load():
    """Load the shared library (once) and attach argtypes/restype from the header."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise AdkError(
                f"{LIB_PATH} is missing: build it with `python -m artdeco_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback for the artdeco_amd operators.")
        # torch must own the HIP runtime that is already in the process (same SONAME
        # libamdhip64.so.7), so import it first when it is available.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is a hard dependency of the wrappers anyway
            pass
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
        protos = parse_header()
        for name, args in protos.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = _CTYPE[RETURN_TYPES[name]]
            fn.argtypes = [ctypes.c_void_p if t == "ptr" else _CTYPE[t] for t, _ in args]
        if lib.adk_abi_version() != _header_abi_version():
            raise AdkError("libartdeco_hip.so ABI version does not match include/artdeco_hip.h; rebuild")
        _lib, _protos = lib, protos
    return _lib


def _header_abi_version() -> int:
    m = re.search(r"#define\s+ADK_ABI_VERSION\s+(\d+)", open(HEADER).read())
    return int(m.group(1))


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise AdkError(f"{what}: {_ERRNAMES.get(rc, rc)}")
    raise AdkError(f"{what}: HIP error {rc}")


def ptr(t) -> int | None:
    """data_ptr of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def raw_stream(device) -> int:
    """hipStream_t of torch's CURRENT stream on `device` (a torch.device).  The raw accessor torch's own compilers use: a step of
    the mapper asks for the stream ~12 times, and building a torch.cuda.Stream object each time was 60 us of host time per step."""
    import torch
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    try:
        return torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:  # pragma: no cover - older torch
        return torch.cuda.current_stream(device).cuda_stream


def stream_of(t) -> int:
    """hipStream_t of torch's current stream on t's device."""
    return raw_stream(t.device)


class on_device:
    """`with torch.cuda.device(dev)` without its cost when `dev` already is the current device (the mapper's case: ~10 entries per
    step at ~5 us each); a different device is switched to and restored exactly as torch does."""
    __slots__ = ("dev", "ctx")

    def __init__(self, dev):
        self.dev, self.ctx = dev, None

    def __enter__(self):
        import torch
        idx = self.dev.index
        if idx is not None and idx != torch.cuda.current_device():
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AdkError("artdeco_amd operators run on the MI355X only: got a CPU tensor "
                           "(there is no CPU fallback; use oracle/ for CPU reference results)")

"""ctypes binding of libtsgnn_hip.so (the C ABI declared in include/tsgnn.h).

The signatures are parsed from the header itself, so header, library and binding cannot drift.
There is NO fallback: if the library is missing or a symbol is absent this module raises — the
product path never computes on the CPU (the CPU oracle lives in oracle/ and is test-only).
"""
import ctypes
import os
import re
import subprocess

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
HEADER = os.path.join(_ROOT, "include", "tsgnn.h")
CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.path.join(_PKG_DIR, "libtsgnn_hip.so")

_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
            "unsigned": ctypes.c_uint, "unsigned long long": ctypes.c_ulonglong, "uint64_t": ctypes.c_uint64}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every `int|const char* tsgnn_*(...)`."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(tsgnn_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a or a.startswith("tsgnn_stream_t"):
                    params.append((ctypes.c_void_p, a.split()[-1].lstrip("*")))
                else:
                    ty = a.split()[:-1]
                    ty = [t for t in ty if t != "const"]
                    params.append((_SCALARS[" ".join(ty)], a.split()[-1]))
        out[name] = (ctypes.c_int if ret == "int" else ctypes.c_char_p, params)
    return out


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


HASH_PATH = LIB_PATH + ".srchash"


def source_hash():
    """sha256 over the contents of every source the library is built from (kernels, headers, the C ABI header)"""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(sources() + [HEADER] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]):
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _object_hash(src, headers):
    """what one object file depends on: its source, every header it may include, the compile flags"""
    import hashlib
    h = hashlib.sha256(" ".join(_FLAGS).encode())
    for d in [src] + headers:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> two-stage-gnn_amd/libtsgnn_hip.so (in-tree, travels with gpurun).  Nothing is trusted by
    time stamp: the library is reused only if the hash of the sources it was built from (stored beside it) equals the hash of
    the sources in the tree, and each object only if ITS hash file does (a tree restored with older mtimes than build/*.o used to
    re-link stale objects and certify them).  One builder at a time (file lock: the ranks of a torchrun launch may all find the
    library stale), and the link goes to a temporary name that is renamed into place, so a process that is loading the library
    never sees a half-written file."""
    import fcntl
    srcs = sources()
    headers = sorted([HEADER] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")])
    bdir = os.path.join(_PKG_DIR, "build")
    os.makedirs(bdir, exist_ok=True)
    with open(os.path.join(bdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        cur = source_hash()                                  # (after the lock: another process may just have built it)
        if not force and os.path.exists(LIB_PATH) and os.path.exists(HASH_PATH) and open(HASH_PATH).read().strip() == cur:
            return LIB_PATH
        objs = []
        procs = []
        for s in srcs:
            o = os.path.join(bdir, os.path.basename(s)[:-4] + ".o")
            objs.append(o)
            oh = _object_hash(s, headers)
            if not force and os.path.exists(o) and os.path.exists(o + ".hash") and open(o + ".hash").read().strip() == oh:
                continue
            if os.path.exists(o + ".hash"):
                os.remove(o + ".hash")
            cmd = ["hipcc"] + _FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, o, oh, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for s, o, oh, p in procs:
            log = p.communicate()[0].decode()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (s, log))
            with open(o + ".hash", "w") as f:
                f.write(oh + "\n")
        tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode())
        os.replace(tmp, LIB_PATH)
        with open(HASH_PATH + ".tmp", "w") as f:
            f.write(cur + "\n")
        os.replace(HASH_PATH + ".tmp", HASH_PATH)
    global _lib
    _lib = None
    return LIB_PATH


_lib = None
_decls = None


def lib():
    global _lib, _decls
    if _lib is not None:
        return _lib
    import shutil
    if shutil.which("hipcc") and (not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH)
                                  or open(HASH_PATH).read().strip() != source_hash()):
        build()                          # compiling the product is not a fallback: the HIP path is still the only path
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libtsgnn_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "from the repo root. There is no CPU fallback for the product path." % LIB_PATH)
    # TSGNN_LIB_PATH: load ANOTHER build of the same sources (an A/B variant compiled with a different -D flag); still a HIP
    # library with the same ABI: not a fallback
    L = ctypes.CDLL(os.environ.get("TSGNN_LIB_PATH") or LIB_PATH)
    _decls = parse_header()
    for name, (ret, params) in _decls.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise RuntimeError("libtsgnn_hip.so does not export %s (declared in include/tsgnn.h); rebuild" % name)
        fn.restype = ret
        fn.argtypes = [p[0] for p in params]
    if L.tsgnn_abi_version() != 1:
        raise RuntimeError("libtsgnn_hip.so ABI version mismatch")
    _lib = L
    return L


def stream_handle():
    return torch.cuda.current_stream().cuda_stream


def _arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            raise RuntimeError("tsgnn kernels take device tensors; got a CPU tensor (no CPU fallback)")
        return a.data_ptr()
    return a


trace = None        # measurement hook (bench.py): a list -> every launch appends (entry point, args, dispatched kernel name)


def last_kernel():
    """device kernel (template arguments included) dispatched by this thread's most recent entry-point call"""
    return lib().tsgnn_last_kernel().decode()


_defer = None       # a list while launches are being RECORDED instead of issued (deferred(); sage_stack.run_paired merges two records)


class deferred:
    """``with deferred() as q``: every call() / defer() inside is appended to q instead of being executed.  The recorded argument
    tensors stay referenced by q, so nothing they point to is recycled before run(q) (or a merge of several records) issues them.
    Only for code whose HOST side does not depend on device results — true of every entry point of this library."""

    def __enter__(self):
        global _defer
        self._prev, self.q = _defer, []
        _defer = self.q
        return self.q

    def __exit__(self, *exc):
        global _defer
        _defer = self._prev
        return False


def defer(fn):
    """a torch-side operation that must keep its place among recorded launches (executed at once outside deferred())"""
    if _defer is not None:
        _defer.append((None, fn))
    else:
        fn()


def defer_zero(t):
    """t.zero_() as a record of its own kind: two of them from paired records share one multi-tensor launch (sage_stack.run_paired)"""
    if _defer is not None:
        _defer.append(("_zero", (t,)))
    else:
        t.zero_()


def run(q):
    for name, args in q:
        if name is None:
            args()
        elif name == "_zero":
            args[0].zero_()
        else:
            call(name, *args)


def try_call(name, *args):
    """like call(), but an argument combination the entry point does not take (TSGNN_EUNSUPPORTED, decided before anything is
    launched) returns False instead of raising"""
    L = lib()
    rc = getattr(L, "tsgnn_" + name)(*[_arg(a) for a in args], stream_handle())
    if rc == -3:
        return False
    if rc != 0:
        raise RuntimeError("tsgnn_%s failed: %s" % (name, L.tsgnn_strerror(rc).decode()))
    if trace is not None:
        trace.append((name, args, L.tsgnn_last_kernel().decode()))
    return True


def call(name, *args):
    """Call tsgnn_<name>(*args, current_stream); tensors -> device pointers, None -> NULL."""
    if _defer is not None:
        _defer.append((name, args))
        return
    L = lib()
    fn = getattr(L, "tsgnn_" + name)
    rc = fn(*[_arg(a) for a in args], stream_handle())
    if rc != 0:
        raise RuntimeError("tsgnn_%s failed: %s" % (name, L.tsgnn_strerror(rc).decode()))
    if trace is not None:
        trace.append((name, args, L.tsgnn_last_kernel().decode()))


def call_nostream(name, *args):
    L = lib()
    rc = getattr(L, "tsgnn_" + name)(*[_arg(a) for a in args])
    if rc != 0:
        raise RuntimeError("tsgnn_%s failed: %s" % (name, L.tsgnn_strerror(rc).decode()))

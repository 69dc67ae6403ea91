"""Builds libeoe_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build() and, lazily, by
eoe_amd._lib when the library is missing or stale and hipcc is available.

Staleness is decided by CONTENT, not by mtimes: every object file and the library carry a stamp with the SHA-256 of what
they were built from (the source, every header under csrc/ and include/, the compiler flags).  The built library travels
to the GPU box while git checkouts and snapshots reset mtimes, so a timestamp comparison there would either rebuild
everything or trust a binary that no longer matches the sources.  The whole check-and-build runs under an exclusive file
lock: N ranks importing the package after a source change compile once, the others wait and find the result.
"""
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libeoe_hip.so")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["api.cpp", "gemm.hip", "gemm_tn.hip", "elementwise.hip", "attention.hip", "conv.hip", "cbam.hip", "augment.hip",
           "vit.cpp", "parity.hip", "gemm256.hip", "comm.cpp", "gemm_tn256.hip", "gemm_w8.hip", "probe.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# gemm_w8.hip names its 128 accumulator registers literally: the compiler must not park spilled VGPRs in AGPRs there (the file's header)
EXTRA_CFLAGS = {"gemm_w8.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _sha(*chunks) -> str:
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else str(c).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _headers_digest() -> str:
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc")))
    hs += sorted(os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h"))
    return _sha(*[_read(h) for h in hs])


def _flags(src: str = ""):
    # EOE_CFLAGS: extra compiler flags for an experiment (part of the content stamp, so a library built with them is rebuilt without them)
    return CFLAGS + EXTRA_CFLAGS.get(src, []) + (["-DEOE_AB"] if os.environ.get("EOE_AB") else []) + os.environ.get("EOE_CFLAGS", "").split()


def source_digest(src: str, headers: str = None) -> str:
    """what the object file of `src` depends on"""
    return _sha(_read(os.path.join(CSRC, src)), headers or _headers_digest(), " ".join(_flags(src)))


def library_digest() -> str:
    headers = _headers_digest()
    return _sha(*[source_digest(s, headers) for s in SOURCES], " ".join(LDFLAGS))


def _stamp(path: str) -> str:
    try:
        with open(path + ".stamp") as f:
            return f.read().strip()
    except OSError:
        return ""


def _write_stamp(path: str, digest: str):
    with open(path + ".stamp", "w") as f:
        f.write(digest + "\n")


def needs_build() -> bool:
    return not os.path.exists(LIB) or _stamp(LIB) != library_digest()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("libeoe_hip.so is missing or does not match the sources, and hipcc was not found to rebuild it")
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():          # another process built it while this one waited for the lock
                return LIB
            return _build_locked(hipcc, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc: str, force: bool, verbose: bool) -> str:
    headers = _headers_digest()
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        digest = source_digest(src, headers)
        if not force and os.path.exists(obj) and _stamp(obj) == digest:
            continue
        cmd = [hipcc] + _flags(src) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, obj, digest, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, obj, digest, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out}")
        else:
            _write_stamp(obj, digest)
    if failed:
        raise RuntimeError("\n".join(failed))
    # link next to the target and rename over it: a process that has the old library mapped keeps a valid image (writing
    # into the mapped file in place leaves it with a torn code object -- every launch then fails with "no ROCm-capable device")
    tmp = LIB + f".tmp{os.getpid()}"
    # --no-undefined: a kernel whose host stub the compiler dropped must fail the build here, not at dlopen on the GPU box
    cmd = [hipcc] + LDFLAGS + ["-o", tmp] + objs + _link_libs()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB)
    _write_stamp(LIB, library_digest())
    return LIB


def _link_libs():
    """comm.cpp binds RCCL at run time (dlopen of the librccl the process already has through torch), so nothing to link but
    libdl"""
    return ["-ldl"]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

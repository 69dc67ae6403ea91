"""Builds libeoe_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build() and, lazily, by
eoe_amd._lib when the library is missing and hipcc is available."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libeoe_hip.so")
SOURCES = ["api.cpp", "gemm.hip", "gemm_tn.hip", "elementwise.hip", "attention.hip", "conv.hip", "cbam.hip", "augment.hip", "vit.cpp"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "eoe_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libeoe_hip.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        extra = ["-DEOE_AB"] if os.environ.get("EOE_AB") else []
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + extra + ["-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    # link next to the target and rename over it: a process that has the old library mapped keeps a valid image (writing
    # into the mapped file in place leaves it with a torn code object -- every launch then fails with "no ROCm-capable device")
    tmp = LIB + f".tmp{os.getpid()}"
    # --no-undefined: a kernel whose host stub the compiler dropped must fail the build here, not at dlopen on the GPU box
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

"""Build librpeflow_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One object per csrc/*.hip (compiled in parallel, only when the source or a header changed), then one link."""
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(SRC_DIR, "build")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(SRC_DIR, "librpeflow_hip.so")

# -ffp-contract=off: KNN/FPS must round exactly where the source says ("Exact arithmetic")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-fvisibility=hidden", "-Wno-unused-result"]
# per source: knn.hip's matrix kernel consumes every MFMA result on the VALU right away, so its accumulators belong in
# VGPRs (the default puts them in AGPRs and copies 16 registers per step)
# (-Wno-pass-failed: the insertion kernel asks for full unrolling of loops whose trip count depends on a template parameter the
# optimiser only unrolls for some instantiations; the remark is repeated per instantiation)
FILE_FLAGS = {"knn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wno-pass-failed"]}
# RPE_EXPERIMENTAL=1 (or --experimental): also the entry points include/rpeflow_hip.h lists under #ifdef RPE_EXPERIMENTAL
# (kernel probes for development); the default library exports only what rpeflow_amd calls
EXPERIMENTAL = ["-DRPE_EXPERIMENTAL"] if os.environ.get("RPE_EXPERIMENTAL") else []


def sources():
    return sorted(glob.glob(os.path.join(SRC_DIR, "*.hip")))


def headers():
    return glob.glob(os.path.join(SRC_DIR, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _command(src):
    return [_hipcc()] + FLAGS + EXPERIMENTAL + FILE_FLAGS.get(os.path.basename(src), []) + ["-I", INCLUDE, "-I", SRC_DIR, "-c", src, "-o", _obj(src)]


def _stamp(src):
    """Hash of the compiler path and the full command line an object was (or would be) built with: an object compiled
    with other flags (-ffp-contract=off, -amdgpu-mfma-vgpr-form: the bit-exact kernels depend on them) or by another
    hipcc is stale even if it is newer than its source."""
    return hashlib.sha256("\0".join(_command(src)).encode()).hexdigest()


def _stamp_matches(src):
    path = _obj(src) + ".cmd"
    return os.path.exists(path) and open(path).read().strip() == _stamp(src)


def _obj_stale(src, hdrs):
    return _stale(_obj(src), [src] + hdrs) or not _stamp_matches(src)


def needs_build():
    return _stale(LIB, sources() + headers()) or any(_obj_stale(s, headers()) for s in sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = headers()
    todo = [s for s in sources() if force or _obj_stale(s, hdrs)]

    def compile_one(src):
        cmd = _command(src)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(_obj(src) + ".cmd", "w") as f:
            f.write(_stamp(src) + "\n")

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    for stale in set(glob.glob(os.path.join(OBJ_DIR, "*.o"))) - {_obj(s) for s in sources()}:
        os.remove(stale)  # object of a source file that no longer exists
        if os.path.exists(stale + ".cmd"):
            os.remove(stale + ".cmd")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    if "--experimental" in sys.argv:
        EXPERIMENTAL[:] = ["-DRPE_EXPERIMENTAL"]
    print(build(force="--force" in sys.argv, verbose=True))

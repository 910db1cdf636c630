"""In-tree build of the C-ABI HIP library ``pretrain_gnns_amd/libpgnn.so`` for gfx950.

``python -m pretrain_gnns_amd.build`` (or ``__graft_entry__.build()``) compiles every
``csrc/*.hip`` with hipcc (cross-compiles without a GPU) and links one shared object.  Objects
are cached under ``csrc/_obj`` keyed on source mtime so rebuilds after an edit take seconds.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpgnn.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False, extra_flags=()):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "pgnn.h"))
    objs, procs = [], []
    for src in sources:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if out.strip():
            print("[%s]\n%s" % (src, out), file=sys.stderr)
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    # --ab: the A/B build (-DPGNN_AB): compile-time ablation instances behind PGNN_* knobs (tools/, profiles/): not what ships
    ab = "--ab" in sys.argv
    defs = tuple("-D" + sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--define")  # --define NAME: one more -D (same-box A/Bs of compile-time choices)
    print(build(verbose=True, force="--force" in sys.argv or ab or bool(defs), extra_flags=(("-DPGNN_AB",) if ab else ()) + defs))

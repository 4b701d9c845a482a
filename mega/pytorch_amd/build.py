"""Builds libmega_hip.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Flags: boxes.hip and frames.hip are built with -ffp-contract=off (bit-exact
box / pixel arithmetic: no FMA contraction the reference's separate torch ops do not have).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmega_hip.so")
STAMP = os.path.join(HERE, "csrc", ".build_stamp")
SOURCES = {
    "igemm.hip": [],
    "igemm8.hip": [],
    "igemm4.hip": [],
    "igemm2.hip": [],
    "stream1x1.hip": [],
    "spatial.hip": [],
    "boxes.hip": ["-ffp-contract=off"],
    # MFMA results in VGPRs: the attention loop otherwise keeps its O accumulators in AGPRs and moves them to VGPRs and
    # back around every (usually skipped) rescale -- 400 v_accvgpr moves in the kernel, 160 per loop iteration
    "relation.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "frames.hip": ["-ffp-contract=off"],      # (x / 255) * 255 - mean must round like the reference's three torch ops
    "fgfa.hip": [],
    "assemble.hip": [],
    "conv64.hip": [],
    "bneck64.hip": [],
}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
if os.environ.get("MEGA_BUILD_EXPERIMENTS") == "1":      # tools/gpu/ablate8.py: igemm8 ablation / timeline variants
    BASE_FLAGS.append("-DMEGA_EXPERIMENTS")


def _probe_flags(flags):
    """Per-file extra flags the installed hipcc accepts: `-mllvm -amdgpu-mfma-vgpr-form=1` is an internal LLVM option
    (a speed tweak: no AGPR <-> VGPR moves around the attention's accumulators) with no stability guarantee -- a compiler
    that does not know it would fail the whole build with 'Unknown command line argument' (ADVICE r03).  Probe once with
    an empty translation unit and drop what is rejected."""
    if not flags or not os.path.exists(HIPCC):
        return list(flags)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.hip")
        with open(src, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n__global__ void mega_probe() {}\n")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O1"] + list(flags) + ["-c", src, "-o", os.path.join(td, "probe.o")]
        if subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode == 0:
            return list(flags)
    sys.stderr.write("build.py: hipcc rejects %s -- building without it\n" % " ".join(flags))
    return []


def _digest():
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(CSRC, fn), "rb").read())
    h.update(" ".join(BASE_FLAGS).encode())
    h.update(repr(sorted(SOURCES.items())).encode())
    h.update(open(os.path.abspath(__file__), "rb").read())      # (the flag probe lives here)
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ and link libmega_hip.so.  Returns the library path."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB):
            return LIB  # prebuilt library on a box without the compiler
        raise RuntimeError("hipcc not found at %s and no prebuilt %s" % (HIPCC, LIB))
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        if "-mllvm" in extra:
            extra = _probe_flags(extra)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [HIPCC] + BASE_FLAGS + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

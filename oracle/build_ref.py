"""TEST INFRASTRUCTURE ONLY -- builds the reference's own CPU native ops into oracle/_ref/.

Recipe for ``oracle/_ref/mega_ref_C*.so``: the *unmodified-in-arithmetic* CPU half of the
reference's ``mega_core._C`` extension (``/root/reference/mega_core/csrc/vision.cpp`` +
``csrc/cpu/nms_cpu.cpp`` + ``csrc/cpu/ROIAlign_cpu.cpp``), compiled from the sources where they
lie under ``/root/reference``.  Nothing from the reference is copied into the repository:

* ``vision.cpp`` and every header are compiled straight from ``/root/reference``;
* the two ``cpu/*.cpp`` files pass ``Tensor.type()`` (a ``DeprecatedTypeProperties``) to
  ``AT_DISPATCH_FLOATING_TYPES`` (``cpu/nms_cpu.cpp:71``, ``cpu/ROIAlign_cpu.cpp:242``), which
  torch >= 1.11 rejects.  The recipe streams those two files through a 2-token rewrite
  (``.type()`` -> ``.scalar_type()`` on exactly those lines) into ``oracle/_ref/src/``
  (git-ignored scratch, regenerated on every build).  No arithmetic changes.

The resulting module exports ``nms`` and ``roi_align_forward`` with the reference pybind
signatures (``csrc/vision.cpp:9-12``); it is used (a) to pin ``oracle/native_oracle.c`` and the
python restatement, (b) injected as ``mega_core._C`` by ``oracle/ref_shim.py`` so the unmodified
python reference runs in this container.  The GPU box has no ``/root/reference``: there only the
prebuilt ``.so`` (which travels with the gpurun snapshot) is loaded.
"""
import glob
import importlib.util
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
REF_CSRC = os.path.join(REF_ROOT, "mega_core", "csrc")
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "mega_ref_C"


def _find_built():
    hits = sorted(glob.glob(os.path.join(OUT_DIR, NAME + "*.so")))
    return hits[0] if hits else None


def build(force=False, verbose=False):
    """Compile the reference CPU ops.  Returns the .so path or None when it cannot be built
    (no /root/reference and no prebuilt .so)."""
    built = _find_built()
    if built and not force:
        return built
    if not os.path.isdir(REF_CSRC):
        return built
    os.makedirs(os.path.join(OUT_DIR, "src"), exist_ok=True)
    patched = []
    for fn in ("nms_cpu.cpp", "ROIAlign_cpu.cpp"):
        src = open(os.path.join(REF_CSRC, "cpu", fn)).read()
        # 2-token rewrite on the AT_DISPATCH lines only (see module docstring).
        src2 = re.sub(r"AT_DISPATCH_FLOATING_TYPES\((\w+)\.type\(\)", r"AT_DISPATCH_FLOATING_TYPES(\1.scalar_type()", src)
        dst = os.path.join(OUT_DIR, "src", fn)
        with open(dst, "w") as f:
            f.write(src2)
        patched.append(dst)
    from torch.utils import cpp_extension
    build_dir = os.path.join(OUT_DIR, "build")
    os.makedirs(build_dir, exist_ok=True)
    cpp_extension.load(
        name=NAME,
        sources=[os.path.join(REF_CSRC, "vision.cpp")] + patched,
        extra_include_paths=[REF_CSRC],
        extra_cflags=["-O2", "-w"],
        build_directory=build_dir,
        verbose=verbose,
        is_python_module=False,
    )
    so = os.path.join(build_dir, NAME + ".so")
    final = os.path.join(OUT_DIR, NAME + ".so")
    if os.path.exists(so):
        import shutil
        shutil.copyfile(so, final)
    return _find_built()


def load():
    """Import the built module (python object with .nms / .roi_align_forward) or None."""
    so = build()
    if so is None:
        return None
    if NAME in sys.modules:
        return sys.modules[NAME]
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[NAME] = mod
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

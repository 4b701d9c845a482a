"""CPU: the C-ABI library loads without a GPU and exports every symbol include/mega_hip.h declares
(no compute calls here), and the ctypes signature table covers exactly that set."""
import os
import re

from mega.pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mega_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mega_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 14
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), "libmega_hip.so does not export " + n
    assert sorted(_lib.SIGNATURES.keys()) == names


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _lib.load()
    # NULL pointers / non-positive sizes must come back as MEGA_ERR_ARG before any launch
    assert lib.mega_conv2d_nhwc(None, None, None, None, None, None, 1, 1, 1, 64, 64, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1, None) == 1
    assert lib.mega_stem_conv_bn_relu(None, None, None, None, None, 1, 8, 8, 0, None) == 1
    assert lib.mega_nms_workspace_bytes(2, 6000) >= 2 * 6000 * 94 * 8
    assert lib.mega_nms(None, None, 9000, 0.5, 1, None, None, None, 0, None) == 1

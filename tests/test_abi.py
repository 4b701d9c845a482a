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
    # round 6 (FlowNetS): NULL operands; a caller-chosen split count is clamped so that every K range holds a K-tile
    assert lib.mega_conv2d_nhwc_subpixel(None, None, None, None, 1, 4, 4, 64, 64, 2, 8, 8, 1, 128, 0, 1, 1, None, 0, None) == 1
    assert lib.mega_conv2d_nhwc_ks(None, None, None, None, None, None, 1, 1, 1, 64, 64, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1, 2, None, 0, None) == 1
    assert lib.mega_flow_level_assemble(None, None, None, None, None, 1, 8, 8, 64, 64, 192, 4, 4, 1, 1, None) == 1
    assert lib.mega_flow_pred_finish(None, 64, None, 1.0, None, 1, 4, 4, 1, None) == 1
    assert lib.mega_flow_conv1_combine(None, None, None, 0, None, 1, 16, 0, 1, None) == 1
    assert lib.mega_fgfa_warp_aggregate_ring_pos(None, None, None, None, 3, 4, 4, 8, 8, None, 1, 1, None) == 1
    assert lib.mega_fgfa_warp_aggregate_ring_pos_batched(None, None, None, None, 3, 4, 4, 8, 8, None, 1, 2, 1, None) == 1
    assert lib.mega_conv2d_nhwc_ks_workspace_bytes(10, 8, 64 * 4, 1, 1) == 0           # one range: no workspace
    assert lib.mega_conv2d_nhwc_ks_workspace_bytes(10, 8, 64 * 4, 1, 3) == 2 * 10 * 8 * 4   # 4 K-tiles in 3 ranges of 2 -> 2 ranges
    assert lib.mega_conv2d_nhwc_ks_workspace_bytes(10, 8, 64 * 4, 1, 100) == 4 * 10 * 8 * 4   # never more ranges than K-tiles

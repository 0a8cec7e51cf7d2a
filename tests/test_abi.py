"""The C-ABI library loads without a GPU and exports every symbol the public header declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ngs_rasterizer.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ngs_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_three_reference_entry_points():
    syms = declared_symbols()
    for name in ("ngs_rasterize_forward", "ngs_rasterize_backward", "ngs_mark_visible"):
        assert name in syms
    # every compute entry point cites the reference interface it replaces
    text = open(HEADER).read()
    for cite in ("rasterizer.h:31-53", "rasterizer.h:55-84", "rasterizer.h:24-29"):
        assert cite in text


def test_library_exports_every_declared_symbol():
    from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C
    lib = _C.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name
        assert ctypes.cast(getattr(lib, name), ctypes.c_void_p).value


def test_size_queries_and_layout_need_no_gpu():
    from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C
    lib = _C.lib()
    g1, g2 = lib.ngs_required_geometry_bytes(1000), lib.ngs_required_geometry_bytes(2000)
    assert 0 < g1 < g2
    assert lib.ngs_required_geometry_bytes(0) >= 0
    assert lib.ngs_required_image_bytes(1920, 1080) > 120 * 68 * 256 * 8
    assert lib.ngs_required_binning_bytes(1_000_000) >= 1_000_000 * 16
    gl, bl, il = _C.state_layout(1000, 5000, 64, 48)
    offs = list(gl.values())
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert gl["grad_accumulator"] + 1000 * 64 <= g1
    assert bl["point_list"] == 0 and il["ranges"] % 256 == 0
    assert isinstance(lib.ngs_last_error(), bytes)  # empty unless an earlier call of this thread failed
    assert lib.ngs_profile_stage_count() == 18

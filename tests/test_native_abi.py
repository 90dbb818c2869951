"""No-GPU checks of the C-ABI boundary: the library builds/loads, exports every symbol include/mmseg_hip.h declares,
and the compute path refuses to run without device tensors (no CPU fallback)."""
import os

import pytest
import torch

from multimodal_segmentation_amd import _native, ops


def test_header_parses_and_library_exports_every_symbol():
    protos = _native.parse_header()
    assert len(protos) >= 50
    for name in ('mmseg_conv2d_fwd', 'mmseg_conv2d_wgrad', 'mmseg_bn_stats', 'mmseg_tps_warp_fwd', 'mmseg_adam',
                 'mmseg_segloss_stats', 'mmseg_spectral_fwd', 'mmseg_instnorm_spade_fwd'):
        assert name in protos
    _native.build()
    lib = _native.load()
    for name in protos:
        assert hasattr(lib, name), name
    # prototypes: every launcher ends with the stream parameter
    ret, types = protos['mmseg_conv2d_fwd']
    assert ret == "int" and types[-1] == "void*" and len(types) == 26


def test_workspace_queries_run_without_gpu():
    _native.build()
    assert _native.call('mmseg_conv2d_wgrad_workspace', 8, 256, 256, 64, 64, 3, 3) > 0
    assert _native.call('mmseg_colsum_workspace_floats', 524288, 8) >= 512 * 64
    assert _native.call('mmseg_norm_workspace_floats', 64) == 1024 * 2 * 64


def test_no_cpu_fallback():
    """The product path must fail loudly on host tensors: there is no eager/PyTorch fallback."""
    _native.build()
    x = torch.zeros(1, 8, 8, 4)
    w = torch.zeros(3, 3, 4, 4)
    with pytest.raises(_native.NativeLibraryError):
        ops.conv2d(x, w)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, 'LIB_PATH', str(tmp_path / 'nope.so'))
    monkeypatch.setattr(_native, '_lib', None)
    with pytest.raises(_native.NativeLibraryError):
        _native.load()


def test_header_is_plain_c_and_query_entry_points_work_from_c(tmp_path):
    """include/mmseg_hip.h compiles with a C compiler (-std=c99 -pedantic) and a C program linked against the shared
    library can call every entry point that launches nothing."""
    import shutil
    import subprocess
    from multimodal_segmentation_amd import _native
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    _native.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'c_abi_check')
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(root, 'include'),
                           os.path.join(root, 'tests', 'c_abi_check.c'), '-o', exe, '-L', libdir, '-l:libmmseg_hip.so',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-L/opt/rocm/lib', '-lamdhip64'])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0 and b'C ABI OK' in out.stdout, out.stdout

"""The 16-bit LDS image of the transposed weight-gradient staging (csrc/conv.hip: lp_row_slot / lp_pair) is free of bank conflicts
under the MI355X LDS model for every tile size and both source element widths: tools/lds_layout_check.py enumerates every
ds_write_b64 lane group of the transposing stores and every ds_read_b128 lane group of the MFMA operand reads (pure Python)."""
import importlib.util
import os

import pytest


def _tool():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'lds_layout_check.py')
    spec = importlib.util.spec_from_file_location('lds_layout_check', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('tile', [32, 64, 128, 192])
@pytest.mark.parametrize('source_bits', [32, 16])
def test_16bit_wgrad_lds_image_is_conflict_free(tile, source_bits):
    size, worst_write, worst_read = _tool().check(tile, source_bits)
    assert worst_write == 1 and worst_read == 1
    assert size <= (4 * tile + 32) * 16          # the kernel's LDS allocation per tile: (4 * T + 32) sixteen-byte slots

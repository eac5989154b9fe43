"""The lane-level model of the fused last SEANet level's schedule (tools/uptail_layout_model.py: tile walk, MFMA operand / result
mappings, LDS row formulas, carried rows, warm-up tile, the transposed convolution running ahead of the other stages) against the
plain layer formulas - on the CPU.  The kernel itself is checked on the GPU (tests/test_gpu_uptail.py)."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("uptail_layout_model", os.path.join(os.path.dirname(__file__), "..", "tools", "uptail_layout_model.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("T,tiles", [(70, 1), (70, 2), (33, 5)])  # ragged last tile; workgroups that start inside the utterance (warm-up tile)
def test_schedule_model_equals_the_layers(T, tiles):
    err, never_written = model.run(T, tiles)
    assert never_written == 0 and err < 1e-12

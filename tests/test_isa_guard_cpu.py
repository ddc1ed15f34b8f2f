"""The normal-burst kernel's burst loop as hipcc compiled it (tools/isa_guard_nb.py): prefetch loads back to back, partial wait
for the work ticket, no spills, no scratch, 4 waves per SIMD.  A source change that passes every parity test can still cost 3-4 %
of the headline through one compiler-placed s_waitcnt; this runs without a GPU (hipcc cross-compiles gfx950)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_nb_kernel_loop_as_compiled():
    spec = importlib.util.spec_from_file_location("isa_guard_nb", os.path.join(ROOT, "tools", "isa_guard_nb.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    errs, info = m.check(m.assembly())
    assert not errs, (errs, info)
    assert info["vgpr_count"] == 128 and info["sgpr_spill_count"] == 0 and info["private_segment_fixed_size"] == 0

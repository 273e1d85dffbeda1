"""The kernels' assembly must not contain the code shape behind round 2's GPU faults outside the one place where it was read
and found harmless (scripts/check_isa.py). CPU only: hipcc cross-compiles the three kernel compilations to
assembly. Validated against revision b25ca33, whose k_paths for scenes in global memory it flags four times."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_no_uniform_condition_rebuilt_as_a_lane_mask_inside_the_traversal_loops():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "scripts", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(verbose=False) == []


def test_detector_sees_the_pattern_and_ignores_its_parts(tmp_path):
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "scripts", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = tmp_path / "bad.s"
    bad.write_text("_Z3foov:\n\tv_cndmask_b32_e64 v7, 0, 1, s[12:13]\n\ts_nop 0\n\tv_cmp_ne_u32_e64 s[12:13], 1, v7\n")
    assert mod.instances(str(bad)) == [("_Z3foov", 2)]
    fine = tmp_path / "fine.s"
    fine.write_text("_Z3barv:\n\tv_cndmask_b32_e64 v7, 0, 1, s[12:13]\n\tv_cmp_ne_u32_e64 s[12:13], 1, v70\n\tv_cmp_ne_u32_e64 s[12:13], 2, v7\n")
    assert mod.instances(str(fine)) == []

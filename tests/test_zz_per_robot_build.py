"""Runs last: the per-robot kernel build (smpl_amd/csrc/specialize.cpp) must be what the GPU box actually ran.
The parity tests before this one hold for either build; a silent fall-back to the generic kernels would only show as
lost speed, so it is made loud here."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_per_robot_build_is_active(small_cfg):
    from smpl_amd import capi
    if os.environ.get("SMPLX_SPECIALIZE", "") == "0":
        pytest.skip("per-robot build disabled by SMPLX_SPECIALIZE=0")
    s = capi.Space.from_config(small_cfg)
    ok, note = s.specialized()
    assert ok, "generic kernels in use: " + note
    # Which compiler built the code object matters (the hiprtc that PyTorch bundles spills the
    # collision kernels to 256 VGPRs + scratch; the ROCm this package is built with does not).  "compiled by smplx_rtc"
    # / "disk cache" = the child-process compiler; "compiled in-process (...)" = the fallback for installs WITHOUT the
    # helper.  With the helper present next to the library the fallback must not have been taken.
    from smpl_amd import build
    print("per-robot build:", note)
    if os.path.exists(build.RTC):
        assert "in-process" not in note, "smplx_rtc is present but the in-process hiprtc fallback ran: " + note
    else:
        pytest.xfail("smplx_rtc helper missing next to the library: in-process hiprtc fallback (" + note + ")")

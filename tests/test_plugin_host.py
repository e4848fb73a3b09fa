"""The plugin boundary on a GPU-less host: the reference's own libsmrender (oracle/_ref) is the
host, our layouts are pinned against its headers at build time (oracle/hostsim.c), and the
action entry points fail loudly -- never fall back -- when no GPU is there."""
import pytest
import smrender_amd as sa
from oracle import pyoracle as po
import hostsim as hs

pytestmark = pytest.mark.skipif(not hs.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def test_layouts_pinned_against_reference_headers():
    assert hs.host().hs_abi_check() == 0


def test_draw_on_node_rule_is_rejected_like_the_reference():
    # act_draw_ini: "'draw' may be applied to ways or relations only" -> 1 (src/smrules_cairo.c:590-594)
    hs.set_chart(po.make_proj(45.0, 15.0, 20000.0, 256.0, 256.0, 300))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_NODE, "color=red", []) == 2001


def test_no_gpu_no_render():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    hs.set_chart(po.make_proj(45.0, 15.0, 20000.0, 256.0, 256.0, 300))
    assert hs.apply_rule(sa.lib(), "draw", hs.OSM_WAY, "color=red", []) == 1001      # _ini < 0: error, nothing drawn
    assert hs.apply_rule(sa.lib(), "cap", hs.OSM_NODE, "key=name;angle=auto", []) == 1001
    assert b"no HIP device" in sa.lib().smr_hip_last_error()

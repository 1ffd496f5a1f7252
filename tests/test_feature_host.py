"""Host logic of the Python mirror of feature::orb_extractor (openvslam_amd/feature.py) against a recording stand-in for the library: no
device needed. Upstream's setters (set_max_num_keypoints, set_scale_factor, set_num_scale_levels, set_initial_fast_threshold,
set_minimum_fast_threshold) each call initialize(); the mirror rebuilds its device handle and must carry the rule switches and the
schedule choices over."""
import ctypes as C

import numpy as np

from openvslam_amd import _lib, feature


class _FakeLib:
    def __init__(self):
        self.calls = []
        self.live = set()
        self.next_handle = 100

    def ovs_orb_create(self, params_ref, rows, cols, batch, device, handle_ref):
        p = params_ref._obj
        self.calls.append(("create", p.max_num_keypts, round(float(p.scale_factor), 4), p.num_levels, p.ini_fast_thr, p.min_fast_thr, rows, cols, batch, device))
        self.next_handle += 1
        handle_ref._obj.value = self.next_handle
        self.live.add(self.next_handle)
        return 0

    def ovs_orb_destroy(self, h):
        self.calls.append(("destroy", h.value))
        self.live.discard(h.value)
        return 0

    def ovs_orb_max_keypoints(self, h):
        return 2003

    def ovs_orb_tables(self, h, *arrays):
        return 0

    def ovs_orb_set_variant(self, h, idx, value):
        self.calls.append(("variant", h.value, idx, value))
        return 0

    def ovs_orb_set_fast_split(self, h, v):
        self.calls.append(("fast_split", h.value, v))
        return 0

    def ovs_orb_set_pipeline(self, h, v):
        self.calls.append(("pipeline", h.value, v))
        return 0


def _field_names():
    return [f[0] for f in _lib.OrbParams._fields_]


def test_setters_reinitialise_and_keep_the_switches(monkeypatch):
    fake = _FakeLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(_lib, "require_device", lambda: 1)
    assert _field_names()[:5] == ["max_num_keypts", "scale_factor", "num_levels", "ini_fast_thr", "min_fast_thr"]
    ex = feature.orb_extractor(feature.orb_params(1000), max_rows=480, max_cols=752, max_batch=2, device=0)
    assert fake.calls == [("create", 1000, 1.2, 8, 20, 7, 480, 752, 2, 0)] and ex.max_keypoints == 2003
    ex.set_variant("tree_switch_factor", 1)
    ex.set_fast_split(False)
    ex.set_pipeline(2)
    first = ex._h.value
    fake.calls.clear()
    ex.set_max_num_keypoints(1500)
    assert ex.get_max_num_keypoints() == 1500
    second = ex._h.value
    assert fake.calls == [("destroy", first), ("create", 1500, 1.2, 8, 20, 7, 480, 752, 2, 0), ("variant", second, 0, 1), ("fast_split", second, 0),
                          ("pipeline", second, 2)]
    fake.calls.clear()
    ex.set_scale_factor(1.5)
    ex.set_num_scale_levels(4)
    ex.set_initial_fast_threshold(30)
    ex.set_minimum_fast_threshold(10)
    creates = [c for c in fake.calls if c[0] == "create"]
    assert creates[-1][1:6] == (1500, 1.5, 4, 30, 10) and len(creates) == 4
    assert (ex.get_scale_factor(), ex.get_num_scale_levels(), ex.get_initial_fast_threshold(), ex.get_minimum_fast_threshold()) == (1.5, 4, 30, 10)
    assert len(ex.get_scale_factors()) == 4 and ex.num_keypts_per_level_.dtype == np.int32
    assert len(fake.live) == 1          # every replaced handle was destroyed
    last = ex._h.value
    del ex
    assert ("destroy", last) in fake.calls and not fake.live


def test_rectangle_mask_follows_upstream_ratios(monkeypatch):
    fake = _FakeLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(_lib, "require_device", lambda: 1)
    ex = feature.orb_extractor(feature.orb_params(500, mask_rects=[(0.0, 0.5, 0.25, 0.75)]), max_rows=100, max_cols=200)
    m = ex.create_rectangle_mask(200, 100)
    assert m.shape == (100, 200) and m.dtype == np.uint8
    assert (m[25:75, 0:100] == 0).all() and (m[:25] == 255).all() and (m[75:] == 255).all() and (m[25:75, 100:] == 255).all()
    assert ex.create_rectangle_mask(200, 100) is m          # cached per size

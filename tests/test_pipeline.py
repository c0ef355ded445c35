"""The resident pipeline (cvx_proj_amd/pipeline.py) against the chain of calls of the mirror class and against
the reference's outputs: the same kernels, so the same bits."""
import hashlib

import numpy as np
import pytest

from cvx_proj_amd.synth import config_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,mode", [("C1", "warp"), ("C1", "stitch"), ("C2", "warp"), ("C2", "none")])
def test_resident_pass_equals_the_chain_of_calls(native, golden, cfg, mode):
    from cvx_proj_amd import apap as A
    from cvx_proj_amd.pipeline import Pipeline
    p = config_pair(cfg)
    m = p.vertices.shape[0]
    center = np.random.default_rng(3).integers(0, 256, p.shape, dtype=np.uint8) if mode == "stitch" else None
    other = p.img if mode != "none" else None
    pipe = Pipeline()
    flat, canvas, grid = pipe.run_pair(p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma, other_img=other,
                                       center_img=center, want_grid=True)
    flat2, canvas2 = A.run_pair_by_calls(p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma, other_img=other,
                                         center_img=center)
    assert np.array_equal(flat, flat2)
    assert (canvas is None and canvas2 is None) or np.array_equal(canvas, canvas2)
    g = golden(cfg.lower() + "_ref")
    assert np.array_equal(grid, g["H_ref"])                     # the reference's own grid
    if mode == "warp":
        assert hashlib.sha256(canvas.tobytes()).digest() == g["warped_sha256"].tobytes()
    assert set(pipe.timeline) == {"host_setup_ms", "upload_and_enqueue_ms", "sync_and_download_ms", "total_ms"}
    # a second pair through the same object reuses its buffers
    flat3, _ = pipe.run_pair(p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma)
    assert np.array_equal(flat3, flat)


def test_pipeline_reports_the_reference_s_exceptions(native):
    from cvx_proj_amd.pipeline import Pipeline
    p = config_pair("C1")
    pipe = Pipeline()
    with pytest.raises(ValueError):
        pipe.run_pair(p.src[:, :1], p.dst, p.Hg, p.shape, p.shape, 20)


def test_equalise_stays_on_the_device_and_matches_the_host_call(native):
    from cvx_proj_amd.pipeline import Pipeline
    img = np.random.default_rng(1).integers(0, 200, (123, 77, 3), dtype=np.uint8)
    pipe = Pipeline()
    d = pipe.equalize(img)
    assert d.is_cuda and np.array_equal(d.cpu().numpy(), native.equalize_hist(img))
    assert np.array_equal(pipe.equalize(img, fetch=True), native.equalize_hist(img))


def test_page_locked_buffers_give_the_same_bytes(native, golden):
    """Images read into Pipeline.pinned_array() buffers and a page-locked ``canvas_out`` take asynchronous copies; warp and
    stitch must give the bytes of the pageable path, and the reference's canvas."""
    from cvx_proj_amd.pipeline import Pipeline
    p = config_pair("C2")
    m = p.vertices.shape[0]
    pipe = Pipeline()
    center = np.random.default_rng(3).integers(0, 256, p.shape, dtype=np.uint8)
    img_pin, cen_pin = pipe.pinned_array(p.img.shape), pipe.pinned_array(center.shape)
    np.copyto(img_pin, p.img)
    np.copyto(cen_pin, center)
    out_pin = pipe.pinned_array((p.final_h, p.final_w, 3))
    args = (p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma)
    for cen_a, cen_b in ((None, None), (center, cen_pin)):
        flat_a, canvas_a = pipe.run_pair(*args, other_img=p.img, center_img=cen_a)
        out_pin[:] = 0
        flat_b, canvas_b = pipe.run_pair(*args, other_img=img_pin, center_img=cen_b, canvas_out=out_pin)
        assert canvas_b is out_pin and np.array_equal(flat_a, flat_b) and np.array_equal(canvas_a, canvas_b)
        if cen_a is None:
            assert hashlib.sha256(canvas_b.tobytes()).digest() == golden("c2_ref")["warped_sha256"].tobytes()
    with pytest.raises(ValueError):
        pipe.run_pair(*args, other_img=img_pin, canvas_out=np.zeros((3, 3, 3), np.uint8))

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


def test_page_locked_passes_in_a_row_do_not_share_results(native):
    """The page-locked pass builds its keypoint table in a staging block and takes the ``.mat`` array and the status word
    through staging: consecutive passes on DIFFERENT pairs must each give their own result (an earlier pass's array is the
    caller's, not a view of the staging), with and without ``canvas_out``, and the status of a singular grid still raises."""
    from cvx_proj_amd.pipeline import Pipeline
    from cvx_proj_amd.synth import synth_pair
    pipe, plain = Pipeline(), Pipeline()
    pairs = [synth_pair(640, 360, 300, 20, seed) for seed in (5, 6, 7)]
    img_pin = pipe.pinned_array(pairs[0].img.shape)
    kept = []
    for k, p in enumerate(pairs):
        m = p.vertices.shape[0]
        args = (p.src, p.dst, p.Hg, p.shape, p.shape, m, p.gamma, p.sigma)
        np.copyto(img_pin, p.img)
        out_pin = pipe.pinned_array((p.final_h, p.final_w, 3)) if k != 1 else None
        flat, canvas = pipe.run_pair(*args, other_img=img_pin, canvas_out=out_pin)
        flat_ref, canvas_ref = plain.run_pair(*args, other_img=p.img)
        assert np.array_equal(flat, flat_ref) and np.array_equal(canvas, canvas_ref)
        lo = pipe._flat_host.data_ptr()
        assert not lo <= flat.ctypes.data < lo + pipe._flat_host.numel() * 8          # the caller's own memory, not the staging
        kept.append((flat, flat_ref.copy()))
    for flat, ref in kept:          # later passes did not write into earlier results
        assert np.array_equal(flat, ref)
    p = pairs[0]
    np.copyto(img_pin, p.img)
    with pytest.raises(np.linalg.LinAlgError):
        # every keypoint the same point: a singular system in every cell (the reference: LinAlgError from np.linalg.inv)
        pipe.run_pair(np.zeros_like(p.src), np.zeros_like(p.dst), p.Hg, p.shape, p.shape, 20, p.gamma, p.sigma, other_img=img_pin)
    flat, canvas = pipe.run_pair(p.src, p.dst, p.Hg, p.shape, p.shape, 20, p.gamma, p.sigma, other_img=img_pin)
    assert np.array_equal(flat, kept[0][1])     # and the pipeline works on afterwards


def test_float64_keypoints_through_every_resident_path(native):
    """Keypoints that are float64 arrays (values no float32 holds) give, through the resident pass, the chain of calls, the
    sharded solver and the pair dealer, the grid of the host-buffer call - which `test_keypoints_that_are_not_float32_vs_reference`
    pins to the reference - and not the grid of the same keypoints narrowed to float32."""
    import copy
    import torch
    from cvx_proj_amd import apap as A
    from cvx_proj_amd.dist import ShardedSolver, solve_pairs
    from cvx_proj_amd.pipeline import Pipeline
    p = config_pair("C1")
    rng = np.random.default_rng(11)
    q = copy.copy(p)
    q.src = p.src.astype(np.float64) + rng.uniform(-1e-4, 1e-4, p.src.shape)
    q.dst = p.dst.astype(np.float64) + rng.uniform(-1e-4, 1e-4, p.dst.shape)
    m = p.vertices.shape[0]
    H, _ = native.local_homography(q.src, q.dst, q.vertices, q.gamma, q.sigma, want_weights=False)
    H32, _ = native.local_homography(q.src.astype(np.float32), q.dst.astype(np.float32), q.vertices, q.gamma, q.sigma, want_weights=False)
    assert not np.array_equal(H, H32)
    flat, canvas, grid = Pipeline().run_pair(q.src, q.dst, q.Hg, q.shape, q.shape, m, q.gamma, q.sigma, other_img=q.img, want_grid=True)
    assert np.array_equal(grid, H)
    flat2, canvas2 = A.run_pair_by_calls(q.src, q.dst, q.Hg, q.shape, q.shape, m, q.gamma, q.sigma, other_img=q.img)
    assert np.array_equal(flat, flat2) and np.array_equal(canvas, canvas2)
    dev = torch.device("cuda", 0)
    s = ShardedSolver(q, dev, same_bits=True)
    assert np.array_equal(s.solve().cpu().numpy().reshape(H.shape), H)
    assert np.array_equal(s.warp().cpu().numpy(), canvas)
    grids = solve_pairs([q, p], dev)
    assert np.array_equal(grids[0], H) and not np.array_equal(grids[1], H)

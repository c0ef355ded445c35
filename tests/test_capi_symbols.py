"""The C-ABI library loads without a GPU, exports every symbol include/apap_hip.h
declares, and fails loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "apap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apap_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(native):
    names = declared_symbols()
    assert len(names) >= 20
    handle = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in apap_hip.h but not exported"
        assert n in native.SIGNATURES, f"{n} has no ctypes signature in _native.SIGNATURES"
    assert sorted(native.SIGNATURES) == names


def test_version_and_constants(native):
    assert b"gfx950" in native.lib().apap_version()
    text = open(os.path.join(ROOT, "include", "apap_hip.h")).read()
    assert int(re.search(r"#define APAP_TABLE_STRIDE (\d+)", text).group(1)) == native.TABLE_STRIDE
    assert int(re.search(r"#define APAP_DENORM_DOUBLES (\d+)", text).group(1)) == native.DENORM_DOUBLES
    for name, val in (("OK", 0), ("ERR_INVALID_ARG", 1), ("ERR_NO_DEVICE", 2), ("ERR_HIP", 3), ("ERR_SINGULAR", 4),
                      ("ERR_INDEX", 5), ("ERR_WORKSPACE", 6)):
        assert int(re.search(rf"#define APAP_{name} (\d+)", text).group(1)) == val == getattr(native, name)


def test_contexts_hold_the_options_not_the_process(native):
    """No process-wide switches: two contexts carry different options side by side, NULL means the
    immutable defaults, invalid values are refused, and nothing of it needs a GPU."""
    a = native.Context(variant=native.VARIANT_VALU, eigen=native.EIGEN_JACOBI)
    b = native.Context()
    assert (a.get("variant"), a.get("eigen"), a.get("careful")) == (native.VARIANT_VALU, native.EIGEN_JACOBI, 1)
    assert (b.get("variant"), b.get("eigen"), b.get("careful"), b.get("warp_rows")) == (native.VARIANT_AUTO, native.EIGEN_AUTO, 1, 1)
    v = ctypes.c_int(-1)
    assert native.lib().apap_ctx_get_option(None, native.OPT_WANT_WAVES, ctypes.byref(v)) == native.OK and v.value == 4096
    assert native.lib().apap_ctx_set_option(None, native.OPT_CAREFUL, 0) == native.ERR_INVALID_ARG      # NULL cannot be changed
    assert (b.get("moments"), b.get("weights_f32")) == (30, 0)           # the bit-identical form is the default
    for name, bad in (("variant", 9), ("eigen", -1), ("careful", 2), ("warp_rows", 3), ("want_waves", 0), ("moments", 25), ("weights_f32", 2)):
        with pytest.raises(ValueError):
            b.set(name, bad)
    # the workspace a solve needs follows the context's variant, not a global
    n, cells = 2000, 40000
    ws_default = native.lib().apap_solve_workspace_bytes(None, n, cells)
    ws_valu = native.lib().apap_solve_workspace_bytes(native._h(a), n, cells)
    assert ws_default > 0 and ws_valu > 0 and ws_valu != ws_default
    assert native.lib().apap_solve_workspace_bytes(native._h(b), n, cells) == ws_default
    c24 = native.Context(moments=24)
    assert native.lib().apap_solve_workspace_bytes(native._h(c24), n, cells) * 30 == ws_default * 24     # 24 of 30 slab rows
    c24.close()
    prof = a.profile_read()
    assert set(prof) == set(native.PROF_NAMES) and all(v == (0.0, 0) for v in prof.values())
    import threading
    seen = {}

    def worker(k):      # contexts created and used on other threads are independent objects
        c = native.Context(want_waves=1000 + k)
        seen[k] = c.get("want_waves")
        c.close()
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen == {k: 1000 + k for k in range(4)} and b.get("want_waves") == 4096
    a.close()
    b.close()
    native.lib().apap_ctx_destroy(None)     # a no-op


def test_no_process_wide_state_in_the_library_sources():
    """The product's native sources keep no mutable globals for options or profiling and read no
    environment variable (knobs are context options)."""
    csrc = os.path.join(ROOT, "cvx_proj_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            text = open(os.path.join(csrc, f)).read()
            assert "getenv" not in text, f
            assert not re.search(r"^\s*(static\s+)?(int|bool)\s+g_\w+\s*=", text, flags=re.M), f


def test_no_cpu_fallback(native):
    """Without a device every compute entry point must refuse, not compute on the host."""
    if native.lib().apap_device_count() > 0:
        pytest.skip("a GPU is visible")
    rng = np.random.default_rng(0)
    src = rng.random((16, 2)).astype(np.float32) * 100
    dst = src + 1
    verts = rng.random((2, 2, 2)) * 100
    with pytest.raises(native.ApapError) as e:
        native.local_homography(src, dst, verts, 0.5, 100.0)
    assert e.value.code == native.ERR_NO_DEVICE
    with pytest.raises(native.ApapError) as e:
        native.invert_normalize_flatten(np.tile(np.eye(3, dtype=np.float32), (4, 1, 1)))
    assert e.value.code == native.ERR_NO_DEVICE
    with pytest.raises(native.ApapError) as e:
        native.local_warp(np.zeros((8, 8, 3), np.uint8), np.tile(np.eye(3, dtype=np.float32), (1, 1, 1, 1)),
                          [0.0, 8.0], [0.0, 8.0], 8, 8, 0, 0)
    assert e.value.code == native.ERR_NO_DEVICE


def test_argument_errors(native):
    with pytest.raises(ValueError):
        native.local_homography(np.zeros((4, 2), np.float32), np.zeros((5, 2), np.float32), np.zeros((2, 2, 2)), .5, 100)
    with pytest.raises(ValueError):
        native.local_homography(np.zeros((4, 2), np.float32), np.zeros((4, 2), np.float32), np.zeros((2, 2)), .5, 100)
    with pytest.raises(native.ApapError) as e:
        native.host_prepare(np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32))
    assert e.value.code == native.ERR_INVALID_ARG
    assert "at least 2" in str(e.value)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under cvx_proj_amd/ may reference it."""
    pkg = os.path.join(ROOT, "cvx_proj_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "apap_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_abi_generation_is_checked_on_load(native, monkeypatch):
    """A build of another ABI generation keeps the symbol names but not the argument lists (generation 2 put the
    context first): the binding must refuse it instead of calling into it."""
    assert native.lib().apap_abi_version() == native.ABI_VERSION
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "ABI_VERSION", native.ABI_VERSION + 1)
    with pytest.raises(native.ApapError, match="ABI generation"):
        native.lib()
    monkeypatch.undo()
    assert native.lib().apap_abi_version() == native.ABI_VERSION


def test_plan_cells_option_fixes_the_launch_plan(native):
    """APAP_OPT_PLAN_CELLS: kernel choice and keypoint splits as for ONE pair of that many cells, whatever the call
    holds (a shard of a mesh then sums like the whole mesh).  Visible without a GPU in the workspace size: a slab of 30
    doubles per cell and keypoint split."""
    lib = native.lib()
    slab = lambda cells: 30 * ((cells + 63) // 64 * 64) * 8      # noqa: E731
    ctx = native.Context(variant=native.VARIANT_MFMA)
    whole, shard = 160000, 20000                                 # C4 and one of its 8 row blocks, 5000 keypoints
    assert lib.apap_solve_workspace_bytes(native._h(ctx), 5000, whole) == slab(whole)            # 10 000 waves: one split
    by_itself = lib.apap_solve_workspace_bytes(native._h(ctx), 5000, shard)
    assert by_itself == 4 * slab(shard)                                                          # 313 tiles alone: four keypoint splits to fill the chip
    ctx.set("plan_cells", whole)
    assert lib.apap_solve_workspace_bytes(native._h(ctx), 5000, shard) == slab(shard)            # as the whole mesh: one split
    assert lib.apap_solve_batch_workspace_bytes(native._h(ctx), 5000, shard, 3) == 3 * slab(shard)
    ctx.set("plan_cells", 0)
    assert lib.apap_solve_workspace_bytes(native._h(ctx), 5000, shard) == by_itself
    with pytest.raises(native.ApapError):
        ctx.set("plan_cells", -1)
    ctx.close()


def test_the_command_line_needs_no_torch():
    """importing the drop-in module and loading the library must not pull torch into the process (2 s of
    start-up for ~1 ms of GPU work); torch is borrowed only when it is already there."""
    import subprocess
    import sys
    code = ("import sys; import cvx_proj_amd.apap as A; from cvx_proj_amd import _native; _native.lib(); "
            "import cvx_proj_amd.utils, cvx_proj_amd.baseline_stitch_test, cvx_proj_amd.synth, cvx_proj_amd.evaluate; "
            "assert 'torch' not in sys.modules, 'torch was imported'; "
            "import numpy as np, tempfile; A.save2mat('H31_apap', np.zeros((4, 9)), name='H', prefix=tempfile.mkdtemp() + '/'); "
            "assert 'scipy' not in sys.modules, 'the .mat writer of the command imported scipy'; print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-1500:]

"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/audiopure.h declares, and its
host-only entry points (context, schedule, sizes, error paths) behave.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from audiopure_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return N.lib()


def _cfg(C_=256, S=256, prec=0):
    return N.ApConfig(C_, S, 36, 12, 128, 512, 512, 200, 1e-4, 0.02, prec)


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "audiopure.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ap_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 28
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/audiopure.h but not exported"
    assert names == set(N.SIGNATURES), names ^ set(N.SIGNATURES)


def test_ctx_create_schedule_and_sizes(lib, golden):
    h = C.c_void_p()
    cfg = _cfg()
    N.check(lib.ap_ctx_create(C.byref(cfg), C.byref(h)))
    # default schedule = closed form in double; the reference's sequential-fp32 tables differ by < 1 ulp-ish
    for which, key in ((0, "Beta"), (1, "Alpha"), (2, "Alpha_bar"), (3, "Sigma")):
        buf = (C.c_float * 200)()
        N.check(lib.ap_ctx_get_schedule(h, which, buf, 200))
        # Sigma: the reference's fp32 (1 - Alpha_bar) cancels at small t, so its own table is ~4e-5 off the closed form
        np.testing.assert_allclose(np.array(buf[:]), golden[f"sched/{key}"], rtol=1e-4 if key == "Sigma" else 3e-6, atol=0)
    # installing the reference tables makes them bit-identical
    arrs = [N.farr(golden[f"sched/{k}"].tolist()) for k in ("Beta", "Alpha", "Alpha_bar", "Sigma")]
    N.check(lib.ap_ctx_set_schedule(h, *arrs, 200))
    buf = (C.c_float * 200)()
    N.check(lib.ap_ctx_get_schedule(h, 2, buf, 200))
    assert np.array_equal(np.array(buf[:], dtype=np.float32), golden["sched/Alpha_bar"])
    assert lib.ap_ctx_get_schedule(h, 2, buf, 199) != 0 and b"entries" in lib.ap_last_error()
    # blob size = number of parameters of the reference net (24,071,681 for configs/config.json)
    assert lib.ap_wavenet_blob_elems(C.byref(cfg)) == 24071681
    # workspace: h ping-pong + skip + 2 clip buffers + FiLM vectors
    need = lib.ap_workspace_bytes(h, 4, 16000)
    assert need >= 4 * (3 * 4 * 256 * 16000 + 2 * 4 * 16000 + 36 * 256 + 512)
    assert need < 4 * (3 * 4 * 256 * 16000) * 1.01
    N.check(lib.ap_ctx_destroy(h))


def test_bad_configs_fail_loudly(lib):
    h = C.c_void_p()
    for cfg, word in ((_cfg(256, 128), b"skip_channels"), (_cfg(96, 96), b"res_channels"), (_cfg(prec=7), b"precision")):
        assert lib.ap_ctx_create(C.byref(cfg), C.byref(h)) == -22
        assert word in lib.ap_last_error()
    with pytest.raises(N.NativeError):
        N.check(lib.ap_ctx_create(None, C.byref(h)), "ap_ctx_create")


def test_unloaded_context_refuses_compute(lib):
    h = C.c_void_p()
    cfg = _cfg(64, 64)
    N.check(lib.ap_ctx_create(C.byref(cfg), C.byref(h)))
    rc = lib.ap_eps_fwd(h, None, 0.0, None, 1, 100, None, 0, None)
    assert rc == -22 and b"not loaded" in lib.ap_last_error()
    assert lib.ap_purify_ddpm(h, None, 0, 1, None, 0, 0, None, 1, 100, None, 0, None) == -22
    assert lib.ap_purify_ddpm(h, None, 201, 1, None, 0, 0, None, 1, 100, None, 0, None) == -22
    N.check(lib.ap_ctx_destroy(h))
    assert lib.ap_m5_blob_elems(10, 32, 80) == 25674   # 25,290 parameters + 4x(running_mean, running_var)
    assert lib.ap_melspec_db(None, None, 32, 0, 1, 16000, None) == -22


def test_skip_group_option_is_validated_and_sizes_the_workspace(lib):
    """ap_ctx_set_skip_group (deferred-skip form of the bf16 block): only on a bf16 context with 256 channels, group size in
    [0, num_res_layers]; ap_workspace_bytes grows by exactly G images of [B][L][C] bf16."""
    h = C.c_void_p()
    cfg = _cfg(256, 256, prec=N.AP_PREC_BF16)
    N.check(lib.ap_ctx_create(C.byref(cfg), C.byref(h)))
    base = lib.ap_workspace_bytes(h, 3, 1001)
    for G in (1, 6, 36):
        N.check(lib.ap_ctx_set_skip_group(h, G))
        grown = lib.ap_workspace_bytes(h, 3, 1001) - base
        assert G * 3 * 1001 * 256 * 2 <= grown < G * 3 * 1001 * 256 * 2 + 256
    N.check(lib.ap_ctx_set_skip_group(h, 0))
    assert lib.ap_workspace_bytes(h, 3, 1001) == base
    assert lib.ap_ctx_set_skip_group(h, 37) == -22 and lib.ap_ctx_set_skip_group(h, -1) == -22
    N.check(lib.ap_ctx_destroy(h))
    f32 = C.c_void_p()
    cfg = _cfg(256, 256)
    N.check(lib.ap_ctx_create(C.byref(cfg), C.byref(f32)))
    assert lib.ap_ctx_set_skip_group(f32, 6) == -22 and b"AP_PREC_BF16" in lib.ap_last_error()
    N.check(lib.ap_ctx_set_skip_group(f32, 0))
    assert lib.ap_resblock_fwd_gate(f32, 0, None, None, None, None, 1, 100, None) == -22
    assert lib.ap_skip_gemm(f32, 0, 1, None, None, 0, 1, 100, None) == -22
    N.check(lib.ap_ctx_destroy(f32))


def test_cpu_tensors_are_rejected():
    import torch
    with pytest.raises(N.NativeError):
        N.ptr(torch.zeros(4))


def test_product_library_ships_no_tools_kernels_or_debug_hooks():
    """The shipped library holds the product kernels only: no ap_debug_* hook, neither the round-1 one-tile-per-workgroup bf16
    block nor the one-wave-per-SIMD experiment (both live in the -DAP_TOOLS library under tools/lib/, which must not sit
    beside the product library)."""
    import os
    import subprocess
    from audiopure_amd import _native as N
    libdir = os.path.dirname(N.LIB_PATH)
    assert [f for f in os.listdir(libdir) if f.endswith(".so")] == ["libaudiopure_hip.so"]
    sym = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "ap_debug_" not in sym
    for name in ("resblock_bf16_kernel", "resblock_bf16w_kernel"):
        assert name not in sym, name
    assert "resblock_bf16p_kernel" in sym

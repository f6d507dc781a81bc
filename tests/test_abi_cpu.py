"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/abr_env.h declares, its structs match the ctypes mirrors,
and argument validation fails with codes + messages (no GPU needed: these
paths return before any HIP call)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "abr_env.h")


@pytest.fixture(scope="module")
def L():
    from abrsimulator_amd import _lib
    _lib.build()
    return _lib


def _declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(abr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(L):
    names = _declared()
    assert "abr_env_step" in names and "abr_mpc_select" in names and len(names) >= 14
    lib = L.lib()
    bound = {n for n, _, _ in L.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), f"{n} declared in abr_env.h but not exported"
        assert n in bound, f"{n} declared in abr_env.h but not bound in _lib.SYMBOLS"
    assert lib.abr_abi_version() == 4 == L.ABI_VERSION


def test_struct_layout_matches_header(L):
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "abr_env.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(abr_env_config), offsetof(abr_env_config, ladder),
         offsetof(abr_env_config, max_ticks), offsetof(abr_env_config, speed),
         offsetof(abr_env_config, interval));
  printf("%zu %zu %zu\n", sizeof(abr_mpc_config), offsetof(abr_mpc_config, chunk_length),
         offsetof(abr_mpc_config, startup_weight));
  printf("%zu %zu %zu %zu\n", sizeof(abr_env_state_view), offsetof(abr_env_state_view, bw_hist),
         sizeof(abr_mpc_options), offsetof(abr_mpc_options, hist_len_dev));
  printf("%d %d %d %d\n", ABR_OBS_DIM, ABR_F64_DIM, ABR_MAX_RATES, ABR_MAX_HORIZON);
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    a = list(map(int, out[0].split()))
    E = L.EnvConfig
    assert a == [C.sizeof(E), E.ladder.offset, E.max_ticks.offset, E.speed.offset, E.interval.offset]
    b = list(map(int, out[1].split()))
    M = L.MpcConfig
    assert b == [C.sizeof(M), M.chunk_length.offset, M.startup_weight.offset]
    c_ = list(map(int, out[2].split()))
    assert c_ == [C.sizeof(L.StateView), L.StateView.bw_hist.offset, C.sizeof(L.MpcOptions),
                  L.MpcOptions.hist_len_dev.offset]
    assert list(map(int, out[3].split())) == [L.OBS_DIM, L.F64_DIM, L.MAX_RATES, L.MAX_HORIZON]
    assert len(L.OBS_ROWS) == L.OBS_DIM and len(L.F64_ROWS) == L.F64_DIM


def _cfg(L, **kw):
    c = L.EnvConfig()
    c.n_rates, c.video_length = 6, 48
    c.chunk_length, c.max_buffer, c.start_up_length, c.interval = 4.0, 20.0, 8.0, 1.0
    c.speed = 1.0
    for i, b in enumerate([0.3, 0.75, 1.2, 1.85, 2.85, 4.3]):
        c.ladder[i] = b
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_workspace_bytes_and_validation(L):
    lib = L.lib()
    n = C.c_size_t()
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L)), 65536, C.byref(n)) == 0
    per_lane = 8 * 8 + 8 + 15 * 4 + 2 + 48 + 48 * 8 + 4 * 8 + 4 + 72   # incl. per-lane speed state, the running variance sum, MPC action + predictor scratch
    assert n.value >= 65536 * per_lane and n.value < 65536 * per_lane + 32 * 2 ** 20
    n2 = C.c_size_t()
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L)), 131072, C.byref(n2)) == 0
    assert n2.value - n.value == 65536 * per_lane      # linear in lanes, tables shared
    for bad in (dict(n_rates=0), dict(n_rates=17), dict(video_length=0), dict(chunk_length=0.0),
                dict(interval=-1.0), dict(max_buffer=0.0), dict(speed=0.0)):
        rc = lib.abr_env_workspace_bytes(C.byref(_cfg(L, **bad)), 64, C.byref(n))
        assert rc == -1, bad
        assert len(lib.abr_last_error()) > 0
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L)), 0, C.byref(n)) == -1
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L, interval=1e-9)), 64, C.byref(n)) == -4
    # a tick bound under which no episode can finish is an error at create time, not 100 % timeouts
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L, max_ticks=48 * 400)), 64, C.byref(n)) == -1
    assert b"ABR_DONE_TIMEOUT" in lib.abr_last_error()
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L, max_ticks=49 * 400)), 64, C.byref(n)) == 0
    # the default bound is never silently capped below the live-stream minimum
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L, video_length=65535, chunk_length=4.0)), 64,
                                       C.byref(n)) == 0
    assert n.value > 2 * 8 * 2 * 65536 * 400
    assert lib.abr_env_workspace_bytes(C.byref(_cfg(L, video_length=65535, chunk_length=100.0)), 64,
                                       C.byref(n)) == -4
    with pytest.raises(L.AbrError):
        L.check(lib.abr_env_workspace_bytes(C.byref(_cfg(L, n_rates=99)), 64, C.byref(n)))


def test_create_rejects_bad_workspace_before_touching_the_gpu(L):
    lib = L.lib()
    h = C.c_void_p()
    one = C.c_void_p(256)
    rc = lib.abr_env_create(C.byref(_cfg(L)), one, one, one, 1, 64, None, 0, None, C.byref(h))
    assert rc == -2 and b"workspace" in lib.abr_last_error()
    rc = lib.abr_env_create(C.byref(_cfg(L)), one, one, one, 1, 64, C.c_void_p(257), 1 << 30, None,
                            C.byref(h))
    assert rc == -2
    rc = lib.abr_env_create(C.byref(_cfg(L)), one, one, one, 1, 64, C.c_void_p(512), 1024, None,
                            C.byref(h))
    assert rc == -2 and b"need" in lib.abr_last_error()
    rc = lib.abr_env_create(C.byref(_cfg(L)), None, one, one, 1, 64, C.c_void_p(512), 1 << 30, None,
                            C.byref(h))
    assert rc == -1
    assert lib.abr_env_step(None, one, None, None, None, None) == -1
    assert lib.abr_env_reset(None, one, None, None, None, None) == -1


def test_mpc_validation(L):
    lib = L.lib()
    m = L.MpcConfig()
    m.n_rates, m.horizon, m.video_length = 6, 5, 48
    one = C.c_void_p(256)
    args = [one] * 7 + [None, one, None, None]
    assert lib.abr_mpc_select(C.byref(m), *args, 0, None) == -1          # n_lanes < 1
    m.horizon = 1                                                        # reference crashes (mpc.py:186)
    assert lib.abr_mpc_select(C.byref(m), *args, 4, None) == -1
    m.horizon, m.n_rates = 8, 16                                         # 16^8 > int32
    assert lib.abr_mpc_select(C.byref(m), *args, 4, None) == -4
    m.horizon, m.n_rates = 5, 6
    assert lib.abr_mpc_select(C.byref(m), None, *args[1:], 4, None) == -1   # NULL pointer


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; the product path must not reach it: no
    import, include, dlopen or path reference anywhere under abrsimulator_amd/."""
    pkg = os.path.join(ROOT, "abrsimulator_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|oracle\.|oracle/|abr_oracle|libabr_oracle|pyloop)")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert not pat.search(txt), os.path.join(dp, f)


def test_env_refuses_cpu_device(L):
    import abrsimulator_amd as A
    mpd = A.MPD(4, 4, 20, 8, A.Chunk([1.0, 2.0]))
    with pytest.raises(ValueError):
        A.BatchedABREnv(mpd, A.QOEMetric(1, 1, 1, 1), A.NetworkInfo(1.0, [1.0, 2.0]), 8, device="cpu")


def test_product_library_ships_only_selectable_kernels(L):
    """The product code object holds what `auto` can select plus the explicit cross-checks; the rejected pipelines
    (the asynchronous one of round 3, the ring-coupled one of round 5) and the cycle-stamp instrumentation live in
    diagnostic builds outside the package (tools/diag/csrc), the out-of-line jump search is gone, and impl 4 / 6 are
    refused by the product.  abrsimulator_amd/csrc holds only what libabr_hip.so is built from."""
    blob = open(L.SO_PATH, "rb").read()
    assert b"env_split3_kernel" in blob and b"env_split_kernel" in blob and b"env_jump_kernel" in blob
    for sym in (b"env_async_kernel", b"env_ring3_kernel", b"env_pair3_kernel", b"g_st_acc", b"g_wg_t", b"g_async_stats", b"jump_fix",
                b"abr_debug_read_stamps", b"abr_debug_read_wg_times"):
        assert sym not in blob, sym
    src = os.path.join(ROOT, "abrsimulator_amd", "csrc")
    assert sorted(f for f in os.listdir(src) if f.endswith((".h", ".hip"))) == [
        "abr_env.hip", "abr_env_roles.h", "abr_exact_jump.h", "abr_lane_jump.h", "abr_tick_tables.h"]
    hot = open(os.path.join(src, "abr_env.hip")).read()
    assert "#ifdef ABR_WITH_ASYNC\n#include \"abr_env_async.h\"" in hot and "#ifdef ABR_WITH_RING\n#include \"abr_env_ring.h\"" in hot
    assert "noinline" not in open(os.path.join(src, "abr_exact_jump.h")).read()
    # the package names no diagnostic library: they are loaded only when a caller passes one (library= / ABR_HIP_LIB)
    pkg = os.path.join(ROOT, "abrsimulator_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            text = open(os.path.join(pkg, f)).read()
            assert "libabr_hip_" not in text and "ASYNC_SO" not in text, f
    assert [L.lib().abr_env_has_impl(i) for i in range(9)] == [1, 1, 1, 1, 0, 1, 0, 0, 0]


def test_env_kernels_have_one_barrier_and_no_calls():
    """Structural properties of the device code, read off the ISA (make asm, ~20 s; skipped without hipcc): every
    role-split env kernel holds exactly ONE s_barrier (the iteration loop and its barrier are written once,
    csrc/abr_env_roles.h) and no env kernel calls a function (the out-of-line jump search of rounds 1-3 is gone from the
    download loop)."""
    import shutil
    import subprocess
    src = os.path.join(ROOT, "abrsimulator_amd", "csrc")
    asm = os.path.join(src, "abr_env.s")
    deps = [os.path.join(src, f) for f in os.listdir(src) if f.endswith((".hip", ".h"))]
    if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(d) for d in deps):
        if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            pytest.skip("no hipcc here: the ISA cannot be regenerated")
        subprocess.run(["make", "-C", src, "-s", "asm"], check=True, capture_output=True, timeout=600)
    text = open(asm).read()
    # kernels by their .amdhsa_kernel directive (present for every kernel whatever the mangling scheme), bodies by label
    names = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M)
    kernels = {}
    for name in names:
        k = re.search(r"env_(split3|split|jump|advance)_kernel", name)
        mode = re.search(r"kernelILi(\d)E", name)
        if not k or not mode:
            continue
        # (to the end of the FUNCTION, not to the first s_endpgm: the role-split kernels return early under abr_debug_selfcheck)
        body = re.search(r"^" + re.escape(name) + r":.*?^\.Lfunc_end\d+:", text, re.S | re.M)
        assert body, name
        kernels[(k.group(1), int(mode.group(1)))] = body.group(0)
    assert {("split3", 1), ("split3", 2), ("split3", 3), ("split", 1), ("split", 2), ("split", 3),
            ("jump", 0), ("jump", 1), ("jump", 2), ("jump", 3)} <= set(kernels)
    assert not any("ring3" in n or "pair3" in n or "async" in n for n in names)         # the product carries no rejected pipeline
    for (kind, mode), body in kernels.items():
        assert "s_swappc" not in body and "s_setpc" not in body, (kind, mode)
        if mode == 9:                       # abr_debug_selfcheck's instances: the product signature, a body that only answers
            assert kind in ("split3", "split") and "s_barrier" not in body and len(body.splitlines()) < 120, (kind, len(body.splitlines()))
        elif kind in ("split3", "split"):
            assert body.count("s_barrier") == 1, (kind, mode, body.count("s_barrier"))
            assert body.count("s_endpgm") == 1, (kind, mode)        # no early return in a product instance
        else:
            assert "s_barrier" not in body, (kind, mode)

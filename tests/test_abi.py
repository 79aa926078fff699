"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol that
include/soccer_hip.h declares, and argument validation (the reference's asserts) happens before
any device work.  No compute calls here — those are the -m gpu tests."""
import ctypes as C
import os
import re

import pytest

from gym_soccer_littman94_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "soccer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:soccer|batched)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libsoccer_hip.so does not export %s" % name
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes prototypes and header disagree"
    assert lib.soccer_abi_version() == 3


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors of the ABI structs have the size and field offsets the C compiler gives the header."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text("""
#include <stdio.h>
#include <stddef.h>
#include "soccer_hip.h"
int main(void) {
    printf("%zu %zu %zu\\n", sizeof(soccer_config), sizeof(soccer_step_args), sizeof(soccer_rollout_args));
    printf("%zu %zu %zu %zu\\n", offsetof(soccer_config, slip_prob), offsetof(soccer_config, seed),
           offsetof(soccer_config, flags), offsetof(soccer_config, stream));
    printf("%zu %zu %zu\\n", offsetof(soccer_rollout_args, act_stride), offsetof(soccer_rollout_args, out_stride),
           offsetof(soccer_rollout_args, mix_a));
    printf("%zu %zu %zu %zu %zu\\n", sizeof(soccer_scalar_io), offsetof(soccer_scalar_io, act_a), offsetof(soccer_scalar_io, obs),
           offsetof(soccer_scalar_io, u_step), offsetof(soccer_scalar_io, u_reset));
    printf("%zu %zu %d\\n", sizeof(soccer_rollout_extra), offsetof(soccer_rollout_extra, prob_code), SOCCER_COMM_ID_BYTES);
    return 0;
}
""")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [int(x) for x in out]
    Cfg, St, Ro, Sc = _lib.Config, _lib.StepArgs, _lib.RolloutArgs, _lib.ScalarIO
    assert got == [C.sizeof(Cfg), C.sizeof(St), C.sizeof(Ro),
                   Cfg.slip_prob.offset, Cfg.seed.offset, Cfg.flags.offset, Cfg.stream.offset,
                   Ro.act_stride.offset, Ro.out_stride.offset, Ro.mix_a.offset,
                   C.sizeof(Sc), Sc.act_a.offset, Sc.obs.offset, Sc.u_step.offset, Sc.u_reset.offset,
                   C.sizeof(_lib.RolloutExtra), _lib.RolloutExtra.prob_code.offset, _lib.COMM_ID_BYTES]


def test_flag_and_error_constants_match_header():
    """the cfg.flags bits and error codes of the Python wrapper are the header's #defines"""
    text = open(os.path.join(ROOT, "include", "soccer_hip.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+SOCCER_(F_[A-Z_]+|E_[A-Z_]+|OK)\s+(-?\d+)u?", text)}
    assert {k: v for k, v in defs.items() if k.startswith("F_")} == {
        "F_AUTORESET": _lib.F_AUTORESET, "F_NULL_STREAM": _lib.F_NULL_STREAM, "F_HOST_MAPPED": _lib.F_HOST_MAPPED,
        "F_STEP_STATS": _lib.F_STEP_STATS, "F_STREAM_ACTIONS": _lib.F_STREAM_ACTIONS}
    flags = [v for k, v in defs.items() if k.startswith("F_")]
    assert len(set(flags)) == len(flags) and all(v & (v - 1) == 0 for v in flags)        # distinct single bits
    for name in ("E_INVALID", "E_HIP", "E_NOMEM", "E_STATE", "OK"):
        assert defs[name] == getattr(_lib, name)


@pytest.mark.parametrize("kw,msg", [
    (dict(width=4), "Width must be at least 5"),
    (dict(height=3), "Height must be at least 4"),
    (dict(slip_prob=1.5), "slip_prob"),
    (dict(n_lanes=0), "n_lanes"),
    (dict(max_steps=0), "max_steps"),
    (dict(envs_per_thread=3), "envs_per_thread"),
])
def test_create_rejects_bad_config_like_the_reference_asserts(kw, msg):
    lib = _lib.load()
    base = dict(n_lanes=8, width=5, height=4, slip_prob=0.0, max_steps=100, device=0, seed=0,
                lane_offset=0, flags=0, envs_per_thread=0, stream=None)
    base.update(kw)
    cfg = _lib.Config(**base)
    h = C.c_void_p()
    code = lib.soccer_create(C.byref(cfg), C.byref(h))
    assert code == _lib.E_INVALID and not h.value
    with pytest.raises(AssertionError, match=msg):
        _lib.check(lib, None, code)


def test_no_cpu_fallback_without_a_device():
    """On a box without a GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gym_soccer_littman94_amd import SoccerBatch
    with pytest.raises(_lib.SoccerHipError, match="no CPU path|no HIP device|failed"):
        SoccerBatch(16)


def test_c_example_builds_from_the_header_alone_and_fails_loudly_without_a_device(tmp_path):
    """examples/host.c is plain C against include/soccer_hip.h (gcc -Wall -Wextra -Werror, no HIP headers).  On a box without a GPU
    it must stop at soccer_create with the library's message — there is no CPU path to fall back to."""
    import subprocess
    import torch
    exe = str(tmp_path / "host_c")
    lib_dir = os.path.join(ROOT, "gym_soccer_littman94_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "host.c"),
                           "-o", exe, "-L", lib_dir, "-lsoccer_hip", "-Wl,-rpath," + lib_dir])
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_c_host.py runs it")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU path" in r.stderr, (r.returncode, r.stderr)

"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/aggmg_hip.h declares; the ctypes table covers exactly those symbols.  No compute
calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "aggmg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aggmg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from agglomerationmultigrid1d_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/aggmg_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms
    assert b"gfx950" in _lib.load().aggmg_version()


def test_no_device_fails_loudly():
    """Without a HIP device the product raises instead of falling back to any CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import agglomerationmultigrid1d_amd as mg
    with pytest.raises(mg.AggmgError):
        mg.Context(0)


def test_product_never_imports_oracle():
    """the package and the measurement aids under tools/ stay clear of the oracle (checker scripts
    that do use it live under tests/manual/); bench.py touches it only inside cpu_baseline()"""
    for top in ("agglomerationmultigrid1d_amd", "tools", "include", "julia"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".jl")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "aggmg_oracle" not in src and "c_oracle" not in src and "oracle/" not in src \
                        and '"oracle"' not in src, os.path.join(dirpath, f)
    # bench.py may load the restatement only inside cpu_baseline() (its docstring may name it)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    head, _, rest = bench.partition("def cpu_baseline(")
    body, sep, tail = rest.partition("\ndef ")
    for part in (head, tail):
        for needle in ("import c_oracle", "import aggmg_oracle", "from oracle", 'ROOT, "oracle"'):
            assert needle not in part, needle
    assert "import c_oracle" in body


def test_header_is_plain_c(tmp_path):
    """include/aggmg_hip.h must be consumable by a C compiler (the ABI a Julia `ccall` / cgo / JNI
    binding targets): compile a C99 translation unit that takes the address of every entry point."""
    import subprocess
    syms = header_symbols()
    src = tmp_path / "abi_check.c"
    body = "\n".join(f"    p[{i}] = (fn)&{s};" for i, s in enumerate(syms))
    src.write_text('#include "aggmg_hip.h"\n#include <stddef.h>\ntypedef void (*fn)(void);\n'
                   'fn table(fn* p) {\n' + body + "\n    return p[0];\n}\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "abi_check.o")])


def test_bench_bare_multi_gpu_form_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher in the environment must start torch.distributed.run itself as
    a child process (never exec, never initialise HIP in the parent) and pass its own arguments through."""
    import subprocess
    import sys
    import bench
    seen = {}

    class Done:
        returncode = 0
        stdout = '{"n_gpus": 4}\n'

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher (WORLD_SIZE set) a mismatch is still an error, not a second launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_rccl_not_loadable_is_reported_not_crashed(tmp_path):
    """the 'RCCL not loadable' route used to read dlerror() twice (second read NULL -> std::string + NULL).
    Forced here with AGGMG_RCCL_LIB naming a file that is not there, in a fresh process (the loader result is cached):
    AGGMG_ERR_UNSUPPORTED and the loader's message, no crash."""
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "from agglomerationmultigrid1d_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = ctypes.create_string_buffer(512)\n"
        "st = lib.aggmg_rccl_available(buf, 512)\n"
        "print(st, buf.value.decode())\n" % ROOT)
    env = dict(os.environ, AGGMG_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    st, _, msg = out.stdout.strip().partition(" ")
    from agglomerationmultigrid1d_amd import _lib
    assert int(st) == _lib.ERR_UNSUPPORTED
    assert "RCCL not loadable" in msg and "no_such_librccl.so" in msg

"""CPU: the C-ABI library loads and exports every symbol include/lidarcast.h declares; the product
fails loudly without a GPU (no compute calls are made here)."""
import os
import re

import pytest

from conftest import REPO


def test_header_and_library_agree():
    import lidarcast
    from lidarcast import _capi
    hdr = open(os.path.join(REPO, "include", "lidarcast.h")).read()
    declared = set(re.findall(r"\b(lrc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"lrc_hits", "lrc_compact_io", "lrc_scene_info", "lrc_frames", "lrc_grid", "lrc_table", "lrc_mt19937_state"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    lib = lidarcast.load()
    for name in declared:
        assert hasattr(lib, name), f"liblidarcast.so does not export {name}"
    assert lidarcast.version().startswith("lidarcast ")
    assert os.path.dirname(lidarcast.LIB_PATH).endswith("_amd")          # in-tree build


def test_struct_layouts_match_header():
    import ctypes as C
    from lidarcast._capi import LrcCompactIO, LrcFrames, LrcHits, LrcSceneInfo
    from conftest import REPO

    def header_fields(struct):
        """member names of ``typedef struct <struct> { ... } <struct>;`` in include/lidarcast.h, in order"""
        src = open(os.path.join(REPO, "include", "lidarcast.h")).read()
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [re.search(r"(\w+)\s*(\[\d+\])?\s*$", decl.strip()).group(1) for decl in body.split(";") if decl.strip()]
    assert [n for n, _ in LrcHits._fields_] == header_fields("lrc_hits")          # all pointers, same order
    assert C.sizeof(LrcHits) == 10 * 8
    assert [n for n, _ in LrcCompactIO._fields_] == header_fields("lrc_compact_io")
    assert C.sizeof(LrcCompactIO) == 14 * 8
    assert [n for n, _ in LrcFrames._fields_] == header_fields("lrc_frames")
    assert C.sizeof(LrcFrames) == 12 * 8
    assert [n for n, _ in LrcSceneInfo._fields_] == header_fields("lrc_scene_info")
    assert C.sizeof(LrcSceneInfo) == 5 * 8 + 2 * 4 + 8 + 2 * 8 + 6 * 4 + 4 * 4


def test_no_gpu_means_loud_failure():
    import lidarcast
    if lidarcast.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(lidarcast.LidarcastError, match="no HIP device"):
        lidarcast.Context(0)
    from raycast_engine import RaycastEngineCPU, RaycastEngineGPU
    for cls in (RaycastEngineGPU, RaycastEngineCPU):       # neither name hides a CPU fallback
        with pytest.raises(RuntimeError):
            cls()


def test_product_never_imports_the_oracle():
    from conftest import PKG
    for root, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(root, fn), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
                assert "liblrc_oracle" not in src and "orc_cast" not in src, fn


def test_oracle_is_only_used_where_it_may_be():
    """oracle/ is test infrastructure: besides tests/, only bench.py's cpu_baseline leg and __graft_entry__'s
    build()/smoke() may touch it; tools/ and the package never do."""
    import ast
    from conftest import REPO
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for root, _, files in os.walk(os.path.join(REPO, "tools")):
        for fn in files:
            assert not pat.search(open(os.path.join(root, fn), errors="replace").read()), f"tools/{fn} imports oracle"

    def importers(path):
        """names of the top-level functions of a module that import oracle (None = module level)"""
        tree = ast.parse(open(path).read())
        found = set()
        for top in tree.body:
            for node in ast.walk(top):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                if any(m == "oracle" or m.startswith("oracle.") for m in mods):
                    found.add(top.name if isinstance(top, (ast.FunctionDef, ast.ClassDef)) else None)
        return found
    assert importers(os.path.join(REPO, "bench.py")) <= {"cpu_baseline"}
    assert importers(os.path.join(REPO, "__graft_entry__.py")) <= {"build", "smoke"}


def test_one_hip_runtime_per_process_whichever_loads_first():
    """A ROCm wheel of torch carries its own libamdhip64.so and asks the loader for it under that unversioned name; this
    library asks for libamdhip64.so.7.  With this library loaded first (a single selected test does that) the process used to
    end up with two HIP runtimes and torch with "No HIP GPUs are available".  lidarcast._capi.load() therefore loads torch's
    copy first when torch is installed: in a fresh interpreter, in either order, exactly one runtime is mapped."""
    import subprocess
    import sys
    pytest.importorskip("torch")
    pkg = os.path.dirname(__import__("lidarcast").LIB_PATH)
    probe = ("import sys; sys.path.insert(0, %r)\n"
             "%s\n"
             "maps = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
             "print(len(maps), sorted(maps))\n" % (pkg, "%s"))
    for order in ("import lidarcast; lidarcast.load(); import torch", "import torch; import lidarcast; lidarcast.load()"):
        r = subprocess.run([sys.executable, "-c", probe % order], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        n, paths = r.stdout.strip().split(" ", 1)
        assert n == "1", f"{order}: HIP runtimes mapped: {paths}"

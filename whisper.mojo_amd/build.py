"""Builds csrc/libwhispermi.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree, so the .so
travels with the repo snapshot to the GPU box.

Staleness is decided by CONTENT (sha256 of sources, headers and flags kept in a `.stamp` file next to the library), not
by mtimes: a snapshot copy to another machine reorders mtimes, and an mtime rule then recompiles the whole library there
(≈ 90 s — that was round 1's slow smoke()).

    python whisper.mojo_amd/build.py [--force] [--dev]

--dev builds csrc/libwhispermi_dev.so with -DWM_DEV: the developer A/B switches (wm_env), timelines and the debug chains
of wm_bench_kernel exist only there (tools/ select it with WM_USE_DEV_LIB=1)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libwhispermi.so")
LIB_DEV = os.path.join(CSRC, "libwhispermi_dev.so")
SOURCES = ["whisper_mi.cpp", "kernels_encoder.hip", "kernels_decoder.hip", "kernels_frontend.hip"]
HEADERS = ["wm_device.h", "wm_kernels.h", "../../include/whisper_mi.h", "../../include/wm_synth.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(target: str, digest: str) -> bool:
    try:
        return os.path.exists(target) and open(target + ".stamp").read().strip() == digest
    except OSError:
        return False


def _write_stamp(target: str, digest: str) -> None:
    with open(target + ".stamp", "w") as f:
        f.write(digest + "\n")


def _lib_digest(flags) -> str:
    return _digest([os.path.join(CSRC, f) for f in SOURCES + HEADERS], flags)


def stale(dev: bool = False) -> bool:
    flags = FLAGS + (["-DWM_DEV"] if dev else [])
    return not _stamp_ok(LIB_DEV if dev else LIB, _lib_digest(flags))


def build(force: bool = False, verbose: bool = False, dev: bool = False) -> str:
    lib = LIB_DEV if dev else LIB
    flags = FLAGS + (["-DWM_DEV"] + os.environ.get("WM_DEV_HIPCC_FLAGS", "").split() if dev else [])  # extra -D switches for developer A/B builds
    digest = _lib_digest(flags)
    if not force and _stamp_ok(lib, digest):
        return lib
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ("_dev.o" if dev else ".o"))
        cmd = [_hipcc(), *flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    subprocess.check_call(cmd)
    _write_stamp(lib, digest)
    return lib


def build_examples(force: bool = False) -> str:
    """examples/whisper_main: main.mojo written against include/whisper_mi.hpp (the C++ host mirror), linked to the library."""
    root = os.path.dirname(HERE)
    src, hdr = os.path.join(root, "examples", "main.cpp"), os.path.join(root, "include", "whisper_mi.hpp")
    exe = os.path.join(root, "examples", "whisper_main")
    digest = _digest([src, hdr, os.path.join(root, "include", "whisper_mi.h"), os.path.join(root, "include", "wm_synth.h")])
    if not force and _stamp_ok(exe, digest):
        return exe
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler for examples/main.cpp")
    subprocess.check_call([cxx, "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), src, "-o", exe,
                           "-L", CSRC, "-lwhispermi", "-Wl,-rpath," + CSRC, "-Wl,-rpath,$ORIGIN/../whisper.mojo_amd/csrc"])
    _write_stamp(exe, digest)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
    if "--dev" not in sys.argv:
        print(build_examples(force="--force" in sys.argv))

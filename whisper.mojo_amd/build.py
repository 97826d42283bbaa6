"""Builds csrc/libwhispermi.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree, so the .so
travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libwhispermi.so")
SOURCES = ["whisper_mi.cpp", "kernels_encoder.hip", "kernels_decoder.hip", "kernels_frontend.hip"]
HEADERS = ["wm_device.h", "wm_kernels.h", "../../include/whisper_mi.h", "../../include/wm_synth.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), *FLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
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
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    subprocess.check_call(cmd)
    return LIB


def build_examples(force: bool = False) -> str:
    """examples/whisper_main: main.mojo written against include/whisper_mi.hpp (the C++ host mirror), linked to the library."""
    root = os.path.dirname(HERE)
    src, hdr = os.path.join(root, "examples", "main.cpp"), os.path.join(root, "include", "whisper_mi.hpp")
    exe = os.path.join(root, "examples", "whisper_main")
    if not force and os.path.exists(exe) and all(os.path.getmtime(f) <= os.path.getmtime(exe) for f in (src, hdr, LIB)):
        return exe
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler for examples/main.cpp")
    subprocess.check_call([cxx, "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), src, "-o", exe,
                           "-L", CSRC, "-lwhispermi", "-Wl,-rpath," + CSRC, "-Wl,-rpath,$ORIGIN/../whisper.mojo_amd/csrc"])
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_examples(force="--force" in sys.argv))

"""The C++ host mirror (include/whisper_mi.hpp) and examples/main.cpp (= main.mojo over the C-ABI).
CPU: it compiles and links against the built library, option errors exit cleanly, the C++ Tokenizer renders exactly
like the Python mirror of tokenizer.mojo.  GPU: the example's token ids equal the Python host layer's on the same seeds."""
import importlib.util
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_mod():
    spec = importlib.util.spec_from_file_location("wm_build", os.path.join(ROOT, "whisper.mojo_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def example_exe():
    mod = _build_mod()
    if not os.path.exists(mod.LIB):
        pytest.skip("libwhispermi.so not built")
    return mod.build_examples()


def test_example_builds_and_rejects_bad_options(example_exe):
    assert os.access(example_exe, os.X_OK)
    r = subprocess.run([example_exe, "--no-such-option"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown option" in r.stderr
    r = subprocess.run([example_exe, "--weights"], capture_output=True, text=True)
    assert r.returncode == 2 and "missing value" in r.stderr


def test_cpp_tokenizer_matches_python(tmp_path):
    from whisper_mojo_amd.tokenizer import Tokenizer
    cxx = shutil.which("g++") or shutil.which("c++")
    exe = os.path.join(ROOT, "tests", "cpp", "tokenizer_check")
    subprocess.check_call([cxx, "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "tokenizer_check.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "whisper.mojo_amd", "csrc"), "-lwhispermi",
                           "-Wl,-rpath," + os.path.join(ROOT, "whisper.mojo_amd", "csrc")])
    vocab = ["!", "Ġhello", "Ġworld", "<|endoftext|>", "<|en|>", "åľº", "line\\\\n", "\\n", "Ġ", "<|", "|>", "<||>", "aĠbĠ", "tail"]
    path = tmp_path / "vocab.txt"
    path.write_text("\n".join(vocab) + "\n", encoding="utf-8")  # trailing newline -> one empty last entry, as split("\n") gives
    py = Tokenizer(str(path))
    for ids in ([1, 2, 0], [3, 1, 4, 2, 5], [6, 7, 8, 9, 10, 11, 12], [13, 14, 99, -1, 1], []):
        out = subprocess.run([exe, str(path)] + [str(i) for i in ids], capture_output=True, check=True).stdout.decode("utf-8")
        n, text = out.split("\n", 1)
        assert int(n) == len(py.vocab)
        assert text == py.decode(ids), ids


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_example_ids_equal_python_host(example_exe, dtype):
    """examples/whisper_main (C++ host over the C-ABI) against whisper.mojo_amd.whisper.Whisper (Python host): same
    synthetic weights and mel, the reference's stop rule with a reachable eot."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from whisper_mojo_amd import WhisperConfig, synth, DT_F32, DT_BF16
    from whisper_mojo_amd.loader import WeightLoader
    from whisper_mojo_amd.whisper import Whisper
    cfg = WhisperConfig.micro()
    m = Whisper(cfg, compute_dtype={"f32": DT_F32, "bf16": DT_BF16}[dtype])
    m.load(WeightLoader.from_array(synth.synth_weights(cfg, 0)))
    mel = synth.synth_mel(cfg, 1000)
    free = m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=-1, max_loop=20)[0]
    eot = free[12]
    want = m.transcribe_batch(mel, prompt=(1, 2, 3, 4), eot=eot, max_loop=40)[0]
    r = subprocess.run([example_exe, "--config", "micro", "--synthetic-weights", "0", "--synthetic-mel", "1000", "--dtype", dtype,
                        "--prompt", "1,2,3,4", "--eot", str(eot), "--max-loop", "40", "--vocab", "/nonexistent"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ids = [int(t) for t in re.search(r"Token IDs:\n([0-9 ]+)\n", r.stdout).group(1).split()]
    assert ids == want
    assert ids[-1] == eot and len(ids) < 4 + 1 + 40

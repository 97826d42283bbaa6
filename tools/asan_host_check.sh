#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the whole library + examples/main.cpp in ONE executable (device code is built
# unsanitised: GPU ASan is not available on this pool), then a micro-model run through every host path main.cpp touches
# (weight load from memory, state arena, encoder + prefill + greedy loop, graph capture / replay, token copy-out, teardown).
#   tools/asan_host_check.sh build      (here: hipcc cross-compiles)
#   tools/asan_host_check.sh run        (on the GPU box)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
EXE="$ROOT/examples/whisper_main_asan"
CSRC="$ROOT/whisper.mojo_amd/csrc"
if [ "${1:-build}" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Wall -Wno-unused-function -ffp-contract=on \
    -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
    -I "$ROOT/include" -x hip "$CSRC/whisper_mi.cpp" "$CSRC/kernels_encoder.hip" "$CSRC/kernels_decoder.hip" "$CSRC/kernels_frontend.hip" \
    -x hip "$ROOT/examples/main.cpp" -o "$EXE"
  echo "built $EXE"
else
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  # A run is judged by its report, not its exit code: at process exit the HSA runtime's own teardown (__cxa_finalize ->
  # libhsa-runtime64 -> operator delete) sometimes trips an internal CHECK of the sanitizer's device allocator
  # ("dev_runtime_unloaded_") after main() has returned and "Done." is printed — not this library's code.
  run() {
    local log; log=$(mktemp)
    "$EXE" "$@" > "$log" 2>&1 || true
    cat "$log"
    if grep -q "ERROR: AddressSanitizer\|runtime error:\|SUMMARY: .*Sanitizer" "$log" || ! grep -q "^Done\.$" "$log"; then
      echo "ASAN/UBSAN host check: FAILED ($*)"; exit 1
    fi
  }
  for dt in f32 bf16; do
    run --config micro --synthetic-weights 0 --synthetic-mel 1000 --dtype $dt --prompt 1,2,3,4 --eot 532 --max-loop 40 --vocab /nonexistent
  done
  # Whisper-tiny in bf16: the encoder's row-panel / full-row GEMM launchers and the fused LayerNorm plumbing (d = 384 only)
  run --config tiny --synthetic-weights 0 --synthetic-mel 1000 --dtype bf16 --max-loop 6 --vocab /nonexistent
  echo "ASAN/UBSAN host check: clean"
fi

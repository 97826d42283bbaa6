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
  for dt in f32 bf16; do
    "$EXE" --config micro --synthetic-weights 0 --synthetic-mel 1000 --dtype $dt --prompt 1,2,3,4 --eot 532 --max-loop 40 --vocab /nonexistent
  done
  echo "ASAN/UBSAN host check: clean"
fi

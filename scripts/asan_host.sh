#!/bin/bash
# SURVEY.md section 5 (sanitizers): AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST side of the C-ABI
# library -- the pointer / size arithmetic of pa_api.hip, the handle code of lstm / convnet / transformer and the JPEG
# marker parser + table builder of mjpeg.hip -- run against the no-GPU tests (argument validation, weight-blob sizes,
# header probe on truncated and corrupted files). Device code is compiled WITHOUT instrumentation (-fno-gpu-sanitize):
# GPU AddressSanitizer needs XNACK, which this pool does not offer. CPU box only; never run this on the GPU box.
#   usage: scripts/asan_host.sh [log]        (log defaults to profiles/r06_asan_host.log)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/profiles/r06_asan_host.log}
B=$R/build/asan
mkdir -p $B
SRC="igemm pigemm psgemm igemm_bf16 patchconv patchconv_bf16 wino stem stem_pool misc preprocess detect lstm convnet transformer jpeg mjpeg savebox yolo pa_api"
for s in $SRC; do
  extra=""
  case $s in preprocess|detect|jpeg|savebox|yolo) extra="-ffp-contract=off";; esac
  if [ ! -f $B/$s.o ] || [ $R/playaid_core_amd/csrc/$s.hip -nt $B/$s.o ]; then
    hipcc --offload-arch=gfx950 -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer $extra \
      -c $R/playaid_core_amd/csrc/$s.hip -o $B/$s.o
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $B/libplayaid_hip_asan.so $(for s in $SRC; do echo $B/$s.o; done)
ASAN_SO=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
{
  echo "# $(date -u +%F) host ASan + UBSan build of libplayaid_hip (device code uninstrumented), tests/test_abi.py + the host half of tests/test_psgemm.py (weight slicing / stage-image packing)"
  echo "# runtime: $ASAN_SO"
  cd $R
  # python itself is not instrumented: preload the runtime, do not fail on its own leaks
  LD_PRELOAD=$ASAN_SO ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    PA_LIB_PATH=$B/libplayaid_hip_asan.so python3 -m pytest tests/test_abi.py tests/test_psgemm.py -m "not gpu" -x -q -p no:cacheprovider 2>&1
  echo "# exit code: $?"
  echo "# canary: the parser is told a 300-byte heap buffer holds 5000 bytes -- the instrumentation must object"
  LD_PRELOAD=$ASAN_SO ASAN_OPTIONS=detect_leaks=0 PA_LIB_PATH=$B/libplayaid_hip_asan.so python3 - <<'PY' 2>&1 | grep -E "ERROR: AddressSanitizer|READ of size|parse_header" | head -3
import ctypes, numpy as np
from playaid_core_amd import _lib, synth
lib = _lib.load()
a = np.frombuffer(synth.encode_jpeg_frames([synth.make_frame(1, 32, 32)])[0][:300], np.uint8).copy()
lib.pa_mjpeg_probe(a.ctypes.data_as(ctypes.c_void_p), 5000, (ctypes.c_int32 * 8)(), ctypes.create_string_buffer(64), 64)
PY
} | tee $LOG

#!/bin/bash
# The emulator suite under ASan + UBSan and under TSan (tests/test_sanitizers.py); summary -> profiles/<TAG>_sanitizers.txt.  No GPU needed.
#   TAG=r06 tools/run_sanitizers.sh
cd "$(dirname "$0")/.." || exit 1
TAG=${TAG:-r06}
OUT=profiles/${TAG}_sanitizers.txt
{
  echo "Sanitizer runs of the CPU thread-emulator build (make -C wasmsnark_amd/csrc emul-san SAN=address,undefined | SAN=thread), $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "gcc $(gcc -dumpversion); every kernel thread is a ucontext coroutine announced to the sanitizer runtime (tests/emul/hip_emul.cpp);"
  echo "reports are written to files (log_path): a line below ends with the number of report files its run left."
  echo
} > "$OUT"
python -u -m pytest tests/test_sanitizers.py -m sanitizer -q -s -p no:cacheprovider ${SAN_K:+-k "$SAN_K"} 2>&1 | grep --line-buffered -E "^(asan|tsan) tests/|passed|failed|error" | tee -a "$OUT"

#!/bin/bash
# The sanitizer recipe:
#   make asan tsan            builds tools/sanitize/driver.cpp + the host / oracle sources under ASan+UBSan and under TSan
#   bash tools/run_sanitizers.sh [fuzz iterations per reader, default 3000]
# runs (1) the ASan+UBSan driver, (2) the TSan driver, (3) the Python CPU test files that exercise the host library and the
# oracle against the ASan+UBSan shared libraries (the sanitizer runtime preloaded into python).  Output: profiles/r03/sanitizers.txt
set -u
cd "$(dirname "$0")/.."
N=${1:-3000}
OUT=profiles/r03/sanitizers.txt
mkdir -p profiles/r03
make asan tsan > /dev/null || { echo "sanitizer builds failed"; exit 1; }
{
echo "== ASan + UBSan: tools/sanitize/driver.cpp (threaded SAH builds, oracle workers, readers, $N mutations per reader)"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 build/asan/driver --fuzz "$N" --dir /tmp/trc_sanitize_asan 2>&1 | tail -20
echo "exit code ${PIPESTATUS[0]}"
echo "== TSan: the same driver (fewer mutations: the readers are single-threaded)"
TSAN_OPTIONS=halt_on_error=0 build/tsan/driver --fuzz 200 --dir /tmp/trc_sanitize_tsan 2>&1 | tail -20
echo "exit code ${PIPESTATUS[0]}"
echo "== ASan + UBSan: python CPU tests on the sanitized libtrc_host.so / liboracle.so (LD_PRELOAD of the runtime)"
ASAN_RT=$(g++ -print-file-name=libasan.so)
LD_PRELOAD="$ASAN_RT" ASAN_OPTIONS=detect_leaks=0 TRC_HOST_LIB="$PWD/build/asan/libtrc_host.so" TRC_ORACLE_DIR="$PWD/build/asan" \
  python -m pytest -q -p no:cacheprovider -m "not gpu" tests/test_bvh_builder.py tests/test_host_scene.py tests/test_oracle_kat.py \
  tests/test_oracle_lbvh.py tests/test_oracle_volume.py tests/test_oracle_sppm.py tests/test_output_stage.py tests/test_sobol.py \
  tests/test_pbrt_scene.py tests/test_pbrt_reader.py tests/test_ply_hdr_readers.py tests/test_envmap.py tests/test_oracle_render.py 2>&1 | tail -6
echo "exit code ${PIPESTATUS[0]}"
} | tee "$OUT"
grep -q "ERROR: \|WARNING: ThreadSanitizer\|runtime error" "$OUT" && { echo "SANITIZER REPORTS FOUND"; exit 1; }
echo "no sanitizer reports" | tee -a "$OUT"

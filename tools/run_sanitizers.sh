#!/bin/bash
# The sanitizer recipe:
#   make asan tsan            builds tools/sanitize/driver.cpp + the host / oracle sources under ASan+UBSan and under TSan
#   bash tools/run_sanitizers.sh [fuzz iterations per reader, default 3000] [round directory, default r06]
# runs (1) the ASan+UBSan driver, (2) the TSan driver, (3) the Python CPU test files that exercise the host library and the
# oracle against the ASan+UBSan shared libraries (the sanitizer runtime preloaded into python).
# Every leg's FULL output goes to build/sanitize_logs/<leg>.log and is what the report patterns are searched in (a
# sanitizer report prints its "ERROR:" header ahead of dozens of lines of stacks, so a tail would cut it off); the summary
# profiles/<round>/sanitizers.txt shows the tails for reading.  The script fails if any leg exits non-zero or any log holds a report.
set -u
cd "$(dirname "$0")/.."
N=${1:-3000}
ROUND=${2:-r06}
OUT=profiles/$ROUND/sanitizers.txt
LOGS=build/sanitize_logs
mkdir -p "profiles/$ROUND" "$LOGS"
make asan tsan > "$LOGS/build.log" 2>&1 || { echo "sanitizer builds failed (see $LOGS/build.log)"; exit 1; }

ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  build/asan/driver --fuzz "$N" --dir /tmp/trc_sanitize_asan > "$LOGS/asan.log" 2>&1
rc_asan=$?
TSAN_OPTIONS=halt_on_error=0 build/tsan/driver --fuzz 200 --dir /tmp/trc_sanitize_tsan > "$LOGS/tsan.log" 2>&1
rc_tsan=$?
ASAN_RT=$(g++ -print-file-name=libasan.so)
LD_PRELOAD="$ASAN_RT" ASAN_OPTIONS=detect_leaks=0 TRC_HOST_LIB="$PWD/build/asan/libtrc_host.so" TRC_ORACLE_DIR="$PWD/build/asan" \
  python -m pytest -q -p no:cacheprovider -m "not gpu" tests/test_bvh_builder.py tests/test_host_scene.py tests/test_oracle_kat.py \
  tests/test_oracle_lbvh.py tests/test_oracle_volume.py tests/test_oracle_sppm.py tests/test_output_stage.py tests/test_sobol.py \
  tests/test_pbrt_scene.py tests/test_pbrt_reader.py tests/test_ply_hdr_readers.py tests/test_envmap.py tests/test_oracle_render.py \
  tests/test_cpu_baseline.py tests/test_capture_layout.py tests/test_detmath.py > "$LOGS/pytest.log" 2>&1
rc_py=$?

PATTERN='ERROR: \|WARNING: ThreadSanitizer\|runtime error\|SUMMARY: .*Sanitizer'
reports=0
for leg in asan tsan pytest; do
  if grep -q "$PATTERN" "$LOGS/$leg.log"; then reports=1; fi
done
{
echo "== ASan + UBSan (+ float-cast-overflow): tools/sanitize/driver.cpp (threaded SAH builds, oracle workers, readers, $N mutations per reader)"
tail -20 "$LOGS/asan.log"; echo "exit code $rc_asan"
echo "== TSan: the same driver (fewer mutations: the readers are single-threaded)"
tail -20 "$LOGS/tsan.log"; echo "exit code $rc_tsan"
echo "== ASan + UBSan: python CPU tests on the sanitized libtrc_host.so / liboracle.so (LD_PRELOAD of the runtime)"
tail -6 "$LOGS/pytest.log"; echo "exit code $rc_py"
echo "== report lines in the full logs ($LOGS/*.log):"
grep -n "$PATTERN" "$LOGS"/asan.log "$LOGS"/tsan.log "$LOGS"/pytest.log | head -40
if [ "$reports" -ne 0 ] || [ "$rc_asan" -ne 0 ] || [ "$rc_tsan" -ne 0 ] || [ "$rc_py" -ne 0 ]; then
  echo "FAILED: reports=$reports exit codes asan=$rc_asan tsan=$rc_tsan pytest=$rc_py"
else
  echo "no sanitizer reports, every leg exited 0"
fi
} | tee "$OUT"
[ "$reports" -eq 0 ] && [ "$rc_asan" -eq 0 ] && [ "$rc_tsan" -eq 0 ] && [ "$rc_py" -eq 0 ]

#!/bin/bash
# The GPU test suite with everything a failure needs kept (a failing run once left only a test name):
#     gpurun --timeout 900 -- "bash tools/gpu_suite.sh $(git rev-parse --short HEAD) [pytest args]"
# writes gpurun_out/suite_<commit>.log (pytest -rA --tb=long, every test's outcome and the full traceback of any
# failure) and gpurun_out/gpu_suite_events.log (tests/conftest.py: per test, the thread's stale HIP error state at
# its start, outcome, seconds).  One process, one run: never in a retry loop.
set -o pipefail
COMMIT=${1:-unknown}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
LOG=$R/gpurun_out/suite_$COMMIT.log
{ echo "# commit $COMMIT  $(date -u +%Y-%m-%dT%H:%MZ)"; python -c "import torch; print('#', torch.cuda.get_device_name(0))"; } > $LOG
rm -f $R/gpurun_out/gpu_suite_events.log
python -m pytest tests -m gpu -x -q -rA --tb=long -p no:cacheprovider "$@" >> $LOG 2>&1
rc=$?
tail -n 15 $LOG
grep -v "stale_hip_error_at_start=0" $R/gpurun_out/gpu_suite_events.log | head -20
exit $rc

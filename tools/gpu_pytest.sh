#!/bin/bash
# tools/gpu_pytest.sh LOGNAME [pytest args...] — run pytest on the GPU box, keep the log under gpurun_out/, print the verdict
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=$R/gpurun_out/$1.log; shift
mkdir -p $R/gpurun_out
cd $R && timeout ${PYTEST_TIMEOUT:-1500} python -m pytest "$@" > $LOG 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" $LOG | tail -3
grep -E "^(FAILED|ERROR)" $LOG | head -20

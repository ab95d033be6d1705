#!/bin/bash
# End to end through the reference's UNMODIFIED CLI on the drop-in libhighs (integration/_build): the headline LP as an
# MPS file -> Highs::readModel (this repository's reader) -> solver=pdlp on the MI355X -> solution.  Prints the CLI's own
# timing lines.  Usage: tools/e2e_cli.sh [b|c|a] [extra highs flags...]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
CFG=${1:-b}; shift || true
python $R/tools/mps_bench.py --config $CFG --threads 0 --reps 1 > /tmp/e2e_read.json 2>/tmp/e2e_read.err
cat /tmp/e2e_read.json
export LD_LIBRARY_PATH=$R/integration/_build:$R/highs_amd/lib:${LD_LIBRARY_PATH:-}
cd /tmp
S=$(date +%s.%N)
$R/integration/_build/highs_ref_cli --solver=pdlp --presolve=off "$@" /tmp/mps_bench_$CFG.mps > /tmp/e2e_cli.log 2>&1
E=$(date +%s.%N)
grep -E "Running HiGHS|LP .* has|MI355X|Model status|Objective value|iterations|run time|P-D|infeas|residual|WARNING|ERROR" /tmp/e2e_cli.log | head -40
echo "process wall time: $(echo "$E - $S" | bc -l 2>/dev/null || python -c "print($E-$S)") s"

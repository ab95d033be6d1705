#!/bin/bash
# tools/r6_bench_lines.sh [ROUND] — only the bench lines of tools/make_profiles.sh (same commands, same file names), for a
# change that touches what bench.py prints and not what the kernels do; profiles/pmc_traffic.json is read as it is
export PDLP_MI355X_DEV=1
R=$(cd "$(dirname "$0")/.." && pwd); RND=${1:-r06}; OUT=$R/gpurun_out/profiles_$RND; mkdir -p $OUT; cd $R
python bench.py > $OUT/${RND}_bench_1M.json 2> $OUT/bench_1M.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${RND}_bench_1M_driver_flags.json 2>> $OUT/bench_1M.err
python bench.py --config a > $OUT/${RND}_bench_100k.json 2>> $OUT/bench_1M.err
python bench.py --config c > $OUT/${RND}_bench_structured.json 2>> $OUT/bench_1M.err
python bench.py --config d > $OUT/${RND}_bench_staircase_dense_columns.json 2>> $OUT/bench_1M.err
python bench.py --config e > $OUT/${RND}_bench_heldout_tall.json 2>> $OUT/bench_1M.err
python bench.py --config f > $OUT/${RND}_bench_heldout_powerlaw_band.json 2>> $OUT/bench_1M.err
python bench.py --config qp > $OUT/${RND}_bench_qp.json 2>> $OUT/bench_1M.err
python bench.py --config qpn > $OUT/${RND}_bench_qp_sparse_hessian.json 2>> $OUT/bench_1M.err
python bench.py --solver hipdlp > $OUT/${RND}_bench_hipdlp_1M.json 2>> $OUT/bench_1M.err
tail -c 300 $OUT/${RND}_bench_1M.json

#!/bin/bash
# tools/bringup_multi_gpu.sh [N] [OUTDIR] — staged bring-up of the multi-GPU path on a node with N GPUs (default: all visible; with
# fewer devices than ranks the ranks are folded onto the devices there are, as in this repository's one-GPU tests).  One
# JSON line per stage into OUTDIR/bringup.jsonl, each stage with its own timeout, in the order the layers depend on each
# other, so that a failed first contact with the hardware says WHICH layer failed:
#   1 peers     device count, peer-access matrix                                  (tools/bringup_multi_gpu.py peers)
#   2 mesh      HIP IPC arena mapping + the exchange's known-answer self-test + flag-hop latency + 200 sharded iterations
#   3 rccl      ncclAllReduce with N ranks (torch.distributed, backend nccl = RCCL)
#   4 bench     python bench.py --gpus 2 / 4 / .. / N (mesh -> mesh with fences -> the same with a kernel per exchange step -> RCCL
#               fall-back chain inside bench.py):
#               the N > 1 line carries ranks_bit_identical, exchange, exchange_fallback, exchange_waits per phase next to scaling_model
# Nothing here needs the reference tree or the network.
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "import ctypes; h=ctypes.CDLL('libamdhip64.so'); n=ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); print(n.value)" 2>/dev/null || echo 0)
N=${1:-$NDEV}; OUT=${2:-gpurun_out/bringup}; mkdir -p $OUT; LOG=$OUT/bringup.jsonl; : > $LOG
say() { echo "$1" | tee -a $LOG; }
stage() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  local res; res=$(timeout $to "$@" 2>$OUT/$name.err | grep '^{' | tail -1); local rc=${PIPESTATUS[0]}
  if [ -z "$res" ]; then res="{\"stage\": \"$name\", \"ok\": false, \"error\": \"no result line (rc $rc, timeout ${to}s); stderr in $OUT/$name.err\"}"; fi
  say "$res"
}
stage peers 60 python tools/bringup_multi_gpu.py peers
ID=$(python -c "import os; print(os.urandom(128).hex())")
for W in $(seq 2 $N | awk '$1==2||$1==4||$1==8||$1=='$N); do
  # (eight or more ranks FOLDED onto fewer devices: one hardware queue per rank process, or the device's queues are
  # oversubscribed and a rank spins on a peer that is switched out — profiles/r06_development_measurements.md section 9)
  if [ "$W" -ge 8 ] && [ "$NDEV" -lt "$W" ]; then export GPU_MAX_HW_QUEUES=1; fi
  pids=(); for r in $(seq 0 $((W-1))); do
    ( timeout 300 python tools/bringup_multi_gpu.py mesh $r $W $ID > $OUT/mesh_w${W}_r$r.json 2> $OUT/mesh_w${W}_r$r.err ) & pids+=($!)
  done
  fail=0; for p in "${pids[@]}"; do wait $p || fail=1; done
  res=$(grep -h '^{' $OUT/mesh_w${W}_r0.json | tail -1)
  [ -z "$res" ] && res="{\"stage\": \"mesh\", \"ok\": false, \"world\": $W, \"error\": \"rank 0 printed nothing; stderr in $OUT/mesh_w${W}_r0.err\"}"
  [ $fail = 1 ] && res=$(echo "$res" | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['ok']=False; d['some_rank_failed']=True; print(json.dumps(d))")
  say "$res"
  ID=$(python -c "import os; print(os.urandom(128).hex())")
done
if [ "$NDEV" -ge 2 ]; then
  stage rccl 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29531 tools/bringup_multi_gpu.py rccl
  for W in $(seq 2 $N | awk '$1==2||$1==4||$1==8||$1=='$N); do
    res=$(timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus $W --steps 20 --warmup 5 2>$OUT/bench_w$W.err | grep '^{' | tail -1)
    if [ -n "$res" ]; then echo "$res" > $OUT/bench_w$W.json
      say "$(echo "$res" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'stage':'bench','ok':bool(d.get('ranks_bit_identical')),'world':d['n_gpus'],'value':d['value'],'ms_per_step':d['ms_per_step'],'exchange':d.get('exchange'),'exchange_fallback':d.get('exchange_fallback'),'exchange_waits':d.get('exchange_waits'),'model_us_per_trial':(d.get('scaling_model') or {}).get('G=%d'%d['n_gpus'],{}).get('us_per_trial'),'measured_us_per_trial':(d.get('scaling_model') or {}).get('measured_us_per_trial')}))")"
    else say "{\"stage\": \"bench\", \"ok\": false, \"world\": $W, \"error\": \"no JSON line; stderr in $OUT/bench_w$W.err\"}"; fi
  done
else
  say "{\"stage\": \"rccl\", \"ok\": null, \"skipped\": \"one device visible: ncclAllReduce with N > 1 ranks needs N devices\"}"
  say "{\"stage\": \"bench\", \"ok\": null, \"skipped\": \"one device visible: bench.py --gpus N > 1 needs N devices (the folded path is covered by tests/test_gpu_bench_multirank.py)\"}"
fi
echo "bring-up summary:"; python -c "
import json
for l in open('$LOG'):
    d = json.loads(l); print(' ', d.get('stage'), d.get('world', ''), 'ok' if d.get('ok') else ('skipped' if d.get('ok') is None else 'FAILED'), d.get('error') or d.get('note') or '')
"

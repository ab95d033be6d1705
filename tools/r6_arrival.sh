export PDLP_MI355X_DEV=1
cd /root/repo
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; }
for cfg in b c d; do
  python bench.py --config $cfg --cpu-iters 0 2>/dev/null | line $cfg
  PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch" | grep fused
done

for i in 1 2; do for e in ${A:-AVSD_GN_FUSED=1} ${B:-AVSD_GN_FUSED=0}; do
env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-vae --no-roofline --also-clips 0 --no-precise 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], d['ms_per_step'])"
done; done

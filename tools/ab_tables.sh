for i in 1 2; do
for t in tiles_oldvals tiles_gfx950; do
AVSD_TILE_CACHE=tmp_ab/$t.json python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-vae --no-roofline --also-clips 0 --no-precise 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
done; done

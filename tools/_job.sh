cd /root/repo
python tools/tune_tiles.py --out gpurun_out/tiles_gfx950.json > gpurun_out/tune_r2h.log 2>&1; tail -2 gpurun_out/tune_r2h.log
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old table', d['value'], d['batched']['value'])"
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new table', d['value'], d['batched']['value'])"
done

cd /root/repo
python tools/tune_tiles.py --xcd-only --out gpurun_out/tiles_xcd.json > gpurun_out/tune_r2g.log 2>&1; tail -3 gpurun_out/tune_r2g.log
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | cut -c1-120
AVSD_TILE_CACHE=gpurun_out/tiles_xcd.json python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | cut -c1-120
done

cd /root/repo
python tools/tune_tiles.py --extend --out gpurun_out/tiles_gfx950.json > gpurun_out/tune_r2j.log 2>&1; tail -2 gpurun_out/tune_r2j.log
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-vae --no-roofline --also-clips 0 --f32-residual | cut -c1-120
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-vae --no-roofline --also-clips 0 | cut -c1-120
AVSD_PRECISION=fp16 AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-vae --no-roofline --also-clips 0 --f32-residual | cut -c1-120

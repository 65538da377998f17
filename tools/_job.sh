set -x
cd /root/repo
python tools/tune_tiles.py --extend --out gpurun_out/tiles_gfx950.json > gpurun_out/tune_r2e.log 2>&1; tail -3 gpurun_out/tune_r2e.log
for sp in 0 1; do AVSD_SHARE_PREFIX=$sp AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | cut -c1-400; done
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_r2h_all.log 2>&1; tail -5 gpurun_out/tests_r2h_all.log
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5

set -x
cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -4
python tools/tune_tiles.py --out gpurun_out/tiles_gfx950.json > gpurun_out/tune_r2f.log 2>&1; tail -3 gpurun_out/tune_r2f.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | cut -c1-300
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-vae --no-roofline --also-clips 4 | cut -c1-300

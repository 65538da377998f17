cd /root/repo
python tools/cfg4_run.py 2>&1 | tail -1
AVSD_TILE_CACHE=gpurun_out/tiles_gfx950.json python tools/cfg4_run.py 2>&1 | tail -1

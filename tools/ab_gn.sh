# same-box A/B of the GroupNorm prologue: off / 32 x 32 level only / 16 x 16 and up / everywhere it fits; table = $1
for i in 1 2; do
for g in "0 1024" "1 1024" "1 256" "1 1"; do
set -- $g $3
AVSD_CONV3R_GN=$1 AVSD_CONV3R_GN_MINPIX=$2 AVSD_TILE_CACHE=${TABLE:-asva_amd/tiles_gfx950.json} python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-vae --no-roofline --also-clips 0 --no-precise 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn_prologue=$1 minpix=$2', d['value'], d['ms_per_step'])"
done; done

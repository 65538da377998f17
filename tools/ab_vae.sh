# same-box A/B of two tile tables on the VAE decode leg of bench.py: bash tools/ab_vae.sh old.json new.json
for i in 1 2; do for t in $1 $2; do
AVSD_TILE_CACHE=$t python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also-clips 0 --no-precise 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['value'], d['vae_decode']['clips_per_s'], d['vae_decode']['tflops'])"
done; done

cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"; }
for i in 1 2; do
a=$(python bench.py --model M2TRACK --steps 100 --warmup 10 --no-secondary 2>/dev/null | ms)
b=$(O3D_FLAT_BATCH=0 python bench.py --model M2TRACK --steps 100 --warmup 10 --no-secondary 2>/dev/null | ms)
c=$(O3D_WGRAD_GROUP=0 O3D_HEAD_PAIRS=0 python bench.py --model M2TRACK --steps 100 --warmup 10 --no-secondary 2>/dev/null | ms)
d=$(O3D_FLAT_ADAM=0 python bench.py --model M2TRACK --steps 100 --warmup 10 --no-secondary 2>/dev/null | ms)
echo "M2TRACK default $a  noflat $b  nogroup $c  torchadam $d"
done

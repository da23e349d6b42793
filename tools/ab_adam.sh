run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for i in 1 2; do
IMT_ADAM_OVERLAP=0 run "in-order"
run "overlap, 2048 blocks"
IMT_ADAM_BLOCKS=256 run "overlap, 256 blocks"
IMT_ADAM_BLOCKS=512 run "overlap, 512 blocks"
IMT_ADAM_BLOCKS=1024 run "overlap, 1024 blocks"
IMT_ADAM_OVERLAP=0 IMT_ADAM_BLOCKS=1024 run "in-order, 1024 blocks"
done

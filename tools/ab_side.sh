cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_side
for side in 0 1; do for v in base norun; do
  lib=$PWD/build/libimt_hip_$v.so; [ $v = base ] && lib=$PWD/imagetranslate_amd/libimt_hip.so
  IMT_DW_SIDE_STREAM=$side IMT_LIB=$lib python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --breakdown > gpurun_out/ab_side/${v}_$side.json 2> gpurun_out/ab_side/${v}_$side.txt
  echo "side=$side $v $(grep -E "^\[bench\] gpu|gemm_bf16_tn_grouped|sum of" gpurun_out/ab_side/${v}_$side.txt | tr '\n' ' ')"
done; done
